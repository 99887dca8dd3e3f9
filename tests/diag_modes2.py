#!/usr/bin/env python3
"""Diagnostics: WHERE the bf16x6 and fp32 GEMM modes differ in the dB-scale C3 step."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = {"default": {}, "gemm_fp32": {"M3T_GEMM_X6": "0"}, "gemm_old": {"M3T_GEMM_X6D": "0", "M3T_GEMM_X6C": "0"}, "nonarrow": {"M3T_GEMM_NARROW": "0"}}
for name, env in MODES.items():
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "diag_modes.py"), "db", "child", name], env=dict(os.environ, M3T_SCAN_LOCK="0", **env))
import torch, numpy as np
ref = torch.load("/tmp/diag_gemm_fp32.pt")
for name in ("default", "gemm_old", "nonarrow"):
    cur = torch.load("/tmp/diag_%s.pt" % name)
    print("==== %s vs gemm_fp32" % name)
    for n in ("fusion.fc.2.weight", "fusion.fc.0.weight", "fusion.fc.0.bias", "fusion.gru.weight_ih_l1", "fusion.gru.weight_hh_l1", "fusion.gru.bias_hh_l1", "proj_v.weight", "audio.gru.weight_ih_l0"):
        a, b = cur[n].double(), ref[n].double()
        rms = float(b.norm()) / np.sqrt(b.numel())
        d = (a - b).abs() / rms
        bad = d > 1e-4
        line = "%-28s shape %s relL2 %.1e max %.1e frac>1e-4 %.3f" % (n, tuple(a.shape), float((a - b).norm() / b.norm()), float(d.max()), float(bad.double().mean()))
        if a.dim() == 2 and bad.any():
            rows, cols = bad.any(1).nonzero().flatten(), bad.any(0).nonzero().flatten()
            line += " rows[%d..%d] n=%d cols[%d..%d] n=%d" % (int(rows.min()), int(rows.max()), rows.numel(), int(cols.min()), int(cols.max()), cols.numel())
        print(line)
