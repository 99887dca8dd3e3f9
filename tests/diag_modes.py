#!/usr/bin/env python3
"""Diagnostics (not a test): the C3 step at 32 x 300 under several kernel-selection switches, full gradient tensors compared
between modes on the GPU, and each mode's digests against the reference golden.  usage: diag_modes.py [normal|db]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
audio = sys.argv[1] if len(sys.argv) > 1 else "db"
MODES = {"default": {}, "scan_fp32mfma": {"M3T_SCAN_X6": "0"}, "scan_perstep": {"M3T_SCAN_PERSIST": "0"}, "gemm_fp32": {"M3T_GEMM_X6": "0"},
         "nosolo": {"M3T_SCAN_SOLO": "0"}}      # the process-wide kernel-selection switches (README.md)
if len(sys.argv) > 2 and sys.argv[2] == "child":
    for p in (ROOT, os.path.join(ROOT, "m3f.pytorch_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import numpy as np, torch
    from golden.recipe import fill_module
    from m3t.workloads import AVFeatureGraph, make_c3_step
    from test_gpu_bench_path import _c3_batch, _digest_err
    from conftest import load_golden
    g = load_golden("c3_av_graph_b32_db" if audio == "db" else "c3_av_graph_b32")
    model = fill_module(AVFeatureGraph(128, 256, 512), 12346).to("cuda:0")
    batch = _c3_batch(12345, 32, 300, 128, 256, audio=audio)
    ddp, step = make_c3_step(model, batch, max_norm=0.0)
    loss, stats, y = step()
    torch.cuda.synchronize()
    worst = max((max(_digest_err(p.grad, g["gd." + n])), n) for n, p in model.named_parameters())
    print("MODE %s: |y-ref| %.2e worst digest err %.2e (%s)" % (sys.argv[3], float((y.detach().cpu().double() - torch.from_numpy(g["y"]).double()).abs().max()), worst[0], worst[1]), flush=True)
    torch.save({n: p.grad.detach().cpu() for n, p in model.named_parameters()}, "/tmp/diag_%s.pt" % sys.argv[3])
    sys.exit(0)
for name, env in MODES.items():
    subprocess.run([sys.executable, os.path.abspath(__file__), audio, "child", name], env=dict(os.environ, M3T_SCAN_LOCK="0", **env))
import torch, numpy as np
base = torch.load("/tmp/diag_default.pt")
for name in MODES:
    if name == "default":
        continue
    other = torch.load("/tmp/diag_%s.pt" % name)
    rows = []
    for n in base:
        a, b = base[n].double().reshape(-1), other[n].double().reshape(-1)
        rms = float(b.norm()) / np.sqrt(b.numel())
        rows.append((float((a - b).abs().max()) / rms, float((a - b).norm() / b.norm()), n))
    rows.sort(reverse=True)
    print("default vs %s: " % name + "; ".join("%s max|d|/rms %.1e relL2 %.1e" % (r[2], r[0], r[1]) for r in rows[:4]))
