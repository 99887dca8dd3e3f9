#!/usr/bin/env python3
"""One GEMM shape through the C ABI with magnitude slots given (the kernel alone): python tools/gemm_one.py tA tB M N K [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from m3t import ops
tA, tB, m, n, k = [int(v) for v in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dev = "cuda:0"
A = torch.randn((k, m) if tA else (m, k), device=dev)
Bm = torch.randn((n, k) if tB else (k, n), device=dev)
Cm = torch.empty(m, n, device=dev)
sl = ops.amax_slots(2, A.device)
ops.measure_amax([(A, sl.data_ptr()), (Bm, sl.data_ptr() + 8)])
run = lambda: ops.sgemm(tA, tB, m, n, k, A, 0, A.shape[1], Bm, 0, Bm.shape[1], Cm, 0, n, amax=(sl.data_ptr(), sl.data_ptr() + 8))
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
import hashlib
ref = (A.double().t() if tA else A.double()) @ (Bm.double().t() if tB else Bm.double())
err = float((Cm.double() - ref).abs().max() / ref.abs().max())
print("tA%d tB%d M%d N%d K%d: %.1f us  %.1f TF/s  digest %s  rel.err %.2e" % (tA, tB, m, n, k, us, 2.0 * m * n * k / us / 1e6,
      hashlib.sha256(Cm.cpu().numpy().tobytes()).hexdigest()[:12], err))
