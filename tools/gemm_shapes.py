#!/usr/bin/env python3
"""Times a few large GEMM shapes through the C ABI (env switches select the kernel variant)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from m3t import ops
dev = "cuda:0"
shapes = [(0, 1, 9600, 1536, 1024), (0, 0, 9600, 1024, 1536), (1, 0, 1536, 1024, 9600), (0, 1, 8192, 2048, 2048), (0, 1, 9600, 1536, 256)]
for tA, tB, m, n, k in shapes:
    A = torch.randn((k, m) if tA else (m, k), device=dev)
    Bm = torch.randn((n, k) if tB else (k, n), device=dev)
    Cm = torch.empty(m, n, device=dev)
    run = lambda: ops.sgemm(tA, tB, m, n, k, A, 0, A.shape[1], Bm, 0, Bm.shape[1], Cm, 0, n)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("tA%d tB%d M%5d N%5d K%5d : %8.1f us  %6.1f TF/s" % (tA, tB, m, n, k, us, 2.0 * m * n * k / us / 1e6))
