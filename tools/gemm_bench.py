#!/usr/bin/env python3
"""Times every GEMM shape of one C3 training step through the C ABI."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from m3t import ops

dev = "cuda:0"
EXCL = os.environ.get("EXCL") == "1"      # pass M3T_GEMM_EXCLUSIVE (the planner may pick the 256-tile kernel)
M = 9600
shapes = []   # (name, tA, tB, M, N, K, count, seg)
def lin(name, I, O, cnt=1):
    shapes.append((name + " fwd", 0, 1, M, O, I, cnt, False))
    shapes.append((name + " dX", 0, 0, M, I, O, cnt, False))
    shapes.append((name + " dW", 1, 0, O, I, M, cnt, False))
lin("audio.l0 ih", 128, 768, 2); lin("audio.l1 ih", 512, 768, 2)
lin("gru_va.l0 ih", 256, 1536, 4); lin("gru_va.l1 ih", 1024, 1536, 4)
lin("proj_v", 2048, 512); lin("scorer ih", 512, 384, 4)
lin("fusion.l0 ih", 512, 1536, 2); lin("fusion.l1 ih", 1024, 1536, 2)
lin("fc0", 1024, 512); lin("fc2", 512, 9)
for nm, H, cnt in (("audio", 256, 4), ("gru_va", 512, 8), ("scorer", 128, 4), ("fusion", 512, 4)):
    shapes.append((nm + " dW_hh", 1, 0, 3 * H, H, 32 * 299, cnt, True))

tot = 0.0
for name, tA, tB, m, n, k, cnt, seg in shapes:
    A = torch.randn((k, m) if tA else (m, k), device=dev)
    Bm = torch.randn((n, k) if tB else (k, n), device=dev)
    if seg:
        A = torch.randn(9600, m, device=dev); Bm = torch.randn(9600, 2 * n, device=dev)
    Cm = torch.empty(m, n, device=dev)
    def run():
        if seg:
            ops.sgemm(1, 0, m, n, k, A, 0, m, Bm, 0, 2 * n, Cm, 0, n, seg=(299, 300, 1, 0), exclusive=EXCL)
        else:
            ops.sgemm(tA, tB, m, n, k, A, 0, A.shape[1], Bm, 0, Bm.shape[1], Cm, 0, n, exclusive=EXCL)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    tf = 2.0 * m * n * k / us / 1e6
    tot += us * cnt
    print("%-18s tA%d tB%d M%5d N%5d K%5d x%d : %8.1f us  %6.1f TF/s" % (name, tA, tB, m, n, k, cnt, us, tf))
print("total per step: %.2f ms" % (tot / 1e3))
