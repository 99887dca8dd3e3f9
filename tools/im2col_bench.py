#!/usr/bin/env python3
"""m3t_im2col3d alone on the C5 stem's convolutions (8 clips x 64 frames of 112 x 112): time and store bandwidth per layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import ctypes as C
import torch
from m3t import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
# (Ci, T, H, W, k, stride, pad): the v2p_split stem (models/backbone.py) on 8 x 64 x 112 x 112 input
layers = [(3, 64, 112, 112, (3, 3, 3), (1, 2, 2), (1, 0, 0)), (64, 64, 27, 27, (3, 3, 3), (1, 1, 1), (1, 0, 0)),
          (128, 64, 12, 12, (3, 3, 3), (1, 1, 1), (1, 0, 0)), (256, 64, 5, 5, (3, 3, 3), (1, 1, 1), (1, 0, 0)),
          (512, 64, 3, 3, (3, 3, 3), (1, 1, 1), (1, 0, 0))]
tot = 0.0
for Ci, T, H, W, k, st, pd in layers:
    N = 8
    x = torch.randn(N, Ci, T, H, W, device=dev)
    To, Ho, Wo = (T + 2 * pd[0] - k[0]) // st[0] + 1, (H + 2 * pd[1] - k[1]) // st[1] + 1, (W + 2 * pd[2] - k[2]) // st[2] + 1
    rows = N * To * Ho * Wo
    Kc = Ci * k[0] * k[1] * k[2]
    Kp = Kc if Kc % 64 == 0 else (Kc + 127) // 128 * 128
    rows_p = (rows + 31) // 32 * 32
    out = torch.empty(rows_p, Kp, device=dev)
    slot = torch.zeros(1, dtype=torch.int64, device=dev)
    def f():
        rc = lib.m3t_im2col3d(C.c_void_p(x.data_ptr()), N, Ci, T, H, W, k[0], k[1], k[2], st[0], st[1], st[2], pd[0], pd[1], pd[2],
                              C.c_void_p(out.data_ptr()), rows_p, Kp, C.c_void_p(slot.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    tot += us
    print("Ci %3d  %3d x %3d x %3d -> rows %8d x Kp %5d = %6.2f GB : %8.1f us  %5.2f TB/s stored" % (Ci, T, H, W, rows_p, Kp, rows_p * Kp * 4 / 1e9, us, rows_p * Kp * 4 / us / 1e6))
print("total %.2f ms" % (tot / 1e3))
