#!/usr/bin/env python3
"""The weight-gradient walk (m3t_conv3d_wgrad_taps) on the per-frame ResNet-18's layers at 512 frames and on the 3-D stem: us and TFLOP/s per layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from m3t import ops, _lib
lib = ops.lib()
dev = torch.device("cuda", 0)
LAYERS = [("stem 3(4)->64 5x7x7 s122", 8, 4, 64, 64, 112, 112, (5, 7, 7), (1, 2, 2), (2, 3, 3)),
          ("l1 64->64 28^2", 512, 64, 64, 1, 28, 28, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
          ("l2.0 64->128 s2", 512, 64, 128, 1, 28, 28, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
          ("l2 128->128 14^2", 512, 128, 128, 1, 14, 14, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
          ("l3.0 128->256 s2", 512, 128, 256, 1, 14, 14, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
          ("l3 256->256 7^2", 512, 256, 256, 1, 7, 7, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
          ("l4.0 256->512 s2", 512, 256, 512, 1, 7, 7, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
          ("l4 512->512 4^2", 512, 512, 512, 1, 4, 4, (1, 3, 3), (1, 1, 1), (0, 1, 1))]
for name, N, Ci, Co, T, H, W, k, st, pd in LAYERS:
    To, Ho, Wo = [(d + 2 * p - kk) // s + 1 for d, p, kk, s in zip((T, H, W), pd, k, st)]
    rows, srows = N * To * Ho * Wo, N * T * H * W
    if rows % 32:
        print(name, "rows % 32"); continue
    x = torch.randn(srows, Ci, device=dev); dy = torch.randn(rows, Co, device=dev)
    sl = ops.amax_slots(2, dev)
    ops.measure_amax([(x, sl.data_ptr()), (dy, sl.data_ptr() + 8)])
    taps = k[0] * k[1] * k[2]
    Mp = (taps * Ci + 127) // 128 * 128
    tiles = (Mp // 128) * ((Co + 127) // 128)
    want = max(1, min((1536 + tiles - 1) // tiles, rows // 256))
    ws = ops.workspace(dev, min(want * Mp * Co * 4, 512 << 20))
    dwt = torch.empty(Mp, Co, device=dev)
    def run():
        _lib.check(lib.m3t_conv3d_wgrad_taps(ops._p(x), ops._p(dy), ops._p(dwt), N, Ci, Co, T, H, W, k[0], k[1], k[2], st[0], st[1], st[2],
                                             pd[0], pd[1], pd[2], _lib.M3T_GEMM_F16X3, sl.data_ptr(), sl.data_ptr() + 8, ops._p(ws),
                                             ws.numel() * 4, ops._stream()), "wgrad")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    fl = 2.0 * rows * taps * Ci * Co
    print("%-26s Mp %5d N %4d K %8d tiles %3d splits~%3d : %8.1f us  %6.1f TF/s (useful)" % (name, Mp, Co, rows, tiles, want, us, fl / us / 1e6))
