#!/usr/bin/env python3
"""The tap-walk data-gradient kernel (m3t_conv3d_taps / _pre) on the VGG-M stem's layers at 8 x 64 frames: us and TFLOP/s per layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from m3t import ops, _lib
lib = ops.lib()
dev = "cuda:0"
N, T = 8, 64
layers = [("conv2", 64, 128, 27, 25), ("conv3", 128, 256, 12, 10), ("conv4", 256, 512, 5, 3), ("conv5", 512, 512, 3, 1)]
for name, Ci, Co, H, Ho in layers:
    rows_o, rows_i = N * T * Ho * Ho, N * T * H * H
    dy = torch.randn(rows_o, Co, device=dev)
    wt = torch.randn(Ci, 27 * Co, device=dev) * 0.05
    sl = ops.amax_slots(2, dy.device)
    ops.measure_amax([(dy, sl.data_ptr()), (wt, sl.data_ptr() + 8)])
    dyi, wi = torch.empty_like(dy), torch.empty_like(wt)
    dx = torch.empty(rows_i, Ci, device=dev)
    st = ops._stream()
    ws = ops.workspace(dy.device)
    def pre():
        _lib.check(lib.m3t_f16x3_split(ops._p(dy), rows_o, Co, Co, ops._p(dyi), Co, sl.data_ptr(), st), "s")
        _lib.check(lib.m3t_f16x3_split(ops._p(wt), Ci, 27 * Co, 27 * Co, ops._p(wi), 27 * Co, sl.data_ptr() + 8, st), "s")
    dxp = torch.empty_like(dx)
    def run(planes=None):
        _lib.check(lib.m3t_conv3d_taps_pre(ops._p(dyi), ops._p(wi), ops._p(dx), N, Co, Ci, T, H, H, T, Ho, Ho, 3, 3, 3, 1, 0, 0, -1,
                                           sl.data_ptr(), sl.data_ptr() + 8, ops._p(ws), ws.numel() * 4, planes, st), "t")
    def run_planes():
        run(ops._p(dxp))
    for fn, tag in ((pre, "split"), (run, "taps_pre"), (run_planes, "-> planes")):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        fl = 2.0 * rows_i * Ci * 27 * Co
        print("%-6s %-9s M %7d N %4d K %6d : %8.1f us %s" % (name, tag, rows_i, Ci, 27 * Co, us, ("%6.1f TF/s" % (fl / us / 1e6)) if tag != "split" else ""))
