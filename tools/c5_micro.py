#!/usr/bin/env python3
"""Micro-timings behind two round-3 changes of the C5 step: the first stem layer's weight gradient (old: fp32-MFMA GEMM on the
[64 x 81] problem; new: transposed + padded on the bf16x6 128 x 64 tile) and dW_hh at K = B (T-1) = 504 (old: segmented fp32-MFMA;
new: full-length bf16x6 against shifted states)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from m3t import ops
dev = "cuda:0"
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
# ---- first stem layer: Conv3d(3, 64, 3, stride (1,2,2), pad (1,0,0)) on [8,3,64,112,112]
x = torch.randn(8, 3, 64, 112, 112, device=dev)
w = torch.randn(64, 3, 3, 3, 3, device=dev, requires_grad=True)
y = ops.conv3d(x, w, None, (1, 2, 2), (1, 0, 0))
dy = torch.randn_like(y)
def new():
    w.grad = None
    y.backward(dy, retain_graph=True)
print("conv1 weight gradient, product path (transposed + padded, bf16x6): %.3f ms" % timeit(new))
Co, Kc = 64, 81
dy_cl = dy.permute(0, 2, 3, 4, 1).contiguous(); rows = dy_cl.numel() // Co
def old():
    xp = torch.nn.functional.pad(x, (0, 0, 0, 0, 1, 1))
    pat = xp.unfold(2, 3, 1).unfold(3, 3, 2).unfold(4, 3, 2).permute(0, 2, 3, 4, 1, 5, 6, 7).reshape(rows, Kc)
    dw = torch.empty(Co, Kc, device=dev)
    ops.sgemm(1, 0, Co, Kc, rows, dy_cl, 0, Co, pat, 0, Kc, dw, 0, Kc)
print("conv1 weight gradient, round-2 path (fp32-MFMA GEMM 64 x 81 x %d): %.3f ms" % (rows, timeit(old)))
# ---- dW_hh at B = 8, T = 64, H = 512
B, T, H = 8, 64, 512
dgh = torch.randn(B, T, 3 * H, device=dev); out = torch.randn(B, T, 2 * H, device=dev); dw = torch.empty(3 * H, H, device=dev)
def seg():
    ops.sgemm(1, 0, 3 * H, H, B * (T - 1), dgh, 0, 3 * H, out, 0, 2 * H, dw, 0, H, seg=(T - 1, T, 1, 0))
def full():
    hp = torch.zeros(B, T, H, device=dev); hp[:, 1:].copy_(out[:, :-1, :H])
    ops.sgemm(1, 0, 3 * H, H, B * T, dgh, 0, 3 * H, hp, 0, H, dw, 0, H)
print("dW_hh 1536 x 512, K = 504 segmented (fp32-MFMA): %.1f us; K = 512 against shifted states (bf16x6): %.1f us" % (timeit(seg, 20) * 1e3, timeit(full, 20) * 1e3))
a = dw.clone(); seg(); torch.cuda.synchronize()
print("max |difference| between the two dW_hh: %.2e (scale %.1f)" % (float((a - dw).abs().max()), float(dw.abs().max())))
