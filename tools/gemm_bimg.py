#!/usr/bin/env python3
"""NT product with the weight operand as a staged image fetched by LDS-DMA (m3t_sgemm_bimg) against m3t_sgemm_scaled: time and bit identity.
usage: python tools/gemm_bimg.py M N K [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from m3t import ops, _lib
m, n, k = [int(v) for v in sys.argv[1:4]]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
dev = "cuda:0"
torch.manual_seed(1)
A = torch.randn(m, k, device=dev)
W = torch.randn(n, k, device=dev) * 0.05
bias = torch.randn(n, device=dev)
C0, C1 = torch.empty(m, n, device=dev), torch.empty(m, n, device=dev)
sl = ops.amax_slots(2, A.device)
ops.measure_amax([(A, sl.data_ptr()), (W, sl.data_ptr() + 8)])
img = torch.empty_like(W)
lib = ops.lib()
ws = ops.workspace(A.device)
def image():
    _lib.check(lib.m3t_f16x3_image_b(ops._p(W), n, k, k, ops._p(img), sl.data_ptr() + 8, ops._stream()), "image")
def ref():
    ops.sgemm(0, 1, m, n, k, A, 0, k, W, 0, k, C0, 0, n, bias=bias, amax=(sl.data_ptr(), sl.data_ptr() + 8))
def dma():
    _lib.check(lib.m3t_sgemm_bimg(m, n, k, ops._p(A), k, ops._p(img), ops._p(C1), n, ops._p(bias), 0, 0, ops._p(ws), ws.numel() * 4,
                                  sl.data_ptr(), sl.data_ptr() + 8, ops._stream()), "bimg")
image(); ref(); dma(); torch.cuda.synchronize()
print("bit-identical:", bool(torch.equal(C0, C1)), " max |diff| %.3e" % float((C0 - C1).abs().max()))
r64 = A.double() @ W.double().t() + bias.double()
print("rel.err vs fp64: %.2e (ref %.2e)" % (float((C1.double() - r64).abs().max() / r64.abs().max()), float((C0.double() - r64).abs().max() / r64.abs().max())))
for fn, tag in ((ref, "m3t_sgemm_scaled"), (dma, "m3t_sgemm_bimg"), (image, "image"), (ref, "m3t_sgemm_scaled"), (dma, "m3t_sgemm_bimg")):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("%-18s %4d x %4d x %4d: %7.1f us  %6.1f TF/s" % (tag, m, n, k, us, 2.0 * m * n * k / us / 1e6))
