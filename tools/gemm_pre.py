#!/usr/bin/env python3
"""NT GEMM on operands split once (m3t_sgemm_pre) against the in-kernel split: python tools/gemm_pre.py M N K"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from m3t import ops, _lib
lib = ops.lib()
m, n, k = [int(v) for v in sys.argv[1:4]]
dev = "cuda:0"
A = torch.randn(m, k, device=dev); Bm = torch.randn(n, k, device=dev) * 0.05
C0, C1 = torch.empty(m, n, device=dev), torch.empty(m, n, device=dev)
sl = ops.amax_slots(2, A.device)
ops.measure_amax([(A, sl.data_ptr()), (Bm, sl.data_ptr() + 8)])
Ai, Bi = torch.empty_like(A), torch.empty_like(Bm)
st = ops._stream()
def split():
    _lib.check(lib.m3t_f16x3_split(ops._p(A), m, k, k, ops._p(Ai), k, sl.data_ptr(), st), "s")
    _lib.check(lib.m3t_f16x3_split(ops._p(Bm), n, k, k, ops._p(Bi), k, sl.data_ptr() + 8, st), "s")
def pre():
    _lib.check(lib.m3t_sgemm_pre(m, n, k, ops._p(Ai), k, ops._p(Bi), k, ops._p(C1), n, None, 0, 0, sl.data_ptr(), sl.data_ptr() + 8, st), "p")
def ref():
    ops.sgemm(0, 1, m, n, k, A, 0, k, Bm, 0, k, C0, 0, n, amax=(sl.data_ptr(), sl.data_ptr() + 8))
for fn, tag in ((split, "split A+B"), (ref, "in-kernel split"), (pre, "pre-split")):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print("%-16s %8.1f us  %6.1f TF/s" % (tag, us, 2.0 * m * n * k / us / 1e6))
print("bit-identical:", bool(torch.equal(C0, C1)), " max |diff|", float((C0 - C1).abs().max()))
