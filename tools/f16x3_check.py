#!/usr/bin/env python3
"""fp16x3 GEMM mode (M3T_GEMM_F16X3) against the six-product default: error vs fp64 and time, through the C ABI."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from m3t import ops, _lib

dev = "cuda:0"
F16X3 = 1024
HIGH = _lib.M3T_GEMM_HIGH
torch.manual_seed(0)

def run(tA, tB, m, n, k, A, Bm, prec, seg=None, ldb=None, reps=0, hinted=False):
    Cm = torch.empty(m, n, device=dev)
    am = (None, None)
    if hinted:
        sl = ops.amax_slots(2, A.device)
        assert ops.measure_amax([(A, sl.data_ptr()), (Bm, sl.data_ptr() + 8)])
        am = (sl.data_ptr(), sl.data_ptr() + 8)
    def go():
        if seg:
            ops.sgemm(1, 0, m, n, k, A, 0, m, Bm, 0, ldb, Cm, 0, n, seg=seg, prec=prec, amax=am)
        else:
            ops.sgemm(tA, tB, m, n, k, A, 0, A.shape[1], Bm, 0, Bm.shape[1], Cm, 0, n, prec=prec, amax=am)
    go(); torch.cuda.synchronize()
    us = 0.0
    if reps:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): go()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
    return Cm, us

def ref64(tA, tB, A, Bm):
    a = A.double().t() if tA else A.double()
    b = Bm.double().t() if tB else Bm.double()
    return a @ b

def report(name, tA, tB, m, n, k, A, Bm, reps=10):
    R = ref64(tA, tB, A, Bm)
    out = []
    for nm, prec in (("x6", 0), ("high", HIGH), ("f16x3", F16X3), ("f16x3+slots", F16X3)):
        Cm, us = run(tA, tB, m, n, k, A, Bm, prec, reps=reps, hinted=nm.endswith("slots"))
        err = (Cm.double() - R)
        rel = (err.norm() / R.norm()).item()
        rowrel = (err.norm(dim=1) / R.norm(dim=1).clamp_min(1e-300)).max().item()
        out.append("%s %.2e/%.2e %7.1fus" % (nm, rel, rowrel, us))
    print("%-34s %s" % (name, " | ".join(out)), flush=True)

M = 9600
# the shapes of the C3 step
for name, tA, tB, m, n, k in (("fwd proj 9600x1536x1024 NT", 0, 1, M, 1536, 1024), ("dX 9600x1024x1536 NN", 0, 0, M, 1024, 1536),
                              ("dW 1536x1024x9600 TN", 1, 0, 1536, 1024, M), ("fc0 9600x512x1024 NT", 0, 1, M, 512, 1024),
                              ("small 9600x768x128 NT", 0, 1, M, 768, 128), ("2048^3 NN", 0, 0, 2048, 2048, 2048)):
    A = torch.randn((k, m) if tA else (m, k), device=dev)
    Bm = torch.randn((n, k) if tB else (k, n), device=dev)
    report(name, tA, tB, m, n, k, A, Bm)
# operand scales and heterogeneous rows
A = torch.randn(M, 1536, device=dev) * 1e-6 * (10.0 ** torch.empty(M, 1, device=dev).uniform_(-4, 0))
Bm = torch.randn(1536, 1024, device=dev) * 0.03
report("grad-like A (1e-6 x 10^U(-4,0) rows)", 0, 0, M, 1024, 1536, A, Bm, reps=0)
A = torch.randn(M, 1536, device=dev) * (10.0 ** torch.empty(1, 1536, device=dev).uniform_(-9, 0)).t().expand(1536, M).t()
A = torch.randn(M, 1536, device=dev); A[:, ::2] *= 1e-9
A2 = A.t().contiguous()                                   # dead units: half of the OUTPUT rows of dW are 1e-9 x the others
X = torch.randn(M, 1024, device=dev)
report("dW with dead units (1e-9)", 1, 0, 1536, 1024, M, A, X, reps=0)
A = torch.randn(M, 1024, device=dev) * 80.0               # dB-scale inputs
Bm = torch.randn(1536, 1024, device=dev) * 0.05
report("dB-scale x (|x| ~ 80)", 0, 1, M, 1536, 1024, A, Bm, reps=0)
A = torch.zeros(M, 1024, device=dev); Bm = torch.randn(1536, 1024, device=dev)
Cm, _ = run(0, 1, M, 1536, 1024, A, Bm, F16X3)
print("all-zero A -> max |C| =", Cm.abs().max().item())
A = torch.randn(M, 1024, device=dev); A[5, 7] = float("inf")
Cm, _ = run(0, 1, M, 1536, 1024, A, Bm, F16X3)
print("inf in A: row 5 non-finite:", bool((~torch.isfinite(Cm[5])).all().item()), " other rows finite:", bool(torch.isfinite(Cm[6:]).all().item()))
A = torch.randn(M, 1024, device=dev) * 1e30
Cm, _ = run(0, 1, M, 1536, 1024, A, Bm, F16X3)
R = A.double() @ Bm.double().t()
print("huge A (1e30): rel", ((Cm.double() - R).norm() / R.norm()).item())
# segmented dW_hh
H = 512
A = torch.randn(9600, 3 * H, device=dev); Bm = torch.randn(9600, 2 * H, device=dev)
for nm, prec in (("x6", 0), ("f16x3", F16X3)):
    Cm, us = run(1, 0, 3 * H, H, 32 * 299, A, Bm, prec, seg=(299, 300, 1, 0), ldb=2 * H, reps=10)
    a = A.view(32, 300, 3 * H)[:, 1:].reshape(-1, 3 * H).double(); b = Bm.view(32, 300, 2 * H)[:, :-1, :H].reshape(-1, H).double()
    R = a.t() @ b
    print("seg dW_hh %s rel %.2e %7.1f us" % (nm, ((Cm.double() - R).norm() / R.norm()).item(), us))
