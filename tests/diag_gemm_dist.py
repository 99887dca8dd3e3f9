#!/usr/bin/env python3
"""Diagnostics (not a test): accuracy of m3t_sgemm (default bf16x6 path) against an fp64 product for operand distributions
other than N(0,1): large mean (dB-scale audio), small scale, heavy tails."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "m3f.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from m3t import ops
dev = "cuda:0"
torch.manual_seed(0)
def dist(kind, shape):
    if kind == "normal": return torch.randn(shape, device=dev)
    if kind == "db": return torch.rand(shape, device=dev) * -80.0
    if kind == "small": return torch.randn(shape, device=dev) * 1e-4
    if kind == "cubed": return torch.randn(shape, device=dev) ** 3
    if kind == "sat": return torch.sign(torch.randn(shape, device=dev)) * (1 - 1e-3 * torch.rand(shape, device=dev))
def run(tA, tB, M, N, K, ka, kb):
    A = dist(ka, (K, M) if tA else (M, K)); B = dist(kb, (N, K) if tB else (K, N))
    Cm = torch.empty(M, N, device=dev)
    ops.sgemm(tA, tB, M, N, K, A, 0, A.shape[1], B, 0, B.shape[1], Cm, 0, N)
    ref = (A.double().t() if tA else A.double()) @ (B.double().t() if tB else B.double())
    ref32 = ((A.t() if tA else A) @ (B.t() if tB else B)).double()
    e = float((Cm.double() - ref).norm() / ref.norm()); e32 = float((ref32 - ref).norm() / ref.norm())
    print("tA=%d tB=%d %5dx%5dx%5d A~%-6s B~%-6s relL2: m3t %.2e torch-fp32 %.2e  plan %s" % (tA, tB, M, N, K, ka, kb, e, e32, ops.sgemm_plan(tA, M, N, K)))
for ka, kb in (("normal", "normal"), ("small", "db"), ("normal", "db"), ("db", "normal"), ("small", "sat"), ("cubed", "normal"), ("small", "normal"), ("db", "db")):
    run(1, 0, 768, 128, 9600, ka, kb)      # dW_ih = dgx^T x
    run(0, 1, 9600, 768, 128, ka, kb)      # xproj = x W^T
    run(0, 0, 9600, 512, 768, ka, kb)      # dX = dgx W
