#!/usr/bin/env python3
"""CBAM gates on the ResNet-18 stage shapes (2048 frames): effective HBM bandwidth per pass structure.
Algorithmic traffic of CBAM fwd+bwd = 10 passes over x (channel gate: read x, write y | spatial: read, read+write | backward:
read dy + x, write dx, twice) -- the figure DESIGN.md quotes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from models.cbam import CBAM
dev = "cuda:0"
torch.manual_seed(0)
SHAPES = ((64, 28, 2048), (128, 14, 2048), (256, 7, 2048), (512, 4, 2048))
if os.environ.get("CBAM_SHAPES"):      # "C,HW" : one shape only (profiling)
    c_, hw_ = [int(v) for v in os.environ["CBAM_SHAPES"].split(",")]
    SHAPES = ((c_, hw_, 2048),)
for (C, HW, N) in SHAPES:
    m = CBAM(C).to(dev).train()
    x = torch.randn(N, C, HW, HW, device=dev, requires_grad=True)
    dy = torch.randn(N, C, HW, HW, device=dev)
    params = list(m.parameters())
    def f():
        for q in params: q.grad = None
        x.grad = None
        m(x).backward(dy)                       # CBAM's own kernels only: no loss ops in the timed region
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    byts = N * C * HW * HW * 4
    print("CBAM C=%3d %2dx%2d N=%d: %.3f ms fwd+bwd, x = %.1f MB -> %.0f GB/s at the fused operator's 8 algorithmic passes (%.0f at the two-operator path's 10)"
          % (C, HW, HW, N, ms, byts / 1e6, 8 * byts / ms / 1e6, 10 * byts / ms / 1e6))
