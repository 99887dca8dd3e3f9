#!/usr/bin/env python3
"""cProfile (by own time) of the enqueue path of one CBAM forward + backward at 512 x 4 x 4."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from models.cbam import CBAM
dev = "cuda:0"
m = CBAM(512).to(dev).train()
x = torch.randn(2048, 512, 4, 4, device=dev, requires_grad=True)
dy = torch.randn(2048, 512, 4, 4, device=dev)
params = list(m.parameters())
def f():
    for p in params: p.grad = None
    x.grad = None
    m(x).backward(dy)
for _ in range(10): f()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200): f()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
