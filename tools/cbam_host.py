#!/usr/bin/env python3
"""Host cost of one CBAM forward + backward (the 512 x 4 x 4 stage: the GPU needs ~0.2 ms, so the enqueue path shows):
wall time per iteration, host enqueue time per iteration (no sync inside the loop), and a cProfile of the enqueue path."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from models.cbam import CBAM
dev = "cuda:0"
C_, HW, N = (int(v) for v in (sys.argv[1:4] + ["512", "4", "2048"][len(sys.argv) - 1:]))
torch.manual_seed(0)
m = CBAM(C_).to(dev).train()
x = torch.randn(N, C_, HW, HW, device=dev, requires_grad=True)
dy = torch.randn(N, C_, HW, HW, device=dev)
def f():
    for p in m.parameters():
        p.grad = None
    x.grad = None
    m(x).backward(dy)
for _ in range(5): f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): f()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("C=%d %dx%d N=%d: enqueue %.1f us / iteration, wall %.1f us / iteration" % (C_, HW, HW, N, (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
y = m(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): y = m(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("forward only: enqueue %.1f us" % ((t1 - t0) / 50 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(50): f()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime" if os.environ.get("BY_OWN_TIME") == "1" else "cumulative").print_stats(22)

# ---- pieces
from m3t import ops
params = list(m.parameters())
acc = {"none": 0.0, "fwd": 0.0, "bwd": 0.0, "bwd_py": 0.0, "fwd_py": 0.0}
orig_b, orig_f = ops._CBAM.backward, ops._CBAM.forward
def tb(ctx, dy_):
    t = time.perf_counter(); r = orig_b(ctx, dy_); acc["bwd_py"] += time.perf_counter() - t; return r
def tf(ctx, *a):
    t = time.perf_counter(); r = orig_f(ctx, *a); acc["fwd_py"] += time.perf_counter() - t; return r
ops._CBAM.backward, ops._CBAM.forward = staticmethod(tb), staticmethod(tf)
for _ in range(200):
    t = time.perf_counter()
    for p in params: p.grad = None
    x.grad = None
    t1 = time.perf_counter(); y = m(x); t2 = time.perf_counter(); y.backward(dy); t3 = time.perf_counter()
    acc["none"] += t1 - t; acc["fwd"] += t2 - t1; acc["bwd"] += t3 - t2
torch.cuda.synchronize()
print({k: round(v / 200 * 1e6, 1) for k, v in acc.items()}, "us per iteration (fwd_py / bwd_py: inside the Function's python)")
