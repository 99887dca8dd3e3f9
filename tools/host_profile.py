#!/usr/bin/env python3
"""cProfile of the host side of the C3 step (where does the Python time of a step go?)."""
import cProfile, pstats, os, sys, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "m3f.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
from m3t.workloads import AVFeatureGraph, make_c3_step
dev = torch.device("cuda", 0)
torch.manual_seed(12345)
model = AVFeatureGraph(128, 256, 512).to(dev)
batch = bench.synth_batch(32, 300, 128, 256, dev, 0)
ddp, step = make_c3_step(model, batch, max_norm=1.0)
for _ in range(5):
    step()
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
N = 20
# (1) plain host time per step with the GPU kept busy (no sync inside)
t0 = time.perf_counter()
for _ in range(N):
    step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print("host enqueue %.3f ms/step, wall %.3f ms/step" % (th / N * 1e3, tt / N * 1e3))
# (2) host time per step when the GPU is NOT the limit: sync before each step, time only the enqueue
hs = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    hs.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host enqueue with an idle queue: median %.3f ms/step (min %.3f)" % (sorted(hs)[5] * 1e3, min(hs) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])

# (3) the backward functions run on the autograd engine's device thread, which cProfile above does not see: profile them from inside
import inspect
from m3t import ops as _ops
prb = cProfile.Profile()
tb = [0.0, 0]
def _wrap(cls):
    raw = cls.backward
    def backward(ctx, *a):
        t0 = time.perf_counter()
        prb.enable()
        try:
            return raw(ctx, *a)
        finally:
            prb.disable()
            tb[0] += time.perf_counter() - t0; tb[1] += 1
    cls.backward = staticmethod(backward)
for name, obj in list(vars(_ops).items()):
    if inspect.isclass(obj) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
        _wrap(obj)
for _ in range(3):
    step()
torch.cuda.synchronize()
tb[0] = 0.0; tb[1] = 0
prb.clear()
for _ in range(N):
    step()
torch.cuda.synchronize()
print("backward functions: %.3f ms/step in %d calls/step (profiled)" % (tb[0] / N * 1e3, tb[1] // N))
s = io.StringIO()
pstats.Stats(prb, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:6000])
