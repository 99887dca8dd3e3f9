#!/usr/bin/env python3
"""Does a host call BLOCK during a C3 step?  Times every C-ABI call (and torch stream / event calls) on the host; prints the slow ones."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "m3f.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
from m3t import _lib, ops
from m3t.workloads import AVFeatureGraph, make_c3_step
dev = torch.device("cuda", 0)
torch.manual_seed(12345)
model = AVFeatureGraph(128, 256, 512).to(dev)
batch = bench.synth_batch(32, 300, 128, 256, dev, 0)
ddp, step = make_c3_step(model, batch, max_norm=1.0)
for _ in range(6):
    step()
torch.cuda.synchronize()
lib = _lib.load()
log = []
T0 = [0.0]
class Wrap:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); t1 = time.perf_counter()
        log.append((t0 - T0[0], t1 - t0, self.name, threading.current_thread().name)); return r
class LibProxy:
    def __init__(self, lib): self._lib = lib; self._c = {}
    def __getattr__(self, n):
        if n not in self._c: self._c[n] = Wrap(n, getattr(self._lib, n))
        return self._c[n]
proxy = LibProxy(lib)
_lib._lib = proxy
for name in ("wait_stream", "wait_event"):
    orig = getattr(torch.cuda.Stream, name)
    def mk(orig, name):
        def f(self, *a):
            t0 = time.perf_counter(); r = orig(self, *a); t1 = time.perf_counter()
            log.append((t0 - T0[0], t1 - t0, "Stream." + name, threading.current_thread().name)); return r
        return f
    setattr(torch.cuda.Stream, name, mk(orig, name))
orig_rec = torch.cuda.Event.record
def rec(self, *a):
    t0 = time.perf_counter(); r = orig_rec(self, *a); t1 = time.perf_counter()
    log.append((t0 - T0[0], t1 - t0, "Event.record", threading.current_thread().name)); return r
torch.cuda.Event.record = rec
for it in range(4):
    log.clear()
    T0[0] = time.perf_counter()
    step()
    th = time.perf_counter() - T0[0]
tot = sum(l[1] for l in log)
print("last step: host %.3f ms, %d timed calls, %.3f ms inside them" % (th * 1e3, len(log), tot * 1e3))
print("calls slower than 60 us (host time since step start, duration, call, thread):")
for t, d, n, thn in log:
    if d > 60e-6:
        print("  at %7.3f ms  %8.1f us  %-28s %s" % (t * 1e3, d * 1e6, n, thn))
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for t, d, n, thn in log:
    agg[n][0] += 1; agg[n][1] += d
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-28s calls %4d  total %8.3f ms  avg %7.1f us" % (n, c, d * 1e3, d / c * 1e6))
torch.cuda.synchronize()
