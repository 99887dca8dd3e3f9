#!/usr/bin/env python3
"""Spike: what would ONE hipGraph of the whole C3 step buy?  Captures make_c3_step's step (forward + loss + backward +
finalize, four streams) with torch.cuda.graph and times replays against eager steps.  TIMING ONLY: the captured launches carry their
capture-time kernel arguments, so every replay reuses the same exchange tag range (a replay may accept the previous replay's granules)
-- results of replays are NOT checked and must not be trusted; a real implementation needs the tag bases in device memory."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "m3f.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from m3t import ops
from m3t.workloads import AVFeatureGraph, make_c3_step
sys.argv = [sys.argv[0]]
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(12345)
model = AVFeatureGraph(128, 256, 512).to(dev)
batch = bench.synth_batch(32, 300, 128, 256, dev, 0)
ddp, step = make_c3_step(model, batch, max_norm=1.0)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3


for _ in range(5):
    step()
torch.cuda.synchronize()
import gc
gc.collect(); gc.freeze()
e_ms, e_host = timeit(lambda: step())
print("eager: %.3f ms per step (host enqueue %.3f ms)" % (e_ms, e_host), flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        out = step()
except Exception as e:  # noqa: BLE001
    print("CAPTURE FAILED: %r" % (e,), flush=True)
    sys.exit(0)
torch.cuda.synchronize()
r_ms, r_host = timeit(lambda: g.replay())
print("graph replay: %.3f ms per step (host %.3f ms)  [timing only: replays reuse the capture's exchange tags]" % (r_ms, r_host), flush=True)
try:
    ops.poll_scan_error()
    print("no scan error raised during replays")
except Exception as e:  # noqa: BLE001
    print("scan error during replays: %r" % (e,))
