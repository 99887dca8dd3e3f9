#!/usr/bin/env python3
"""Which torch.empty calls of a C3 step are slow, and why (caching-allocator statistics before / after a step)."""
import os, sys, time, threading
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
for _ in range(6):
    step()
torch.cuda.synchronize()
log = []
real_empty = torch.empty
def timed_empty(*a, **k):
    t0 = time.perf_counter()
    r = real_empty(*a, **k)
    log.append((time.perf_counter() - t0, r.numel() * r.element_size(), threading.current_thread().name))
    return r
torch.empty = timed_empty
s0 = torch.cuda.memory_stats()
for _ in range(3):
    log.clear()
    step()
s1 = torch.cuda.memory_stats()
torch.cuda.synchronize()
torch.empty = real_empty
tot = sum(l[0] for l in log)
print("torch.empty calls in the last step: %d, total %.3f ms; by thread:" % (len(log), tot * 1e3), {n: round(sum(l[0] for l in log if l[2] == n) * 1e3, 3) for n in set(l[2] for l in log)})
for d, nb, th in sorted(log, reverse=True)[:15]:
    print("  %8.1f us  %10.2f MB  %s" % (d * 1e6, nb / 1e6, th))
for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams", "allocation.all.allocated", "segment.all.allocated", "reserved_bytes.all.current", "active_bytes.all.peak"):
    print(k, s0.get(k), "->", s1.get(k))
