#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one training step vs how long the GPU needs to run it?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from m3t.workloads import AVFeatureGraph
from m3t.ddp import FlatGradDDP
from m3t import ops

dev = torch.device("cuda", 0)
torch.manual_seed(12345)
m = AVFeatureGraph().to(dev)
ddp = FlatGradDDP(m, max_norm=1.0)
b = bench.synth_batch(32, 300, 128, 256, dev, 0)

def step():
    ddp.zero_grad()
    y = m(b["x_a"], b["x_v"])
    loss, _ = ops.va_loss(y, b["valence"], b["arousal"], b["class_expr"], b["expr_valid"], iv=7, ia=8, n_expr=7)
    loss.backward()
    ddp.finish()

for _ in range(3): step()
torch.cuda.synchronize()
for prof in (False, True):
    ops.PROFILE_ON[0] = prof
    ops.PROFILE.clear()
    t0 = time.perf_counter(); enq = 0.0
    for _ in range(5):
        t1 = time.perf_counter(); step(); enq += time.perf_counter() - t1
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("profile=%s: enqueue %.2f ms/step, wall %.2f ms/step" % (prof, enq / 5 * 1e3, tot / 5 * 1e3))

import cProfile, pstats, io
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(5): step()
pr.disable()
torch.cuda.synchronize()
sio = io.StringIO()
pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(14)
print(sio.getvalue()[:3500])
