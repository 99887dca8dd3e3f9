#!/usr/bin/env python3
"""Per-parameter gradient-digest errors of the C3 step at 32 x 300 against the reference golden, sorted (diagnostics).
usage: diag_digests.py [normal|db]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "m3f.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from golden.recipe import fill_module
from m3t.workloads import AVFeatureGraph, make_c3_step
from test_gpu_bench_path import _c3_batch, _digest_err
from conftest import load_golden
audio = sys.argv[1] if len(sys.argv) > 1 else "normal"
g = load_golden("c3_av_graph_b32_db" if audio == "db" else "c3_av_graph_b32")
model = fill_module(AVFeatureGraph(128, 256, 512), 12346).to("cuda:0")
batch = _c3_batch(12345, 32, 300, 128, 256, audio=audio)
ddp, step = make_c3_step(model, batch, max_norm=0.0)
loss, stats, y = step()
torch.cuda.synchronize()
print("audio=%s |y-ref| %.3e loss %.6f ref %.6f" % (audio, float((y.detach().cpu().double() - torch.from_numpy(g["y"]).double()).abs().max()), float(loss), float(g["loss"])))
rows = []
for n, p in model.named_parameters():
    n_rel, h_rel = _digest_err(p.grad, g["gd." + n])
    rows.append((max(n_rel, h_rel), n_rel, h_rel, float(g["gd." + n][0]), n))
rows.sort(reverse=True)
for r in rows[:int(os.environ.get("TOP", "20"))]:
    print("%.2e  norm %.2e head %.2e  |g| %.3e  %s" % r)
