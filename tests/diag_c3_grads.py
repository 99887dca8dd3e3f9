#!/usr/bin/env python3
"""Diagnostics (not a test): FULL gradient tensors of the C3 step at 32 x 300 on the GPU against the stock-torch CPU graph
(oracle/torch_ref.py) evaluated in float64 on the same batch and recipe weights.  Prints, per parameter, the relative L2 error
and the largest element error in units of the gradient's rms.   usage: diag_c3_grads.py [normal|db] [B] [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "m3f.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from golden.recipe import fill_module
from m3t.workloads import AVFeatureGraph, make_c3_step
from test_gpu_bench_path import _c3_batch
from oracle import torch_ref as R
audio = sys.argv[1] if len(sys.argv) > 1 else "normal"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
torch.set_num_threads(os.cpu_count() or 8)
ref = fill_module(R.RefAVFeatureGraph(128, 256, 512), 12346).double()
cb = _c3_batch(12345, B, T, 128, 256, device="cpu", audio=audio)
y64 = ref(cb["x_a"].double(), cb["x_v"].double())
R.mtl_loss(y64, cb["valence"].double(), cb["arousal"].double(), cb["class_expr"], cb["expr_valid"]).backward()
model = fill_module(AVFeatureGraph(128, 256, 512), 12346).to("cuda:0")
batch = {k: v.to("cuda:0") for k, v in cb.items()}
ddp, step = make_c3_step(model, batch, max_norm=0.0)
loss, stats, y = step()
torch.cuda.synchronize()
print("audio=%s B=%d T=%d |y - y64| %.3e" % (audio, B, T, float((y.detach().cpu().double() - y64.detach()).abs().max())))
rows = []
rg = dict(ref.named_parameters())
for n, p in model.named_parameters():
    a, b = p.grad.detach().cpu().double().reshape(-1), rg[n].grad.reshape(-1)
    rms = float(b.norm()) / np.sqrt(b.numel())
    d = (a - b).abs()
    rows.append((float((a - b).norm() / b.norm()), float(d.max()) / rms, int(d.argmax()), b.numel(), n))
rows.sort(reverse=True)
for r in rows[:int(os.environ.get("TOP", "40"))]:
    print("relL2 %.2e  max|err|/rms %.2e at %d of %d  %s" % r)
