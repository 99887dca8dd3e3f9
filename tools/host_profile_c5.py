#!/usr/bin/env python3
"""cProfile of the host side of the C5 step (AffWild2VA on raw frames, 8 x 64): where does the enqueue time go?"""
import cProfile, pstats, os, sys, io, time, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "m3f.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from models.model import AffWild2VA
from m3t.ddp import FlatGradDDP
dev = torch.device("cuda", 0)
rs = np.random.RandomState(0)
f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
hp = AffWild2VA.add_model_specific_args(argparse.ArgumentParser(add_help=False)).parse_args([])
hp.modality, hp.fusion_type, hp.loss, hp.window = "audiovisual", "attention", "ccc_mtl", 64
Bc, Tc = 8, 64
torch.manual_seed(12345)
m = AffWild2VA(hp).to(dev).train()
batch = {"video": f(rs.randint(0, 256, (Bc, 3, Tc, 112, 112)).astype(np.float32)), "se_features": f(rs.standard_normal((Bc, 512, Tc)).astype(np.float32)),
         "audio": f(rs.standard_normal((Bc, Tc, 200)).astype(np.float32)),
         "label_valence": f(rs.uniform(-1, 1, (Bc, Tc)).astype(np.float32)), "label_arousal": f(rs.uniform(-1, 1, (Bc, Tc)).astype(np.float32)),
         "class_expr": f(rs.randint(0, 7, (Bc, Tc)).astype(np.int64)), "expr_valid": f(rs.uniform(size=(Bc, Tc)) < 0.7)}
ddp = FlatGradDDP(m, max_norm=1.0)
def step():
    ddp.zero_grad()
    m.training_step(batch, 0)["loss"].backward()
    ddp.finish()
for _ in range(4):
    step()
torch.cuda.synchronize()
hs = []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); hs.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host enqueue with an idle queue: median %.3f ms" % (sorted(hs)[3] * 1e3))
# forward only vs backward
torch.cuda.synchronize(); t0 = time.perf_counter(); ddp.zero_grad(); l = m.training_step(batch, 0)["loss"]; t1 = time.perf_counter(); l.backward(); t2 = time.perf_counter(); ddp.finish(); t3 = time.perf_counter()
print("forward %.3f ms, backward %.3f ms, finish %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(6):
    step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:5000])

# the backward functions run on the autograd engine's thread: profiled from inside (as tools/host_profile_resnet3d.py)
import inspect
from m3t import ops
prb = cProfile.Profile(); tb = [0.0, 0]
def _wrap(cls):
    raw = cls.backward
    def backward(ctx, *a):
        t0 = time.perf_counter(); prb.enable()
        try:
            return raw(ctx, *a)
        finally:
            prb.disable(); tb[0] += time.perf_counter() - t0; tb[1] += 1
    cls.backward = staticmethod(backward)
for name, obj in list(vars(ops).items()):
    if inspect.isclass(obj) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
        _wrap(obj)
for _ in range(6):
    step()
torch.cuda.synchronize()
print("backward functions: %.3f ms/step in %d calls/step (profiled)" % (tb[0] / 6 * 1e3, tb[1] // 6))
s = io.StringIO(); pstats.Stats(prb, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue()[:6000])

# --- the two halves of the step with the GPU parked behind a spin kernel: what the host needs to enqueue each half (no back-pressure, no
# wait for the loss statistics) and what the GPU needs to run it once everything is queued (HIP events).  training_step's read-back of the
# statistics splits the step in two: time = max(host_fwd, gpu_fwd) + read-back + max(host_bwd, gpu_bwd) when the host is the slower side.
def halves():
    ddp.zero_grad()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda._sleep(int(0.04 * 2.4e9))
    ev[0].record(); t0 = time.perf_counter()
    loss, stats = m.va_objective(m.forward(batch), batch)
    t1 = time.perf_counter(); ev[1].record()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(0.04 * 2.4e9))
    ev[2].record(); t2 = time.perf_counter()
    loss.backward(); ddp.finish()
    t3 = time.perf_counter(); ev[3].record()
    torch.cuda.synchronize()
    return (t1 - t0) * 1e3, ev[0].elapsed_time(ev[1]), (t3 - t2) * 1e3, ev[2].elapsed_time(ev[3])
rows = [halves() for _ in range(5)]
med = lambda i: sorted(r[i] for r in rows)[2]
print("halves (median of 5): host enqueue forward+loss %.3f ms, backward+finish %.3f ms | GPU (events after the spin, whole half queued ahead) "
      "forward+loss %.3f ms, backward+finish %.3f ms" % (med(0), med(2), med(1), med(3)))
