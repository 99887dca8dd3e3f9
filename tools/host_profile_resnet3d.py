#!/usr/bin/env python3
"""cProfile of the host side of the ResNet3D+CBAM step (bench.py's cbam_resnet3d leg, 8 x 64 frames): where does the enqueue time go?
The backward functions run on the autograd engine's thread: they are profiled from inside (as tools/host_profile.py)."""
import cProfile, pstats, os, sys, io, time, inspect
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "m3f.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from models.backbone import VA_3DResNet
from m3t.ddp import FlatGradDDP
from m3t import ops
dev = torch.device("cuda", 0)
rs = np.random.RandomState(0)
f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
Bc, Tc = 8, 64
torch.manual_seed(12345)
net = VA_3DResNet(inputDim=512, hiddenDim=512, nLayers=2, nClasses=2, frameLen=Tc, use_cbam=True, resnet_ver="v1").to(dev).train()
vid = f(((rs.randint(0, 256, (Bc, 3, Tc, 112, 112)).astype(np.float32)) - 127.5) / 127.5)
val, aro = f(rs.uniform(-1, 1, (Bc, Tc)).astype(np.float32)), f(rs.uniform(-1, 1, (Bc, Tc)).astype(np.float32))
ddp = FlatGradDDP(net, max_norm=1.0)
def step():
    ddp.zero_grad()
    y = net(vid)
    ops.va_loss(y, val, aro)[0].backward()
    ddp.finish()
for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6): step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print("host enqueue %.3f ms/step, wall %.3f ms/step" % (th / 6 * 1e3, tt / 6 * 1e3))
hs = []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); hs.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host enqueue with an idle queue: median %.3f ms" % (sorted(hs)[3] * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(6):
    step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(26); print(s.getvalue()[:5000])
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
s = io.StringIO(); pstats.Stats(prb, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
