#!/usr/bin/env python3
"""Which stock (at::native) kernels are left in a C5 step, and who calls them?  torch.profiler over 3 steps, grouped by aten op and by call stack."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "m3f.pytorch_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
from models.model import AffWild2VA
from m3t.ddp import FlatGradDDP
dev = torch.device("cuda", 0)
rs = np.random.RandomState(0)
f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
Bc, Tc = 8, 64
torch.manual_seed(12345)
if which == "c5":
    hp = AffWild2VA.add_model_specific_args(argparse.ArgumentParser(add_help=False)).parse_args([])
    hp.modality, hp.fusion_type, hp.loss, hp.window = "audiovisual", "attention", "ccc_mtl", 64
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
else:
    from models.backbone import VA_3DResNet
    from m3t import ops
    m = VA_3DResNet(resnet_ver='v1', use_cbam=True, hiddenDim=512, frameLen=Tc, backend='gru', nClasses=2, nFCs=2).to(dev).train()
    x = f(rs.randint(0, 256, (Bc, 3, Tc, 112, 112)).astype(np.float32))
    val, aro = f(rs.uniform(-1, 1, (Bc, Tc)).astype(np.float32)), f(rs.uniform(-1, 1, (Bc, Tc)).astype(np.float32))
    ddp = FlatGradDDP(m, max_norm=1.0)
    def step():
        ddp.zero_grad()
        ops.va_loss(m(x), val, aro)[0].backward()
        ddp.finish()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = [e for e in ka if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.count)
print("aten ops with device time, 3 steps:")
for e in rows[:40]:
    print("  %-40s calls/step %6.1f  device us/step %8.1f" % (e.key, e.count / 3, e.device_time_total / 3))
# who calls them: the Python entry points that launch stock kernels, patched to record their caller (first frame inside the repo);
# what the profiler counts above and this table does not show comes from C++ (AccumulateGrad, materialised zero gradients, autograd's own copies)
import traceback
from collections import Counter
calls = Counter()
def patch(owner, name, only=None):
    orig = getattr(owner, name)
    def wrapped(*a, **k):
        if only is None or only(*a, **k):
            fr = [f for f in traceback.extract_stack()[:-1] if "m3f.pytorch_amd" in f.filename or f.filename.endswith("c5_aten_ops.py")]
            calls[(name, "%s:%d %s" % (os.path.basename(fr[-1].filename), fr[-1].lineno, fr[-1].name) if fr else "?")] += 1
        return orig(*a, **k)
    setattr(owner, name, wrapped)
patch(torch, "zeros"); patch(torch, "cat"); patch(torch, "zeros_like"); patch(torch, "ones_like")
for n in ("copy_", "clone", "add_", "zero_", "fill_", "to", "add", "sub", "div", "mul", "sum", "relu_", "__add__", "__iadd__", "__truediv__", "__mul__", "__sub__"):
    patch(torch.Tensor, n)
patch(torch.Tensor, "contiguous", only=lambda t, *a, **k: not t.is_contiguous())
patch(torch.Tensor, "reshape", only=lambda t, *a, **k: not t.is_contiguous())
for _ in range(2):
    step()
torch.cuda.synchronize()
print("\nPython call sites of stock-kernel entry points, per step:")
for key, c in sorted(calls.items(), key=lambda kv: -kv[1]):
    print("  %-12s %-70s %5.1f" % (key[0], key[1], c / 2))
