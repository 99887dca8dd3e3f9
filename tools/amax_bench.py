import os, sys, time
sys.path.insert(0, "/root/repo/m3f.pytorch_amd"); sys.path.insert(0, "/root/repo")
import torch
from m3t import ops
from m3t.workloads import AVFeatureGraph
dev = torch.device("cuda:0")
m = AVFeatureGraph(128, 256, 512).to(dev)
ps = list(m.parameters())
ws = [p for p in ps if p.dim() >= 2]
print(len(ws), sum(p.numel() for p in ws) * 4 / 1e6, "MB", sorted(p.numel() for p in ws)[-5:])
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("measure_weight_amax: %.1f us" % t(lambda: ops.measure_weight_amax(ps)))
flat = torch.cat([p.detach().reshape(-1) for p in ws])
print("flat abs().max(): %.1f us" % t(lambda: flat.abs().max()))
sl = ops.amax_slots(1, dev)
f2 = flat[: flat.numel() // 4 * 4].view(-1, 4096) if flat.numel() % 4096 == 0 else flat[: flat.numel() // 4096 * 4096].view(-1, 4096)
print("one region over the flat copy: %.1f us" % t(lambda: ops.measure_amax([(f2, sl.data_ptr())])))
# the kernels without the host: 20 calls queued behind a spin kernel
def tq(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(0.02 * 2.4e9)); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("measure_weight_amax, queued ahead: %.1f us" % tq(lambda: ops.measure_weight_amax(ps)))
