"""Run-to-run variance probe: several timed segments of the C3 step inside ONE process (is the variance per process or in time?).
SV_MODE letters make it progressively more like bench.py: s = set_device first, d = keep the detached loss, m = loop inside a
function, p = events on every 4th step, f = bench's fence, i = bench's import order."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODE = os.environ.get("SV_MODE", "")
if "i" in MODE:
    for _p in (ROOT, os.path.join(ROOT, "m3f.pytorch_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, _p)
else:
    sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd")); sys.path.insert(0, ROOT)
import torch
import bench
if "s" in MODE:
    torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
from m3t.workloads import AVFeatureGraph, make_c3_step
from m3t import ops, _lib


def main():
    torch.manual_seed(12345)
    model = AVFeatureGraph(128, 256, 512).to(dev)
    batch = bench.synth_batch(32, 300, 128, 256, dev, 0)
    ddp, step_fn = make_c3_step(model, batch, max_norm=1.0)

    def step():
        return step_fn()[0].detach() if "d" in MODE else step_fn()

    def fence():
        torch.cuda.synchronize()
        if "f" in MODE:
            torch.cuda.synchronize()

    for _ in range(5):
        step()
    fence()
    ops.PROFILE.clear()
    segs, host = [], []
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
        t0 = time.perf_counter()
        for i in range(L):
            if "p" in MODE:
                ops.PROFILE_ON[0] = i % 4 == 0
            loss = step()
        ops.PROFILE_ON[0] = False
        th = time.perf_counter() - t0
        fence()
        segs.append((time.perf_counter() - t0) / L * 1e3)
        host.append(th / L * 1e3)
    print("mode=%-6s L=%d" % (MODE, L), " ".join("%.2f" % x for x in segs), "| host", " ".join("%.2f" % x for x in host))


main()
