#!/usr/bin/env python3
"""Secondary workloads through the drop-in modules: C1 (TCN head), C2 (TCN -> BiGRU), CBAM gates, fwd+bwd."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from m3t.workloads import TcnHead, TcnGru
from models.cbam import CBAM

dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def step(m, x):
    def f():
        for p in m.parameters(): p.grad = None
        y = m(x)
        y.square().mean().backward()
    return f

ONLY_C5 = os.environ.get("ONLY_C5") == "1"


def secondary():
    m = TcnHead(128, 512, 2).to(dev).eval(); x = torch.randn(32, 128, 300, device=dev)
    ms = timeit(step(m, x)); print("C1 TcnHead  B=32 T=300: %.2f ms/step  %.0f clips/s  (%.1f TF/s alg.)" % (ms, 32 / ms * 1e3, 32 * 1.573e9 * 3 / ms / 1e9))
    m = TcnGru(256, 512).to(dev).eval(); x = torch.randn(32, 256, 300, device=dev)
    ms = timeit(step(m, x)); print("C2 TcnGru   B=32 T=300 fp32: %.2f ms/step  %.0f clips/s" % (ms, 32 / ms * 1e3))
    from m3t import ops

    def step_bf16():
        with ops.precision("bf16"):
            step(m, x)()
    ms = timeit(step_bf16); print("C2 TcnGru   B=32 T=300 bf16 operands (BASELINE configs[1]): %.2f ms/step  %.0f clips/s" % (ms, 32 / ms * 1e3))
    x = torch.randn(256, 256, 300, device=dev)
    ms = timeit(step_bf16, 3); print("C2 TcnGru   B=256 T=300 bf16 operands: %.2f ms/step  %.0f clips/s" % (ms, 256 / ms * 1e3))
    for (C, HW, N) in ((64, 28, 2048), (128, 14, 2048), (256, 7, 2048), (512, 4, 2048)):
        m = CBAM(C).to(dev).train(); x = torch.randn(N, C, HW, HW, device=dev, requires_grad=True)
        ms = timeit(step(m, x))
        byts = N * C * HW * HW * 4
        print("CBAM C=%3d %2dx%2d N=%d: %.3f ms fwd+bwd, x = %.1f MB -> %.0f GB/s effective (10 passes over x)" % (C, HW, HW, N, ms, byts / 1e6, 10 * byts / ms / 1e6))



if not ONLY_C5:
    secondary()

# C5: full AffWild2VA audiovisual/attention/v2p_split on raw frames (conv stem on MIOpen), ccc_mtl training step
import argparse
from models.model import AffWild2VA
hp = AffWild2VA.add_model_specific_args(argparse.ArgumentParser(add_help=False)).parse_args([])
hp.modality, hp.fusion_type, hp.loss, hp.window = "audiovisual", "attention", "ccc_mtl", 64
Bc, Tc = 8, 64
m = AffWild2VA(hp).to(dev).train()
batch = {"video": torch.randint(0, 256, (Bc, 3, Tc, 112, 112), device=dev).float(), "se_features": torch.randn(Bc, 512, Tc, device=dev),
         "audio": torch.randn(Bc, Tc, 200, device=dev), "label_valence": torch.rand(Bc, Tc, device=dev) * 2 - 1,
         "label_arousal": torch.rand(Bc, Tc, device=dev) * 2 - 1, "class_expr": torch.randint(0, 7, (Bc, Tc), device=dev),
         "expr_valid": torch.rand(Bc, Tc, device=dev) < 0.7}
def c5():
    for p in m.parameters(): p.grad = None
    out = m.training_step(batch, 0)
    out["loss"].backward()
ms = timeit(c5, 3)
print("C5 (cudnn.benchmark=False) %.1f ms" % ms)
torch.backends.cudnn.benchmark = True
ms = timeit(c5, 3)
print("C5 AffWild2VA AV attention B=%d T=%d 112x112: %.1f ms/step  %.1f clips/s" % (Bc, Tc, ms, Bc / ms * 1e3))
