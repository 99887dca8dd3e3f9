"""Opt-in "high" matmul precision (m3t.ops.precision("high"), flag M3T_GEMM_HIGH): every fp32 GEMM / convolution operand is treated
as the sum of two bfloat16 numbers (four bf16 products, fp32 accumulation) -- the meaning of
torch.set_float32_matmul_precision('high').  The default ('highest' = six products, fp32-accurate) is what the benchmark and every
other test use; this file pins what the opt-in mode costs in accuracy, against fp64 and against the reference goldens."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from golden.recipe import fill_module, draw

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M,N,K,tA,tB", [(9600, 512, 1024, 0, 1), (1536, 1024, 9600, 1, 0), (9600, 1024, 1536, 0, 0), (256, 384, 512, 1, 1),
                                         (300, 257, 130, 0, 1)])
def test_sgemm_high_precision_is_16_bit_accurate(M, N, K, tA, tB):
    """interior shapes: relative error of a product ~2^-16 per term (measured against fp64), clearly above the default's and far
    below bf16's; edge shapes (last case) run exact fp32 in this mode"""
    from m3t import ops
    rs = np.random.RandomState(M + N + K)
    A = rs.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
    Bm = rs.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    ref = (A.T if tA else A).astype(np.float64) @ (Bm.T if tB else Bm).astype(np.float64)
    a, b = torch.from_numpy(A).to(DEV), torch.from_numpy(Bm).to(DEV)

    def run(mode):
        out = torch.empty(M, N, device=DEV)
        with ops.precision(mode):
            ops.sgemm(tA, tB, M, N, K, a, 0, a.shape[1], b, 0, b.shape[1], out, 0, N)
        torch.cuda.synchronize()
        return float(np.abs(out.cpu().numpy().astype(np.float64) - ref).max())

    e_hi, e_def, e_bf = run("high"), run("fp32"), run("bf16")
    e_t = float(np.abs((a.t() if tA else a).double().matmul((b.t() if tB else b).double()).cpu().numpy() - ref).max())   # (fp64 on the GPU: ~0)
    e_torch = float(np.abs(((a.t() if tA else a) @ (b.t() if tB else b)).cpu().numpy().astype(np.float64) - ref).max())  # torch's fp32 GEMM
    print("M%d N%d K%d: max abs error vs fp64 -- default %.2e, high %.2e, bf16 %.2e, torch fp32 %.2e" % (M, N, K, e_def, e_hi, e_bf, e_torch))
    assert e_t < 1e-9
    assert e_def <= 3 * e_torch + 1e-6                         # the default is as accurate as an fp32 GEMM
    if M % 128 == 0 and N % 64 == 0 and K % 32 == 0:
        assert 1.5 * e_def < e_hi < e_bf / 20, (e_def, e_hi, e_bf)      # ~2^-16 per term: between the default and bf16 (2^-9)
    else:
        assert e_hi == e_def                                   # not an interior shape: the fp32-MFMA kernel, exact operands


def test_c3_step_high_precision_against_the_reference_golden():
    """the benchmark step (32 clips x 300 frames) in the opt-in mode against the reference-generated golden: outputs still within
    the north_star's 1e-4, gradient norms within 2e-3 (default mode: 2e-4)"""
    from m3t import ops
    from m3t.workloads import AVFeatureGraph, make_c3_step
    from test_gpu_bench_path import _c3_batch
    g = load_golden("c3_av_graph_b32")
    model = fill_module(AVFeatureGraph(128, 256, 512), int(g["seed"]) + 1).to(DEV)
    batch = _c3_batch(int(g["seed"]), 32, 300, 128, 256)
    ddp, step = make_c3_step(model, batch, max_norm=1.0)
    with ops.precision("high"):
        for _ in range(2):
            loss, stats, y = step()
    torch.cuda.synchronize()
    err_y = float((y.detach().cpu().double() - torch.from_numpy(g["y"]).double()).abs().max())
    gn = float(g["grad_norm"])
    print("c3 high: |y - y_ref| = %.2e, loss %.6f vs %.6f, grad norm %.6f vs %.6f" % (err_y, float(loss), float(g["loss"]), float(ddp.last_norm), gn))
    assert err_y <= 1e-4, err_y
    assert abs(float(loss) - float(g["loss"])) <= 1e-4
    assert abs(float(ddp.last_norm) - gn) <= 2e-3 * gn
