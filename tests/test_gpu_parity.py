"""GPU parity tests (run with -m gpu on an MI355X).  Every test drives the HIP kernels through
the C ABI (ctypes) and checks them against (a) golden vectors produced by the reference
itself (tests/golden) and (b) the numpy oracle on fresh seeded inputs.
Tolerance: north_star demands VA outputs within 1e-4 (fp32) of the reference."""
import argparse
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, ROOT
from golden.recipe import (fill_module, fill_by_shapes, draw, grad_digest, c3_param_shapes)
from oracle import m3t_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
C5_DIGEST_TOL = 5e-4      # gradient digests of the full A+V model from raw frames (conv stem gradients on MIOpen)
DEV = "cuda:0"


def close(a, b, tol=TOL, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    a, b = a.astype(np.float64), b.astype(np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float(np.abs(a - b).max()) if a.size else 0.0
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    assert err <= tol * scale, "%s: max abs err %.3e (scale %.2f, tol %.1e)" % (what, err, scale, tol)


def dev(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.requires_grad_(True) if grad else t


def load_params(module, g, prefix="p."):
    with torch.no_grad():
        for n, t in list(module.named_parameters()) + list(module.named_buffers()):
            if t.dtype.is_floating_point:
                t.copy_(torch.from_numpy(g[prefix + n]))
    return module


def check_grads(module, g, tol=TOL, prefix="g."):
    names = [k[len(prefix):] for k in g if k.startswith(prefix)]
    got = dict(module.named_parameters())
    assert sorted(names) == sorted(n for n, p in got.items() if p.grad is not None)
    for n in names:
        close(got[n].grad, g[prefix + n], tol, n)


def check_digests(named_grads, g, tol=3e-4, prefix="gd."):
    for n, grad in named_grads:
        ref = g[prefix + n]
        got = grad_digest(grad.detach().cpu().numpy())
        assert abs(got[0] - ref[0]) <= tol * max(1.0, ref[0]), (n, got[0], ref[0])
        scale = max(1.0, float(np.abs(ref[2:]).max()))
        assert float(np.abs(got[2:] - ref[2:]).max()) <= tol * scale, (n, got[2:], ref[2:])


# ------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (33, 9, 17), (128, 128, 16), (300, 257, 130), (1024, 96, 2048), (64, 48, 9600),
                                   (128, 128, 32), (256, 384, 512), (1152, 256, 4096), (9600, 512, 1024),    # these four: bf16x6 path
                                   (128, 1728, 4096), (256, 64, 640), (384, 192, 96)])   # N % 64 == 0 only: bf16x6 on the 128 x 64 tile
@pytest.mark.parametrize("tA,tB", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("mode", ["fp32", "x6"])      # interior shapes: fp16x3 (three products, the default) / bf16x6 (six products)
def test_sgemm(M, N, K, tA, tB, mode):
    from m3t import ops
    with ops.precision(mode):
        _check_sgemm(M, N, K, tA, tB)


def _check_sgemm(M, N, K, tA, tB):
    from m3t import ops
    rs = np.random.RandomState(M * 7 + N * 3 + K + tA * 2 + tB)
    A = rs.standard_normal((K, M) if tA else (M, K)).astype(np.float32)
    B = rs.standard_normal((N, K) if tB else (K, N)).astype(np.float32)
    bias = rs.standard_normal(N).astype(np.float32)
    C0 = rs.standard_normal((M, N)).astype(np.float32)
    ref = (A.T if tA else A).astype(np.float64) @ (B.T if tB else B).astype(np.float64)
    dA, dB, dbias = dev(A), dev(B), dev(bias)
    out = torch.empty(M, N, device=DEV)
    ops.sgemm(tA, tB, M, N, K, dA, 0, A.shape[1], dB, 0, B.shape[1], out, 0, N)
    close(out, ref, 2e-5 * max(1, K ** 0.5 / 8), "plain")
    out2 = dev(C0.copy())
    ops.sgemm(tA, tB, M, N, K, dA, 0, A.shape[1], dB, 0, B.shape[1], out2, 0, N, bias=dbias, act=1, accumulate=True)
    close(out2, np.maximum(ref + bias, 0) + C0, 2e-5 * max(1, K ** 0.5 / 8), "bias+relu+acc")


def test_sgemm_strided_and_segmented():
    """column-sliced operands (lda/ldc > width) and the per-clip segment map used for dW_hh."""
    from m3t import ops
    rs = np.random.RandomState(5)
    B_, T, H = 3, 7, 20
    dgh = rs.standard_normal((B_, T, 3 * H)).astype(np.float32)
    out = rs.standard_normal((B_, T, 2 * H)).astype(np.float32)
    for d, (a_off, b_off) in enumerate([(1, 0), (0, 1)]):
        ref = np.zeros((3 * H, H))
        for b in range(B_):
            for t in range(T):
                tp = t - 1 if d == 0 else t + 1
                if 0 <= tp < T:
                    ref += np.outer(dgh[b, t], out[b, tp, d * H:(d + 1) * H])
        got = torch.empty(3 * H, H, device=DEV)
        ops.sgemm(1, 0, 3 * H, H, B_ * (T - 1), dev(dgh), 0, 3 * H, dev(out), d * H, 2 * H, got, 0, H,
                  seg=(T - 1, T, a_off, b_off))
        close(got, ref, 1e-5, "dW_hh dir %d" % d)
    # round 6: K = clips x (T - 1) need not be a multiple of the 16-bit-term kernels' 32-deep k tile (8 clips x 63 steps = 504: the C5
    # batch) -- the segmented reduction's ragged last tile reads zeros; with and without split-K slabs (whose last chunk is the ragged one)
    B_, T, H = 16, 64, 128          # (K = 1008 = 31.5 k tiles: deep enough for the planner to cut slabs)
    dgh = rs.standard_normal((B_, T, 3 * H)).astype(np.float32)
    out = rs.standard_normal((B_, T, 2 * H)).astype(np.float32)
    assert ops.sgemm_plan(1, 3 * H, H, B_ * (T - 1), seg_len=T - 1, ws_bytes=0)[0] == 1          # the fp16x3 kernel, not the fp32-MFMA fallback
    for d, (a_off, b_off) in enumerate([(1, 0), (0, 1)]):
        hp = out[:, :-1, d * H:(d + 1) * H] if d == 0 else out[:, 1:, d * H:(d + 1) * H]
        g = dgh[:, 1:] if d == 0 else dgh[:, :-1]
        ref = np.einsum("btg,bth->gh", g.astype(np.float64), hp.astype(np.float64))
        for use_ws in (False, True):
            got = torch.full((3 * H, H), float("nan"), device=DEV)
            ops.sgemm(1, 0, 3 * H, H, B_ * (T - 1), dev(dgh), 0, 3 * H, dev(out), d * H, 2 * H, got, 0, H,
                      seg=(T - 1, T, a_off, b_off), use_ws=use_ws)
            close(got, ref, 2e-5 * (1008 ** 0.5 / 8), "ragged dW_hh dir %d ws %d" % (d, use_ws))


def test_sgemm_fp16x3_scales_slots_and_edge_values():
    """The default products (M3T_GEMM_F16X3): two fp16 terms of each SCALED operand.  Against fp64: operands of very different
    magnitudes (gradient-like 1e-7, dB-like 80, 1e30), rows 1e-4 apart inside one operand, caller-provided magnitude slots (exact,
    loose, and the |x| <= 1 constant) give the library-measured result or stay inside the bar; an all-zero operand gives zeros; a
    non-finite value poisons exactly its row; results are identical run to run and no less accurate than the six-product form."""
    from m3t import ops
    M, N, K = 1152, 256, 4096
    rs = np.random.RandomState(7)

    def relerr(out, ref):
        return float(np.linalg.norm(out.cpu().numpy().astype(np.float64) - ref) / np.linalg.norm(ref))

    for sa, sb in ((1.0, 1.0), (1e-7, 0.03), (80.0, 0.05), (1e30, 1e-3)):
        A = (rs.standard_normal((M, K)) * sa).astype(np.float32)
        A *= (10.0 ** rs.uniform(-4, 0, (M, 1))).astype(np.float32)
        B = (rs.standard_normal((N, K)) * sb).astype(np.float32)
        ref = A.astype(np.float64) @ B.astype(np.float64).T
        dA, dB = dev(A), dev(B)
        out = torch.empty(M, N, device=DEV)
        ops.sgemm(0, 1, M, N, K, dA, 0, K, dB, 0, K, out, 0, N)
        e3 = relerr(out, ref)
        rowrel = (np.linalg.norm(out.cpu().numpy() - ref, axis=1) / np.linalg.norm(ref, axis=1)).max()
        with ops.precision("x6"):
            out6 = torch.empty(M, N, device=DEV)
            ops.sgemm(0, 1, M, N, K, dA, 0, K, dB, 0, K, out6, 0, N)
        e6 = relerr(out6, ref)
        assert e3 <= 1e-6 and e3 <= 1.5 * e6 and rowrel <= 2e-6, (sa, sb, e3, e6, rowrel)
        # exact slots reproduce the library's own measurement bit for bit; a bound 2^6 too large stays inside the bar
        sl = ops.amax_slots(4, dA.device)
        assert ops.measure_amax([(dA, sl.data_ptr()), (dB, sl.data_ptr() + 8)])
        out_s = torch.empty(M, N, device=DEV)
        ops.sgemm(0, 1, M, N, K, dA, 0, K, dB, 0, K, out_s, 0, N, amax=(sl.data_ptr(), sl.data_ptr() + 8))
        assert torch.equal(out, out_s)
        amax = torch.tensor([float(np.abs(A).max()) * 64.0, float(np.abs(B).max()) * 64.0], dtype=torch.float32)
        if np.isfinite(amax.numpy()).all():
            sl[2:] = amax.view(torch.int32).to(torch.int64).to(DEV)
            out_l = torch.empty(M, N, device=DEV)
            ops.sgemm(0, 1, M, N, K, dA, 0, K, dB, 0, K, out_l, 0, N, amax=(sl.data_ptr() + 16, sl.data_ptr() + 24))
            assert relerr(out_l, ref) <= 1e-6
        out2 = torch.empty(M, N, device=DEV)
        ops.sgemm(0, 1, M, N, K, dA, 0, K, dB, 0, K, out2, 0, N)
        assert torch.equal(out, out2)
    # |x| <= 1 constant slot (GRU outputs)
    A = np.tanh(rs.standard_normal((M, K))).astype(np.float32)
    B = (rs.standard_normal((N, K)) * 0.05).astype(np.float32)
    out = torch.empty(M, N, device=DEV)
    ops.sgemm(0, 1, M, N, K, dev(A), 0, K, dev(B), 0, K, out, 0, N, amax=(ops.amax_one(out.device), None))
    assert relerr(out, A.astype(np.float64) @ B.astype(np.float64).T) <= 1e-6
    # zeros, and a non-finite value
    Z = torch.zeros(M, K, device=DEV)
    ops.sgemm(0, 1, M, N, K, Z, 0, K, dev(B), 0, K, out, 0, N)
    assert float(out.abs().max()) == 0.0
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    A[5, 7] = np.inf
    ops.sgemm(0, 1, M, N, K, dev(A), 0, K, dev(B), 0, K, out, 0, N)
    assert not bool(torch.isfinite(out[5]).any())
    keep = np.arange(M) != 5
    assert relerr(out[torch.from_numpy(keep).to(DEV)], ref[keep]) <= 1e-6


def test_sgemm_fp16x3_stated_limit_and_the_x6_escape_hatch():
    """VERDICT r3 item 4a / ADVICE r3: the default products are NORMWISE fp32-accurate.  An operand element far below its tensor's
    maximum keeps an ABSOLUTE error (2^-25 / s with s max|A| in [2^14, 2^15): at most 2^-39 max|A|), not a relative one -- so an output
    row built ONLY of such elements loses relative accuracy (error model of the fp16x3 mode).  Rows of A spread over NINE
    decades (1 ... 1e-9 of the maximum), against fp64:
      * every output element stays inside the documented bound  2^-39 (max|A| sum_k |B_jk| + max|B| sum_k |A_ik|) + 2^-21 sum_k |A_ik B_jk|
        (operand representation + the dropped lo x lo term and fp32 accumulation);
      * rows within 2^-16 of the maximum keep fp32-grade ROW-relative accuracy (2e-6);
      * the smallest rows do lose it (that is the stated limit: asserted, so that a future change of the scaling shows up here);
      * ops.precision("x6") -- three bf16 terms, six products, no scaling -- has no such limit: 1e-6 row-relative on EVERY row."""
    from m3t import ops
    M, N, K = 1152, 256, 4096
    rs = np.random.RandomState(11)
    decades = np.linspace(0.0, -9.0, M)
    A = (rs.standard_normal((M, K)) * (10.0 ** decades)[:, None]).astype(np.float32)
    B = (rs.standard_normal((N, K)) * 0.05).astype(np.float32)
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    ref = A64 @ B64.T
    dA, dB = dev(A), dev(B)
    out = torch.empty(M, N, device=DEV)
    ops.sgemm(0, 1, M, N, K, dA, 0, K, dB, 0, K, out, 0, N)
    with ops.precision("x6"):
        out6 = torch.empty(M, N, device=DEV)
        ops.sgemm(0, 1, M, N, K, dA, 0, K, dB, 0, K, out6, 0, N)
    err = np.abs(out.cpu().numpy().astype(np.float64) - ref)
    bound = 2.0 ** -39 * (np.abs(A64).max() * np.abs(B64).sum(1)[None, :] + np.abs(B64).max() * np.abs(A64).sum(1)[:, None]) \
        + 2.0 ** -21 * (np.abs(A64) @ np.abs(B64).T)
    assert bool((err <= bound).all()), "fp16x3 left its documented bound: worst ratio %.2f" % float((err / bound).max())
    rowrel = np.linalg.norm(err, axis=1) / np.linalg.norm(ref, axis=1)
    big = decades >= np.log10(2.0 ** -16)
    assert rowrel[big].max() <= 2e-6, rowrel[big].max()
    assert rowrel[decades <= -8.5].max() > 1e-5, "rows at 1e-9 of the maximum are expected to lose relative accuracy (stated limit)"
    rowrel6 = np.linalg.norm(out6.cpu().numpy().astype(np.float64) - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert rowrel6.max() <= 1e-6, "x6 (the escape hatch) must be row-relative accurate on every row: %.2e" % rowrel6.max()
    print("fp16x3 on rows 1 .. 1e-9 of max|A|: worst err / bound %.2f; row-relative error %.1e (rows >= 2^-16 of the maximum), %.1e (rows at 1e-9); "
          "x6: %.1e on every row" % (float((err / bound).max()), rowrel[big].max(), rowrel[decades <= -8.5].max(), rowrel6.max()))


def test_colsum_transpose():
    from m3t import ops
    rs = np.random.RandomState(6)
    X = rs.standard_normal((9600, 70)).astype(np.float32)
    out = torch.empty(40, device=DEV)
    ops.colsum(dev(X), 10, 9600, 40, 70, out)
    close(out, X[:, 10:50].astype(np.float64).sum(0), 2e-5, "colsum")
    S = rs.standard_normal((37, 65)).astype(np.float32)
    close(ops.transpose2d(dev(S)), S.T, 0, "transpose")


# ------------------------------------------------------------------------------ GRU
@pytest.mark.parametrize("name", ["gru_small", "gru_nofc_h", "gru_fc3", "gru_scorer", "gru_t1"])
def test_gru_golden(name):
    from models.rnn import GRU
    g = load_golden(name)
    a = [int(v) for v in g["args"]]
    (I, H, L, nC), nFC, ret_h = a[:4], (a[4] if len(a) == 6 else 1), a[-1]
    m = load_params(GRU(I, H, L, nC, nFC, return_h=bool(ret_h)), g).to(DEV)
    x = dev(g["x"], True)
    out = m(x)
    if ret_h:
        y, h = out
        close(h, g["h"], TOL, "h_n")
        loss = (y * dev(g["ct"])).sum() + (h * dev(g["ct_h"])).sum()
    else:
        y = out
        loss = (y * dev(g["ct"])).sum()
    close(y, g["y"], TOL, "y")
    loss.backward()
    close(x.grad, g["dx"], TOL, "dx")
    check_grads(m, g)


@pytest.mark.parametrize("B,T,I,H,L", [(5, 33, 40, 48, 2), (32, 20, 64, 128, 1), (37, 6, 30, 20, 2), (2, 50, 12, 256, 1)])
def test_gru_vs_oracle(B, T, I, H, L):
    """fresh seeded inputs, shapes that exercise row/unit masking (B>32, H%16!=0, H%4!=0 is excluded by torch shapes)."""
    from models.rnn import GRU
    rs = np.random.RandomState(B + T + I + H)
    m = fill_module(GRU(I, H, L, 3, 2), 77).to(DEV)
    xn = draw(rs, (B, T, I))
    ct = draw(rs, (B, T, 3))
    p = {n: t.detach().cpu().numpy().astype(np.float64) for n, t in m.named_parameters()}
    y_ref, _, cache = O.gru_module_fwd(xn.astype(np.float64), p, L, 3, 2)
    dx_ref, g_ref = O.gru_module_bwd(ct.astype(np.float64), cache, p, L)
    x = dev(xn, True)
    y = m(x)
    close(y, y_ref, TOL, "y")
    (y * dev(ct)).sum().backward()
    close(x.grad, dx_ref, TOL, "dx")
    for n, prm in m.named_parameters():
        close(prm.grad, g_ref[n], 2e-4, n)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("B,T,I,H,L", [(5, 9, 20, 128, 2), (32, 40, 24, 256, 2), (19, 23, 16, 512, 1), (32, 12, 32, 384, 1),
                                       (40, 6, 16, 128, 1), (1, 2, 8, 128, 1), (33, 5, 12, 256, 1), (48, 4, 16, 512, 1),
                                       (64, 3, 16, 512, 1), (100, 3, 8, 512, 1)])
def test_gru_persistent_scan_equals_per_step(B, T, I, H, L, exact):
    """The persistent scan (one launch for all T steps, W_hh in registers, tagged-granule exchange between CUs)
    must take over for H % 128 == 0 levels.  exact=True (flag M3T_SCAN_FP32: fp32 MFMAs everywhere): it reproduces the
    launch-per-step kernels bit for bit, forward and backward (same MFMA chain and reduction order).  exact=False (the
    default): the recurrent products at H = 256 / 512 run as bf16x6 and H = 128 levels take the solo kernels (one workgroup per
    clip, fp32 FMA chains on the vector ALUs, gru_solo.hip) -- fp32-accurate, equal to the per-step result to fp32 rounding.
    Both must match the oracle."""
    from models.rnn import GRU
    from m3t import ops, _lib
    lib = _lib.load()
    rs = np.random.RandomState(B * 3 + T + H)
    m = fill_module(GRU(I, H, L, 3, 2), 78).to(DEV)
    xn, ct = draw(rs, (B, T, I)), draw(rs, (B, T, 3))

    def run(per_step):
        ops.SCAN_PER_STEP[0], ops.SCAN_FP32[0] = per_step, exact
        try:
            m.zero_grad()
            x = dev(xn, True)
            n0 = lib.m3t_gru_persist_count()
            y = m(x)
            (y * dev(ct)).sum().backward()
            torch.cuda.synchronize()
            launches = lib.m3t_gru_persist_count() - n0
            return y.detach().clone(), x.grad.clone(), {n: prm.grad.clone() for n, prm in m.named_parameters()}, launches
        finally:
            ops.SCAN_PER_STEP[0], ops.SCAN_FP32[0] = False, False

    y1, dx1, g1, n1 = run(False)
    y0, dx0, g0, n0 = run(True)
    assert n0 == 0 and n1 == 2 * L, (n0, n1)            # one persistent launch per layer, forward and backward
    rt2 = 2 * ((B + 15) // 16) * (H // 16) > 256         # 16-row grid too large for the chip: 32-row workgroups, fp32 MFMAs
    if exact or H == 384 or (rt2 and H != 128):          # those always take the fp32-MFMA kernel
        assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
        for n in g0:
            assert torch.equal(g1[n], g0[n]), n
    else:
        close(y1, y0, 3e-6, "y vs per-step")
        close(dx1, dx0, 3e-6, "dx vs per-step")
        for n in g0:
            close(g1[n], g0[n], 1e-5, n + " vs per-step")
        assert not torch.equal(y1, y0)                   # it really is the other arithmetic
    p = {n: t.detach().cpu().numpy().astype(np.float64) for n, t in m.named_parameters()}
    y_ref, _, cache = O.gru_module_fwd(xn.astype(np.float64), p, L, 3, 2)
    dx_ref, g_ref = O.gru_module_bwd(ct.astype(np.float64), cache, p, L)
    close(y1, y_ref, TOL, "y")
    close(dx1, dx_ref, TOL, "dx")
    for n in g1:
        close(g1[n], g_ref[n], 2e-4, n)


@pytest.mark.parametrize("B,T,I,H,L", [(19, 23, 16, 512, 1), (33, 5, 12, 256, 2), (32, 40, 24, 512, 2), (5, 2, 8, 256, 1), (48, 3, 16, 512, 1),
                                       (64, 3, 16, 512, 1)])
def test_gru_wide_scan_kernels_ragged_batches_and_short_clips(B, T, I, H, L):
    """Round 4: the WIDE persistent kernels (16 rows x 32 units per workgroup, memory-order HBM traffic handed over through LDS one step
    late, W_hh fragments built in the forward kernel, 32-deep MFMAs over producer pairs in the backward kernel) on shapes the bench never
    sees: batches that are no multiple of 16 (the last row block is ragged: clamped loads, no stores), T = 2 / 3 (the one-step-late
    hand-over must flush), 1 .. 4 row blocks (XCD slot mapping with 2, 4, 6 and 8 groups).  Forward: forced wide (ops.FORCE_WIDE_FWD);
    backward: every level asks for it by default.  Against the launch-per-step kernels (fp32 MFMAs) to fp32 rounding, and the oracle."""
    from models.rnn import GRU
    from m3t import ops, _lib
    lib = _lib.load()
    rs = np.random.RandomState(B * 5 + T + H)
    m = fill_module(GRU(I, H, L, 3, 2), 79).to(DEV)
    xn, ct = draw(rs, (B, T, I)), draw(rs, (B, T, 3))
    flags = _lib.M3T_GEMM_F16X3 | _lib.M3T_SCAN_WIDE
    nrb = (B + 15) // 16
    assert lib.m3t_gru_scan_workgroups(2, H, B, T, flags, 0) == lib.m3t_gru_scan_workgroups(2, H, B, T, flags, 1) == 2 * nrb * (H // 32)
    assert lib.m3t_gru_scan_workgroups(2, H, B, T, _lib.M3T_GEMM_F16X3, 0) == 2 * nrb * (H // 16)

    def run(per_step):
        ops.SCAN_PER_STEP[0], ops.FORCE_WIDE_FWD[0] = per_step, not per_step
        try:
            m.zero_grad()
            x = dev(xn, True)
            n0 = lib.m3t_gru_persist_count()
            y = m(x)
            (y * dev(ct)).sum().backward()
            torch.cuda.synchronize()
            ops.poll_scan_error()
            return y.detach().clone(), x.grad.clone(), {n: prm.grad.clone() for n, prm in m.named_parameters()}, lib.m3t_gru_persist_count() - n0
        finally:
            ops.SCAN_PER_STEP[0], ops.FORCE_WIDE_FWD[0] = False, False

    y1, dx1, g1, n1 = run(False)
    y0, dx0, g0, n0 = run(True)
    assert n0 == 0 and n1 == 2 * L, (n0, n1)
    close(y1, y0, 3e-6, "y vs per-step")
    close(dx1, dx0, 3e-6, "dx vs per-step")
    for n in g0:
        close(g1[n], g0[n], 1e-5, n + " vs per-step")
    assert not torch.equal(y1, y0)
    p = {n: t.detach().cpu().numpy().astype(np.float64) for n, t in m.named_parameters()}
    y_ref, _, cache = O.gru_module_fwd(xn.astype(np.float64), p, L, 3, 2)
    dx_ref, g_ref = O.gru_module_bwd(ct.astype(np.float64), cache, p, L)
    close(y1, y_ref, TOL, "y")
    close(dx1, dx_ref, TOL, "dx")
    for n in g1:
        close(g1[n], g_ref[n], 2e-4, n)


@pytest.mark.parametrize("exact", [True, False])
def test_gru_persistent_scan_final_state_and_its_gradient(exact):
    """h_n out of the one-launch forward scan and dL/dh_n into the one-launch backward scan (return_h=True, the
    scorer configuration of AttFusion): equal to the launch-per-step path -- bit for bit with M3T_SCAN_FP32 (persistent fp32-MFMA
    kernels), to fp32 rounding by default (H = 128: the solo kernels) -- and to the oracle."""
    from models.rnn import GRU
    from m3t import ops, _lib
    lib = _lib.load()
    rs = np.random.RandomState(44)
    B, T, I, H, L = 21, 11, 12, 128, 2
    m = fill_module(GRU(I, H, L, -1, return_h=True), 79).to(DEV)
    xn, ct, cth = draw(rs, (B, T, I)), draw(rs, (B, T, 2 * H)), draw(rs, (2 * L, B, H))
    res = []
    for per_step in (False, True):
        ops.SCAN_PER_STEP[0], ops.SCAN_FP32[0] = per_step, exact
        try:
            m.zero_grad()
            x = dev(xn, True)
            n0 = lib.m3t_gru_persist_count()
            y, h = m(x)
            ((y * dev(ct)).sum() + (h * dev(cth)).sum()).backward()
            torch.cuda.synchronize()
            res.append((y.detach().clone(), h.detach().clone(), x.grad.clone(), [p.grad.clone() for p in m.parameters()],
                        lib.m3t_gru_persist_count() - n0))
        finally:
            ops.SCAN_PER_STEP[0], ops.SCAN_FP32[0] = False, False
    assert res[0][4] == 2 * L and res[1][4] == 0
    if exact:
        for a, b in zip(res[0][:3], res[1][:3]):
            assert torch.equal(a, b)
        assert all(torch.equal(a, b) for a, b in zip(res[0][3], res[1][3]))
    else:
        for a, b in zip(res[0][:3], res[1][:3]):
            close(a, b, 3e-6, "vs per-step")
        for a, b in zip(res[0][3], res[1][3]):
            close(a, b, 1e-5, "gradient vs per-step")
        assert not torch.equal(res[0][0], res[1][0])
    p = {n: t.detach().cpu().numpy().astype(np.float64) for n, t in m.named_parameters()}
    out_ref, hn_ref, caches = O.bigru_fwd(xn.astype(np.float64), p, L)
    close(res[0][0], out_ref, TOL, "y")
    close(res[0][1], hn_ref, TOL, "h_n")
    dx_ref, g_ref = O.bigru_bwd(ct.astype(np.float64), caches, p, L, dh_n=cth.astype(np.float64))
    close(res[0][2], dx_ref, TOL, "dx")
    for (n, _), gr in zip(m.named_parameters(), res[0][3]):
        close(gr, g_ref[n], 2e-4, n)


def test_scan_exchange_arena_launch_unique_tags_survive_wraps_and_layout_changes():
    """The persistent scans keep their exchange granules in an arena and draw launch-unique tags instead of zeroing the buffers
    before every launch (include/m3t_hip.h, m3t_gru_scan_arena).  Stale granules of earlier launches -- other shapes, other
    layouts in the same sub-arena, 16-bit tag counters that wrap after ~200 launches of 300 steps -- must never be accepted:
    260 forward+backward rounds over alternating shapes stay bit-identical to their first result."""
    from models.rnn import GRU
    from m3t import _lib, ops
    lib = _lib.load()
    torch.manual_seed(11)
    nets = [GRU(16, 256, 1, -1).to(DEV), GRU(24, 512, 1, -1).to(DEV), GRU(8, 128, 1, -1).to(DEV)]
    xs = [torch.randn(32, 300, 16, device=DEV), torch.randn(16, 300, 24, device=DEV), torch.randn(20, 77, 8, device=DEV)]
    first = [None] * 3
    n0 = lib.m3t_gru_persist_count()
    for it in range(260):
        k = it % 3 if it % 7 else 0
        net, x = nets[k], xs[k].clone().requires_grad_(True)
        y = net(x)
        y.square().mean().backward()
        res = (y.detach(), x.grad)
        if first[k] is None:
            first[k] = tuple(t.clone() for t in res)
        else:
            assert torch.equal(res[0], first[k][0]) and torch.equal(res[1], first[k][1]), "round %d, net %d" % (it, k)
    torch.cuda.synchronize()
    assert lib.m3t_gru_persist_count() - n0 == 2 * 260
    ops.poll_scan_error()


@pytest.mark.parametrize("B,T,H", [(20, 77, 128), (32, 77, 128), (32, 31, 256)])
def test_persistent_scan_waves_end_with_nothing_in_flight(B, T, H):
    """Regression: the persistent scans request "the next step's inputs" by inline asm on every step, the last one included; a
    wave that ended with that load outstanding let it land in registers the NEXT wave on the SIMD already owned -- a wild
    address, HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in roughly one run of this loop in two (odd T, H=128, two row blocks).
    Every kernel now drains its vector-memory counter before it ends."""
    from models.rnn import GRU
    torch.manual_seed(3)
    net = GRU(8, H, 1, -1).to(DEV)
    x0 = torch.randn(B, T, 8, device=DEV)
    ref = None
    for it in range(150):
        x = x0.clone().requires_grad_(True)
        y = net(x)
        y.square().mean().backward()
        if ref is None:
            ref = (y.detach().clone(), x.grad.clone())
        elif it % 25 == 0:
            assert torch.equal(y, ref[0]) and torch.equal(x.grad, ref[1])
    torch.cuda.synchronize()


def test_gru_persistent_scan_repeatable_and_long():
    """T = 300 at the C3 width (4 x H=512 scans = 256 workgroups, the whole chip): two runs are bit-identical and no
    wait expires (a later scan call would raise M3T_ESPIN)."""
    from models.rnn import GRU, run_grus
    from m3t import _lib
    lib = _lib.load()
    torch.manual_seed(5)
    a, b = GRU(64, 512, 1, -1).to(DEV), GRU(48, 512, 1, -1).to(DEV)
    xa, xb = torch.randn(32, 300, 64, device=DEV), torch.randn(32, 300, 48, device=DEV)
    n0 = lib.m3t_gru_persist_count()
    ya, yb = run_grus([a, b], [xa, xb])
    ya2, yb2 = run_grus([a, b], [xa, xb])
    torch.cuda.synchronize()
    assert lib.m3t_gru_persist_count() - n0 == 2
    assert torch.equal(ya, ya2) and torch.equal(yb, yb2)
    assert torch.isfinite(ya).all() and float(ya.detach().abs().max()) <= 1.0


def test_grouped_grus_equal_separate():
    """one grouped scan over several modules == module-by-module scans, bit for bit."""
    from models.rnn import GRU, run_grus
    torch.manual_seed(3)
    a, b = GRU(24, 32, 2, -1).to(DEV), GRU(10, 16, 2, 4, 2).to(DEV)
    xa, xb = torch.randn(4, 15, 24, device=DEV), torch.randn(4, 15, 10, device=DEV)
    ya, yb = run_grus([a, b], [xa, xb])
    assert torch.equal(ya, a(xa)) and torch.equal(yb, b(xb))


# ------------------------------------------------------------------------------ TCN
@pytest.mark.parametrize("name", ["tcn_small", "tcn_k2_deep", "tcn_short"])
def test_tcn_golden(name):
    from models.tcn import TemporalConvNet
    g = load_golden(name)
    a = [int(v) for v in g["args"]]
    m = load_params(TemporalConvNet(a[0], a[2:], a[1]), g).to(DEV).eval()
    x = dev(g["x"], True)
    y = m(x)
    close(y, g["y"], TOL, "y")
    (y * dev(g["ct"])).sum().backward()
    close(x.grad, g["dx"], TOL, "dx")
    check_grads(m, g)


@pytest.mark.parametrize("name,which,k,n_out", [("tcn_simple_split_train", "split", 5, 7),
                                                 ("tcn_simple_split_eval", "split", 5, 7),
                                                 ("tcn_simple_vggm_train", "vggm", 3, 2)])
def test_tcn_simple_golden(name, which, k, n_out):
    """`tcn_simple` back-end (Conv1d same-padding + BatchNorm1d + ReLU twice, Linear) against the reference's own
    module in train and eval mode, called the way models/backbone.py:139-141 / 285-286 call it."""
    from models.backbone import VA_3DVGGM, VA_3DVGGM_Split
    from m3t import ops
    g = load_golden(name)
    B, in_dim, T, training = [int(v) for v in g["dims"]]
    if which == "split":
        host = VA_3DVGGM_Split(inputDim=in_dim - 512, hiddenDim=512, backend="tcn_simple", split_layer=3, use_mtl=True,
                               nClasses=8)
        m = host.tcn_v
    else:
        host = VA_3DVGGM(inputDim=in_dim, hiddenDim=512, backend="tcn_simple")
        m = host.tcn
    assert sorted(m.state_dict().keys()) == list(g["state_dict_keys"])
    fill_module(m, int(g["seed"]) + 1)
    m = m.to(DEV).train(bool(training))
    x = dev(g["x"], True)
    y = ops.linear(ops.simple_tcn(ops.bct_to_btc(x), m[0]), m[1].weight, m[1].bias, 0)
    close(y, g["y"], TOL, "y")
    (y * dev(g["ct"])).sum().backward()
    close(x.grad, g["dx"], TOL, "dx")
    for n, b in m.named_buffers():
        if b.dtype.is_floating_point:
            close(b, g["rs." + n], TOL, n)
    assert int(m[0][1].num_batches_tracked) == int(training)
    check_digests([(n, prm.grad) for n, prm in m.named_parameters()], g)


def test_tcn_simple_vs_oracle_full_width():
    """Conv1d(1024,512,5,1,2)+BN+ReLU x2 at B=8 x T=300 (the v2p_split back-end's width) against the numpy oracle."""
    import torch.nn as nn
    from m3t import ops
    rs = np.random.RandomState(21)
    seq = nn.Sequential(nn.Conv1d(1024, 512, 5, 1, 2), nn.BatchNorm1d(512), nn.ReLU(True),
                        nn.Conv1d(512, 512, 5, 1, 2), nn.BatchNorm1d(512), nn.ReLU(True))
    fill_module(seq, 22).to(DEV).train()
    p = {n: t.detach().cpu().numpy().astype(np.float64) for n, t in list(seq.named_parameters()) + list(seq.named_buffers())}
    xn, ct = draw(rs, (8, 1024, 300)), draw(rs, (8, 512, 300))
    y_ref, caches, stats = O.simple_tcn_fwd(xn.astype(np.float64), p, 2, True)
    dx_ref, g_ref = O.simple_tcn_bwd(ct.astype(np.float64), caches, 2)
    x = dev(xn, True)
    y = ops.btc_to_bct(ops.simple_tcn(ops.bct_to_btc(x), seq))
    close(y, y_ref, TOL, "y")
    (y * dev(ct)).sum().backward()
    close(x.grad, dx_ref, 2e-4, "dx")
    for n, prm in seq.named_parameters():
        close(prm.grad, g_ref[n], 3e-4, n)
    for n, v in stats.items():
        close(dict(seq.named_buffers())[n], v, TOL, n)


def test_tcn_vs_oracle_full_width():
    from models.tcn import TemporalConvNet
    rs = np.random.RandomState(11)
    m = fill_module(TemporalConvNet(128, [512, 512], 3), 12).to(DEV).eval()
    xn, ct = draw(rs, (3, 128, 100)), draw(rs, (3, 512, 100))
    p = {n: t.detach().cpu().numpy().astype(np.float64) for n, t in m.named_parameters()}
    y_ref, caches = O.tcn_fwd(xn.astype(np.float64), p, 2)
    dx_ref, g_ref = O.tcn_bwd(ct.astype(np.float64), caches, p)
    x = dev(xn, True)
    y = m(x)
    close(y, y_ref, TOL, "y")
    (y * dev(ct)).sum().backward()
    close(x.grad, dx_ref, TOL, "dx")
    for n, prm in m.named_parameters():
        close(prm.grad, g_ref[n], 2e-4, n)


@pytest.mark.parametrize("K,dil,lead,anti,act", [(3, 1, 0, 0, 1), (3, 2, 0, 0, 2), (5, 1, 2, 0, 0), (3, 2, 0, 1, 0), (5, 1, 2, 1, 0), (2, 4, 0, 0, 1)])
def test_conv1d_on_the_bf16x6_pipe(K, dil, lead, anti, act):
    """interior shapes ((B*T) % 128 == 0, Co % 128 == 0, Ci % 32 == 0) run the dilated convolution as an implicit GEMM on the
    bf16x6 kernel (gemm_x6.hip, CONV): causal / look-ahead / time-flipped taps, clip-boundary masking, fused bias + pre copy +
    ReLU x dropout mask + residual + ReLU -- against a direct fp64 evaluation of models/tcn.py:16-46 on the same operands"""
    from m3t import ops, _lib
    import ctypes as C
    rs = np.random.RandomState(K * 100 + dil * 10 + lead + anti)
    B, T, Ci, Co = 4, 96, 64, 128                       # B*T = 384 = 3 row tiles; clips end inside tiles (96 rows per clip)
    x = rs.standard_normal((B, T, Ci)).astype(np.float32)
    w = (rs.standard_normal((K, Ci, Co) if anti else (K, Co, Ci)) / np.sqrt(K * Ci)).astype(np.float32)
    bias = rs.standard_normal(Co).astype(np.float32) if not anti else None
    res = rs.standard_normal((B, T, Co)).astype(np.float32) if act == 2 or anti else None
    mask = (rs.uniform(size=(B, T, Co)) < 0.8).astype(np.float32) / 0.8 if act else None
    ref = np.zeros((B, T, Co))
    for j in range(K):
        sft = (K - 1 - j) * dil
        off = sft - lead if anti else lead - sft
        for t in range(T):
            if 0 <= t + off < T:
                ref[:, t] += x[:, t + off].astype(np.float64) @ (w[j].astype(np.float64) if anti else w[j].astype(np.float64).T)
    if bias is not None:
        ref += bias
    pre_ref = ref.copy()
    if act == 1:
        ref = np.maximum(ref, 0) * mask
    elif act == 2:
        ref = np.maximum(np.maximum(ref, 0) * mask + res, 0)
    elif res is not None:
        ref = ref + res
    y, pre = torch.empty(B, T, Co, device=DEV), torch.empty(B, T, Co, device=DEV)
    p = lambda a: C.c_void_p(dev(a).data_ptr()) if a is not None else None
    keep = [dev(a) if a is not None else None for a in (x, w, bias, res, mask)]
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    rc = _lib.load().m3t_conv1d_fwd(ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), ptr(keep[4]), ptr(y), ptr(pre), B, T, Ci, Co, K,
                                    dil, lead, act, anti, 0.0, 0, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    close(y, ref, 2e-5, "y")
    close(pre, pre_ref, 2e-5, "pre")


@pytest.mark.parametrize("C_in,width,B,T", [(16, 32, 2, 40), (64, 128, 4, 96)])     # fp32-MFMA conv kernel / bf16x6 implicit GEMM
def test_tcn_train_mode_dropout_matches_oracle(C_in, width, B, T):
    """TRAIN mode with dropout p = 0.2 active: the masks are generated inside the conv epilogues (Philox4x32-10 keyed by two seeds
    per block) and regenerated in backward -- no mask tensor.  The oracle rebuilds the same masks from the seeds (its numpy Philox is
    pinned on the Random123 known answers) and runs models/tcn.py's arithmetic: outputs and every gradient must agree."""
    from models.tcn import TemporalConvNet
    rs = np.random.RandomState(31)
    m = fill_module(TemporalConvNet(C_in, [width, width], 3), 32).to(DEV).train()
    xn, ct = draw(rs, (B, C_in, T)), draw(rs, (B, width, T))
    seeds = [(0x1234567887654321, 0x0fedcba912345678), (3, 2 ** 63 - 5)]
    x = dev(xn, True)
    from m3t import ops
    h = ops.bct_to_btc(x)
    for blk, sd in zip(m.network, seeds):
        h = blk.forward_btc(h, seeds=sd)
    y = ops.btc_to_bct(h)
    p = {n: t.detach().cpu().numpy().astype(np.float64) for n, t in m.named_parameters()}
    # oracle masks: channel-last [B*T, width] -> the oracle's channel-first [B, width, T]
    masks = [tuple(O.dropout_mask(B * T, width, 0.2, s_).reshape(B, T, width).transpose(0, 2, 1) for s_ in sd) for sd in seeds]
    y_ref, caches = O.tcn_fwd(xn.astype(np.float64), p, 2, masks=masks)
    dx_ref, g_ref = O.tcn_bwd(ct.astype(np.float64), caches, p)
    assert float((y_ref == 0).mean()) > 0.05           # the masks really are in the result
    close(y, y_ref, TOL, "y")
    (y * dev(ct)).sum().backward()
    close(x.grad, dx_ref, TOL, "dx")
    for n, prm in m.named_parameters():
        close(prm.grad, g_ref[n], 2e-4, n)
    # fresh seeds from torch's CPU generator on every ordinary call; reproducible under torch.manual_seed
    torch.manual_seed(5); y1 = m(x.detach())
    y2 = m(x.detach())
    torch.manual_seed(5); y3 = m(x.detach())
    assert not torch.equal(y1, y2) and torch.equal(y1, y3)


# ------------------------------------------------------------------------------ AttFusion
@pytest.mark.parametrize("name", ["attfusion_same", "attfusion_proj"])
def test_att_fusion_golden(name):
    from models.att_fusion import AttFusion
    g = load_golden(name)
    a = [int(v) for v in g["args"]]
    m = load_params(AttFusion([a[0], a[1]], a[2]), g).to(DEV)
    xa, xv = dev(g["x_a"], True), dev(g["x_v"], True)
    y = m(xa, xv)
    close(y, g["y"], TOL, "y")
    (y * dev(g["ct"])).sum().backward()
    close(xa.grad, g["dx_a"], TOL, "dx_a")
    close(xv.grad, g["dx_v"], TOL, "dx_v")
    check_grads(m, g)


def test_att_fuse_kernel_vs_oracle_wide():
    from m3t import ops
    rs = np.random.RandomState(21)
    rows, D = 1000, 512
    sv, sa = draw(rs, (rows, 1)), draw(rs, (rows, 1))
    xv, xa, ct = draw(rs, (rows, D)), draw(rs, (rows, D)), draw(rs, (rows, D))
    f_ref, cache = O.att_fuse_core_fwd(*(a.astype(np.float64) for a in (sv, sa, xv, xa)))
    ref = O.att_fuse_core_bwd(ct.astype(np.float64), cache)
    t = [dev(a, True) for a in (sv, sa, xv, xa)]
    f = ops.att_fuse(*t)
    close(f, f_ref, 1e-5, "f")
    (f * dev(ct)).sum().backward()
    for got, want, nm in zip(t, ref, ("ds_v", "ds_a", "dx_v", "dx_a")):
        close(got.grad, want, 1e-5, nm)
    w0 = (f - t[3]).detach() / (t[2] - t[3]).detach()          # convex weights confined to [0.269, 0.731]
    assert float(w0.min()) > 0.26 and float(w0.max()) < 0.74


@pytest.mark.parametrize("num_fcs", [2, 3])
def test_gru_head_train_mode_dropout_matches_oracle(num_fcs):
    """GRU(..., dropout=True) in train mode (reference models/rnn.py:24-28,40-49: Linear -> ReLU -> Dropout(0.5) [-> Linear -> ReLU ->
    Dropout(0.5)] -> Linear): the masks are generated inside the kernels from Philox4x32-10 (no torch RNG kernel, no mask tensor);
    with the seeds pinned, outputs and every gradient must equal the oracle's head run with the restated generator's masks
    (oracle.dropout_mask, pinned on the Random123 known answers).  Also: train != eval, two seeds differ, eval ignores the seeds."""
    from models.rnn import GRU
    I, H, L, nC, B, T = 12, 16, 1, 3, 3, 11
    rs = np.random.RandomState(31 + num_fcs)
    m = fill_module(GRU(I, H, L, nC, num_fcs, dropout=True), 9).to(DEV).train()
    seeds = [0x1234567811223344, 0x0FEDCBA987654321]
    m.drop_seeds = seeds
    xn, ct = draw(rs, (B, T, I)), draw(rs, (B, T, nC))
    x = dev(xn, True)
    y = m(x)
    (y * dev(ct)).sum().backward()
    # oracle: the BiGRU, then the head with explicit masks
    p = {n: t.detach().cpu().numpy().astype(np.float64) for n, t in m.named_parameters()}
    out, _, caches = O.bigru_fwd(xn.astype(np.float64), p, L)
    lin = [k for k in range(0, 3 * num_fcs, 3)]               # fc.0, fc.3, (fc.6): Linear / ReLU / Dropout triples
    h, saved = out.reshape(B * T, 2 * H), []
    for j, k in enumerate(lin):
        w, b = p["fc.%d.weight" % k], p["fc.%d.bias" % k]
        pre = h @ w.T + b
        if j == len(lin) - 1:
            saved.append((h, w, None, None))
            h = pre
        else:
            mask = O.dropout_mask(B * T, w.shape[0], 0.5, seeds[j])
            saved.append((h, w, pre > 0, mask))
            h = np.maximum(pre, 0) * mask
    close(y, h.reshape(B, T, nC), TOL, "y (train mode, pinned seeds)")
    g = ct.astype(np.float64).reshape(B * T, nC)
    grads = {}
    for j in range(len(lin) - 1, -1, -1):
        hin, w, relu, mask = saved[j]
        if mask is not None:
            g = g * mask * relu
        grads["fc.%d.weight" % lin[j]] = g.T @ hin
        grads["fc.%d.bias" % lin[j]] = g.sum(0)
        g = g @ w
    for n, ref in grads.items():
        close(dict(m.named_parameters())[n].grad, ref, TOL, n)
    dx_ref, ggru = O.bigru_bwd(g.reshape(B, T, 2 * H), caches, p, L)
    close(x.grad, dx_ref, TOL, "dx")
    with torch.no_grad():
        m.drop_seeds = [1, 2]
        y2 = m(x)
        y_eval = m.eval()(x)
        m.drop_seeds = seeds
        y_eval2 = m(x)
    assert not torch.equal(y2, y) and not torch.equal(y_eval, y) and torch.equal(y_eval, y_eval2)
    kept = float((m.train().head(torch.ones(B, T, 2 * H, device=DEV)) != 0).float().mean())      # sanity: something survives p = 0.5


# ------------------------------------------------------------------------------ CBAM
@pytest.mark.parametrize("name", ["cbam_train", "cbam_eval", "cbam_c64"])
def test_cbam_golden(name):
    from models.cbam import CBAM
    g = load_golden(name)
    C_ = g["x"].shape[1]
    m = load_params(CBAM(C_), g).to(DEV)
    m.train(bool(g["training"]))
    x = dev(g["x"], True)
    y = m(x)
    close(y, g["y"], TOL, "y")
    close(m.SpatialGate.spatial.bn.running_mean, g["running_mean_after"], 1e-5, "running_mean")
    close(m.SpatialGate.spatial.bn.running_var, g["running_var_after"], 1e-5, "running_var")
    (y * dev(g["ct"])).sum().backward()
    close(x.grad, g["dx"], TOL, "dx")
    check_grads(m, g)


@pytest.mark.parametrize("name", ["cbam_train", "cbam_eval", "cbam_c64"])
def test_cbam_golden_two_operator_path(name):
    """the same goldens through the two separate gates (csrc/cbam.hip: what CBAM.forward uses for maps the fused operator
    does not cover, and what standalone ChannelGate / SpatialGate modules run)"""
    from models.cbam import CBAM
    from m3t import ops
    g = load_golden(name)
    m = load_params(CBAM(g["x"].shape[1]), g).to(DEV)
    m.train(bool(g["training"]))
    x = dev(g["x"], True)
    ops.CBAM_FUSED[0] = False
    try:
        y = m(x)
    finally:
        ops.CBAM_FUSED[0] = True
    close(y, g["y"], TOL, "y")
    (y * dev(g["ct"])).sum().backward()
    close(x.grad, g["dx"], TOL, "dx")
    check_grads(m, g)


@pytest.mark.parametrize("C_,H,W,N,training", [(64, 28, 28, 3, True), (128, 14, 14, 5, True), (256, 7, 7, 6, True), (512, 4, 4, 9, True),
                                              (64, 28, 28, 2, False), (32, 9, 5, 4, True), (16, 1, 1, 7, True), (48, 20, 20, 2, True),
                                              (64, 64, 64, 2, True), (16, 32, 64, 2, True)])
def test_cbam_stage_shapes_against_the_oracle(C_, H, W, N, training):
    """the fused operator on every ResNet-18 stage map (28^2 float4 units over two channel slices, 14^2, the ragged 7^2 with
    single-pixel units, 4^2 with 16 planes per wave pass), odd / degenerate maps, a 64 x 64 map the fused operator does not
    cover (falls to the two gates) and the largest it does (32 x 64: B2's padded maps need more than 64 KB of LDS): forward, running statistics, input and parameter gradients against the numpy oracle"""
    from models.cbam import CBAM
    from m3t import ops
    rs = np.random.RandomState(C_ + H)
    m = fill_module(CBAM(C_), 77).to(DEV)
    m.train(training)
    assert bool(ops.cbam_fused_ok(torch.empty(1, C_, H, W, device=DEV), C_ // 16)) == (H * W <= 2048 if (H * W) % 4 == 0 else H * W <= 512)
    p = {n: t.detach().cpu().numpy().astype(np.float64) for n, t in list(m.named_parameters()) + list(m.named_buffers()) if t.dtype.is_floating_point}
    xn, ct = draw(rs, (N, C_, H, W)), draw(rs, (N, C_, H, W))
    x = dev(xn, True)
    y = m(x)
    y_ref, cache, (rm, rv) = O.cbam_fwd(xn.astype(np.float64), p, training)
    close(y, y_ref, TOL, "y")
    if training:
        close(m.SpatialGate.spatial.bn.running_mean, rm, 1e-5, "running_mean")
        close(m.SpatialGate.spatial.bn.running_var, rv, 1e-5, "running_var")
    (y * dev(ct)).sum().backward()
    dx_ref, grads = O.cbam_bwd(ct.astype(np.float64), cache, p)
    close(x.grad, dx_ref, TOL, "dx")
    for n, q in m.named_parameters():
        close(q.grad, grads[n], 2e-4, n)


def test_cbam_fused_equals_the_two_gates_at_full_size_and_is_deterministic():
    """2048 frames of the first ResNet stage (64 x 28 x 28, the size bench.py's aux.cbam leg times): the fused operator against
    the two-operator path on the same inputs (y, dx, every parameter gradient, running statistics), bit-identical reruns"""
    from models.cbam import CBAM
    from m3t import ops
    torch.manual_seed(5)
    a, b = CBAM(64).to(DEV).train(), CBAM(64).to(DEV).train()
    with torch.no_grad():
        a.SpatialGate.spatial.bn.weight.fill_(0.7)          # (ResNet.__init__ zeroes it: the gradient path through the conv would be dead)
    b.load_state_dict(a.state_dict())
    x1 = torch.randn(2048, 64, 28, 28, device=DEV, requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    ct = torch.randn_like(x1)
    y1 = a(x1)
    y1.backward(ct)
    ops.CBAM_FUSED[0] = False
    try:
        y2 = b(x2)
        y2.backward(ct)
    finally:
        ops.CBAM_FUSED[0] = True
    sc = lambda t: max(1.0, float(t.abs().max()))
    assert float((y1 - y2).abs().max()) <= 1e-5 * sc(y2)
    assert float((x1.grad - x2.grad).abs().max()) <= 1e-5 * sc(x2.grad)
    for (n, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters()):
        assert float((p1.grad - p2.grad).abs().max()) <= 2e-4 * sc(p2.grad), (n, float((p1.grad - p2.grad).abs().max()), sc(p2.grad))
    assert torch.allclose(a.SpatialGate.spatial.bn.running_var, b.SpatialGate.spatial.bn.running_var, rtol=1e-6, atol=0)
    g1 = [p.grad.clone() for p in a.parameters()]
    dx1 = x1.grad.clone()
    a.zero_grad()
    x1.grad = None
    y3 = a(x1)
    y3.backward(ct)
    assert torch.equal(y1, y3) and torch.equal(dx1, x1.grad) and all(torch.equal(u, p.grad) for u, p in zip(g1, a.parameters()))


@pytest.mark.parametrize("C_,H,N", [(256, 7, 70), (512, 4, 33), (32, 3, 5)])
def test_cbam_frame_resident_kernels_are_deterministic_and_cover_ragged_batches(C_, H, N):
    """the frame-resident kernels (csrc/cbam_fused.hip F1L / B2L: 7 x 7 with S = H W, 4 x 4 with the padded stride 17, a 3 x 3 map
    with 9-pixel planes) on batches that are not a multiple of anything: bit-identical reruns, and every frame's result independent
    of its neighbours (frame i of the batch == the same frame in a batch of one, up to the batch statistics: checked in eval mode)"""
    from models.cbam import CBAM
    torch.manual_seed(C_ + H)
    m = CBAM(C_).to(DEV).train()
    with torch.no_grad():
        m.SpatialGate.spatial.bn.weight.fill_(0.8)
    x = torch.randn(N, C_, H, H, device=DEV, requires_grad=True)
    ct = torch.randn_like(x)
    outs = []
    for _ in range(2):
        m.zero_grad()
        x.grad = None
        with torch.no_grad():
            m.SpatialGate.spatial.bn.running_mean.zero_()
            m.SpatialGate.spatial.bn.running_var.fill_(1.0)
        y = m(x)
        y.backward(ct)
        outs.append([y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in m.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    m.eval()
    with torch.no_grad():
        yb = m(x)
        for i in (0, N // 2, N - 1):
            assert torch.equal(m(x[i:i + 1].contiguous()), yb[i:i + 1])


@pytest.mark.parametrize("name", ["resnet_cbam_eval", "resnet_cbam_train"])
def test_resnet_cbam_golden(name):
    from models.resnet import ResNet, BasicBlock
    g = load_golden(name)
    m = fill_module(ResNet(BasicBlock, [1, 1, 1, 1], use_cbam=True), int(g["seed"]) + 1).to(DEV)
    m.train(bool(g["training"]))
    x = dev(g["x"], True)
    y = m(x)
    close(y, g["y"], 2e-4, "y")
    (y * dev(g["ct"])).sum().backward()
    close(x.grad, g["dx"], 5e-4, "dx")
    check_digests([(n, p.grad) for n, p in m.named_parameters() if p.grad is not None], g, tol=1e-3)


# ------------------------------------------------------------------------------ loss
def test_losses_golden():
    from m3t import ops
    g = load_golden("losses")
    y = dev(g["y_hat"], True)
    loss, stats = ops.va_loss(y, dev(g["valence"]), dev(g["arousal"]), dev(g["class_expr"]), dev(g["expr_valid"]),
                              iv=7, ia=8, n_expr=7)
    s = stats.cpu().numpy()
    close(loss, g["loss"], 1e-5, "loss")
    close(s[1], g["loss_v"], 1e-5, "loss_v"); close(s[2], g["loss_a"], 1e-5, "loss_a"); close(s[3], g["loss_expr"], 1e-5, "ce")
    close(s[6], g["ccc_v"], 1e-5, "ccc_v")
    assert int(s[4]) == int(g["expr_valid"].sum())
    loss.backward()
    close(y.grad, g["dy_hat"], 1e-6, "dy_hat")


def test_loss_helper_methods_and_no_valid_rows():
    from models.model import AffWild2VA
    g = load_golden("losses")
    m = AffWild2VA(_hp(modality="audio"))
    yh = dev(g["y_hat"])
    close(m.ccc_loss(yh[..., 7].contiguous(), dev(g["valence"])), g["loss_v"], 1e-5, "ccc_loss")
    close(m.ce_loss(yh[..., :7].contiguous(), dev(g["class_expr"]), dev(g["expr_valid"])), g["loss_expr"], 1e-5, "ce_loss")
    from m3t import ops
    none_valid = torch.zeros_like(dev(g["expr_valid"]))
    loss, stats = ops.va_loss(yh, dev(g["valence"]), dev(g["arousal"]), dev(g["class_expr"]), none_valid, iv=7, ia=8, n_expr=7)
    close(loss, 0.5 * g["loss_v"] + 0.5 * g["loss_a"], 1e-5, "loss without expr term (model.py:173-174)")


@pytest.mark.parametrize("B,T,mode", [(32, 300, "mtl"), (17, 301, "mtl"), (32, 300, "none_valid"), (20, 250, "ccc"), (128, 256, "mtl"), (130, 300, "mtl")])
def test_va_loss_in_one_launch_against_the_oracle(B, T, mode):
    """csrc/fuse_loss.hip, round 6: the loss of AffWild2VA.training_step (reference models/model.py:132-182, models/utils.py:6-17) as ONE grid-wide
    launch -- raw fp64 moments in one sweep, the blocks meet once inside the kernel -- at the bench's 9 600 rows, at a ragged row count (the last
    block half empty), with no valid expression label (the CE term drops out, model.py:173-174), as the plain 'ccc' loss on two outputs, at the
    128-block limit of the fused form and just past it (three launches): loss, parts and the full dL/dy against the numpy oracle; reruns
    bit-identical (fixed-order sums)"""
    from m3t import ops
    from oracle import m3t_oracle as O
    rs = np.random.RandomState(B + T)
    C_ = 2 if mode == "ccc" else 9
    yn = draw(rs, (B, T, C_)) * 0.7
    val, aro = draw(rs, (B, T), "uniform_pm1"), draw(rs, (B, T), "uniform_pm1")
    expr = rs.randint(0, 7, (B, T)).astype(np.int64)
    valid = (rs.uniform(size=(B, T)) < 0.7) & (mode != "none_valid")
    outs = []
    for _ in range(2):
        y = dev(yn, True)
        if mode == "ccc":
            loss, stats = ops.va_loss(y, dev(val), dev(aro))
        else:
            loss, stats = ops.va_loss(y, dev(val), dev(aro), dev(expr), dev(valid), iv=7, ia=8, n_expr=7)
        loss.backward()
        outs.append((loss.detach().clone(), stats.clone(), y.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), "reruns differ"
    l_ref, parts, dy_ref = O.training_loss_fwd_bwd(yn.astype(np.float64), val.astype(np.float64), aro.astype(np.float64),
                                                   expr if mode != "ccc" else None, valid if mode != "ccc" else None, mtl=mode != "ccc")
    loss, stats, dy = outs[0]
    close(loss, l_ref, 1e-5, "loss")
    s = stats.cpu().numpy()
    close(s[1], parts["loss_v"], 1e-5, "loss_v"); close(s[2], parts["loss_a"], 1e-5, "loss_a")
    if "loss_expr" in parts:
        close(s[3], parts["loss_expr"], 1e-5, "loss_expr")
        assert int(s[4]) == int(valid.sum())
    close(dy, dy_ref, 1e-6 if B * T <= 20000 else 3e-7, "dL/dy")


# ------------------------------------------------------------------------------ configs
def _hp(**kw):
    from models.model import AffWild2VA
    ns = AffWild2VA.add_model_specific_args(argparse.ArgumentParser(add_help=False)).parse_args([])
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def _c3_inputs(seed, B, T, d_a, d_v):
    rs = np.random.RandomState(seed)
    xa, xv = draw(rs, (B, T, d_a)), draw(rs, (B, T, d_v))
    val, aro = draw(rs, (B, T), "uniform_pm1"), draw(rs, (B, T), "uniform_pm1")
    expr = rs.randint(0, 7, (B, T)).astype(np.int64)
    valid = rs.uniform(size=(B, T)) < 0.7
    return xa, xv, val, aro, expr, valid


@pytest.mark.parametrize("name", ["c3_av_graph_small", "c3_av_graph"])
def test_c3_graph_golden(name):
    """Config C3/C4 (the bench workload) against the reference's own forward/backward: outputs
    within 1e-4, CCC identical to 3 d.p., gradient digests."""
    from m3t.workloads import AVFeatureGraph
    from m3t import ops
    g = load_golden(name)
    B, T, d_a, d_v, nh = [int(v) for v in g["dims"]]
    seed = int(g["seed"])
    m = fill_module(AVFeatureGraph(d_a, d_v, nh), seed + 1).to(DEV)
    xa, xv, val, aro, expr, valid = _c3_inputs(seed, B, T, d_a, d_v)
    txa, txv = dev(xa, True), dev(xv, True)
    y = m(txa, txv)
    close(y, g["y"], TOL, "y")
    loss, stats = ops.va_loss(y, dev(val), dev(aro), dev(expr), dev(valid), iv=7, ia=8, n_expr=7)
    close(loss, g["loss"], TOL, "loss")
    s = stats.cpu().numpy()
    assert round(float(s[6]), 3) == round(float(g["ccc_v"]), 3)
    assert round(float(s[7]), 3) == round(float(g["ccc_a"]), 3)
    loss.backward()
    close(txa.grad[:, ::25], g["dx_a_full"], TOL, "dx_a")
    close(txv.grad[:, ::25], g["dx_v_full"], TOL, "dx_v")
    check_digests(list((n, p.grad) for n, p in m.named_parameters()), g)


def test_c1_tcn_head_golden():
    from m3t.workloads import TcnHead
    from m3t import ops
    g = load_golden("c1_tcn_head")
    seed = int(g["seed"])
    m = fill_module(TcnHead(128, 512, 2), seed + 1).to(DEV).eval()
    rs = np.random.RandomState(seed)
    x = dev(draw(rs, tuple(int(v) for v in g["in_shape"])), True)
    y = m(x)
    close(y, g["y"], TOL, "y")
    B, T = y.shape[:2]
    val, aro = draw(rs, (B, T), "uniform_pm1"), draw(rs, (B, T), "uniform_pm1")
    loss, _ = ops.va_loss(y, dev(val), dev(aro))
    close(loss, g["loss"], TOL, "loss")
    loss.backward()
    close(x.grad[:, :, ::10], g["dx_full"], TOL, "dx")
    check_digests(list((n, p.grad) for n, p in m.named_parameters()), g)


def test_c2_tcn_gru_golden():
    from m3t.workloads import TcnGru
    from m3t import ops
    g = load_golden("c2_tcn_gru")
    seed = int(g["seed"])
    m = fill_module(TcnGru(256, 512), seed + 1).to(DEV).eval()
    rs = np.random.RandomState(seed)
    x = dev(draw(rs, tuple(int(v) for v in g["in_shape"])), True)
    y = m(x)
    close(y, g["y"], TOL, "y")
    B, T = y.shape[:2]
    val, aro = draw(rs, (B, T), "uniform_pm1"), draw(rs, (B, T), "uniform_pm1")
    loss, _ = ops.va_loss(y, dev(val), dev(aro))
    close(loss, g["loss"], TOL, "loss")
    loss.backward()
    close(x.grad[:, :, ::10], g["dx_full"], TOL, "dx")
    check_digests(list((n, p.grad) for n, p in m.named_parameters()), g)


def _affwild_batch(rs, B, T, video=False):
    batch = {}
    if video:
        batch["video"] = dev(rs.randint(0, 256, (B, 3, T, 112, 112)).astype(np.float32))
        batch["se_features"] = dev(draw(rs, (B, 512, T)))
    batch["audio"] = dev(draw(rs, (B, T, 200)))
    batch["label_valence"] = dev(draw(rs, (B, T), "uniform_pm1"))
    batch["label_arousal"] = dev(draw(rs, (B, T), "uniform_pm1"))
    batch["class_expr"] = dev(rs.randint(0, 7, (B, T)).astype(np.int64))
    batch["expr_valid"] = dev(rs.uniform(size=(B, T)) < 0.7)
    return batch


def test_c1_affwild_audio_training_step_golden():
    from models.model import AffWild2VA
    g = load_golden("c1_affwild_audio")
    seed = int(g["seed"])
    m = fill_module(AffWild2VA(_hp(modality="audio", loss="ccc_mtl")), seed + 1).to(DEV).eval()
    batch = _affwild_batch(np.random.RandomState(seed), 4, 100)
    close(m(batch), g["y"], TOL, "y")
    out = m.training_step(batch, 0)
    close(out["loss"], g["loss"], TOL, "loss")
    close(out["log"]["loss_v"], g["loss_v"], TOL, "loss_v")
    close(out["log"]["loss_expr"], g["loss_expr"], TOL, "loss_expr")
    assert abs(out["progress_bar"]["acc_expr"] - float(g["acc_expr"])) < 1e-6
    out["loss"].backward()
    check_digests(list((n, p.grad) for n, p in m.named_parameters()), g)


def test_training_step_decides_on_the_expression_branch_without_a_stall(monkeypatch):
    """AffWild2VA.training_step (reference models/model.py:166-182): the `if valid_expr > 0` decision from the count that travels to the host
    under the forward pass (round 6, models.model._count_valid_ahead) against the read-back after the loss (M3T_STEP_SYNC=1, the reference's
    two .item() calls): the same keys, bit-equal loss terms and gradients, acc_expr equal to the quotient of the counts; a batch without one
    valid expression label drops loss_expr / acc_expr in both modes and leaves the loss at the valence / arousal terms."""
    import models.model as mm
    from models.model import AffWild2VA
    m = fill_module(AffWild2VA(_hp(modality="audio", loss="ccc_mtl")), 77).to(DEV).train()
    batch = _affwild_batch(np.random.RandomState(5), 4, 100)
    none_valid = dict(batch, expr_valid=torch.zeros_like(batch["expr_valid"]))
    res = {}
    for sync in (False, True):
        monkeypatch.setattr(mm, "_STEP_SYNC", sync)
        for name, b in (("some", batch), ("none", none_valid)):
            torch.manual_seed(3)            # (dropout inside the GRU stack)
            for p_ in m.parameters():
                p_.grad = None
            out = m.training_step(b, 0)
            out["loss"].backward()
            res[sync, name] = (out, [p_.grad.clone() for p_ in m.parameters()])
    for name in ("some", "none"):
        (a, ga), (b_, gb) = res[False, name], res[True, name]
        assert sorted(a["progress_bar"]) == sorted(b_["progress_bar"]) and sorted(a["log"]) == sorted(b_["log"])
        for k in a["log"]:
            assert torch.equal(a["log"][k], b_["log"][k]), k
        assert all(torch.equal(x, y) for x, y in zip(ga, gb))
    some, none = res[False, "some"][0], res[False, "none"][0]
    assert "loss_expr" in some["log"] and "acc_expr" in some["progress_bar"] and torch.is_tensor(some["progress_bar"]["acc_expr"])
    assert isinstance(res[True, "some"][0]["progress_bar"]["acc_expr"], float)
    assert abs(float(some["progress_bar"]["acc_expr"]) - res[True, "some"][0]["progress_bar"]["acc_expr"]) < 1e-6
    assert "loss_expr" not in none["log"] and "acc_expr" not in none["progress_bar"]
    close(none["loss"], 0.5 * none["log"]["loss_v"] + 0.5 * none["log"]["loss_a"], 1e-6, "loss without expression labels")


@pytest.mark.parametrize("Ci,Co,k,stride,pad,N,T,H,W", [
    (3, 64, (3, 3, 3), (1, 2, 2), (1, 0, 0), 2, 6, 13, 12),        # a stem's first layer: C_in k^3 = 81, the transposed / padded GEMM form
    (64, 128, (3, 3, 3), (1, 1, 1), (1, 0, 0), 2, 5, 9, 10),       # interior layers: C_in k^3 = 1728 = 27 x 64
    (3, 64, (5, 7, 7), (1, 2, 2), (2, 3, 3), 1, 7, 20, 18),        # the 3-D ResNet stem: 735 columns, padding on every axis
    (8, 24, (3, 2, 3), (2, 1, 2), (0, 1, 1), 3, 8, 7, 9),          # neither output tile: the plain padded form; odd strides / kernel
])
def test_conv3d_weight_gradient_through_the_patch_matrix(Ci, Co, k, stride, pad, N, T, H, W):
    """m3t.ops.conv3d (reference models/backbone.py:73-103,179-271: the 3-D conv stems): forward and data gradient are torch's, the
    WEIGHT gradient is one fp16x3 GEMM over the patch matrix csrc/conv3d.hip writes (rows = output positions, 64 at a time through an
    LDS tile; ragged row / column tails, zero padding of the convolution on every axis) -- against float64 autograd on the CPU"""
    from m3t import ops
    rs = np.random.RandomState(Ci + Co + H)
    xn, wn, bn_ = draw(rs, (N, Ci, T, H, W)), draw(rs, (Co, Ci) + k) * 0.2, draw(rs, (Co,))
    x, w, b = dev(xn, True), dev(wn, True), dev(bn_, True)
    y = ops.conv3d(x, w, b, stride, pad)
    ctn = draw(rs, tuple(y.shape))
    (y * dev(ctn)).sum().backward()
    x64, w64, b64 = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (xn, wn, bn_))
    y64 = torch.conv3d(x64, w64, b64, stride, pad)
    (y64 * torch.tensor(ctn, dtype=torch.float64)).sum().backward()
    close(y, y64.detach().numpy(), 2e-4, "y")
    close(w.grad, w64.grad.numpy(), 2e-4, "dw")
    close(b.grad, b64.grad.numpy(), 2e-4, "db")
    close(x.grad, x64.grad.numpy(), 2e-4, "dx")


@pytest.mark.parametrize("Ci,Co,k,stride,pad,N,T,H,W", [
    (3, 64, (3, 3, 3), (1, 2, 2), (1, 0, 0), 2, 8, 17, 17),        # VGG-M conv1: 81 columns padded to 128 on both operands, C_out = 64 (the 128 x 64 tile)
    (64, 128, (3, 3, 3), (1, 1, 1), (1, 0, 0), 2, 4, 10, 10),      # interior layers: K = 1728
    (3, 64, (5, 7, 7), (1, 2, 2), (2, 3, 3), 1, 8, 32, 32),        # the 3-D ResNet stem: 735 -> 768 columns, padding on every axis
    (128, 256, (3, 3, 3), (1, 1, 1), (1, 0, 0), 4, 8, 4, 4),       # deep layer: K = 3456 (split-K), small maps
    (64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), 8, 1, 8, 8),         # the per-frame ResNet's 3 x 3 convolution (unit time axis, padding 1): tap-walk data gradient
    (64, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), 2, 4, 8, 8),        # padding on every axis
])
def test_conv3d_forward_on_the_patch_matrix_gemm(Ci, Co, k, stride, pad, N, T, H, W):
    """Round 5: m3t.ops.conv3d's FORWARD is patch matrix x W^T on the fp16x3 GEMM (bias in the epilogue, tiled transpose back to
    [N, Co, T', H', W']), the patch matrix is kept for the weight gradient (reference models/backbone.py:73-103,179-271,327-332) --
    output, weight / bias / input gradients against float64 autograd on the CPU, and against the MIOpen forward (M3T_CONV3D_MIOPEN=1)"""
    from m3t import ops
    rs = np.random.RandomState(Ci + Co + H)
    xn, wn, bn_ = draw(rs, (N, Ci, T, H, W)), draw(rs, (Co, Ci) + k) * 0.2, draw(rs, (Co,))
    x, w, b = dev(xn, True), dev(wn, True), dev(bn_, True)
    assert ops._conv3d_plan(x, w, stride, pad) is not None, "this shape must take the GEMM forward"
    y = ops.conv3d(x, w, b, stride, pad)
    ctn = draw(rs, tuple(y.shape))
    (y * dev(ctn)).sum().backward()
    x64, w64, b64 = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (xn, wn, bn_))
    y64 = torch.conv3d(x64, w64, b64, stride, pad)
    (y64 * torch.tensor(ctn, dtype=torch.float64)).sum().backward()
    close(y, y64.detach().numpy(), 1e-4, "y")
    close(w.grad, w64.grad.numpy(), 2e-4, "dw")
    close(b.grad, b64.grad.numpy(), 2e-4, "db")
    close(x.grad, x64.grad.numpy(), 2e-4, "dx")
    ops.CONV3D_GEMM[0] = False
    try:
        x2, w2, b2 = dev(xn, True), dev(wn, True), dev(bn_, True)
        y2 = ops.conv3d(x2, w2, b2, stride, pad)
        (y2 * dev(ctn)).sum().backward()
    finally:
        ops.CONV3D_GEMM[0] = True
    close(y, y2.detach().cpu().numpy(), 1e-4, "y vs MIOpen")
    close(w.grad, w2.grad.cpu().numpy(), 2e-4, "dw vs the MIOpen-forward path")
    close(x.grad, x2.grad.cpu().numpy(), 2e-4, "dx (stride-1 layers: the tap-walk kernel m3t_conv3d_taps) vs MIOpen's data gradient")


@pytest.mark.parametrize("Ci,Co,k,stride,pad,N,T,H,W", [
    (64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), 8, 1, 8, 8),         # the per-frame ResNet's 3 x 3 convolution
    (32, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), 32, 1, 14, 14),      # W % 4 != 0: the weight gradient's row counter carries inside a thread's four rows
    (64, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), 8, 1, 16, 16),      # the first convolution of a ResNet stage: stride 2 (data gradient: MIOpen)
    (64, 128, (1, 1, 1), (1, 2, 2), (0, 0, 0), 8, 1, 16, 16),      # its 1 x 1 stride-2 shortcut: a strided row gather
    (96, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), 8, 1, 8, 8),         # 864 weight-gradient rows padded to 896; a 128-column tile spans three taps
    (64, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), 2, 4, 8, 8),        # 3-D, padding on every axis
    (128, 256, (3, 3, 3), (1, 1, 1), (1, 0, 0), 4, 8, 4, 4),       # deep layer: split-K in the forward walk too
    (64, 64, (3, 1, 1), (2, 1, 1), (1, 0, 0), 4, 8, 4, 4),         # stride on the time axis
    (3, 64, (5, 7, 7), (1, 2, 2), (2, 3, 3), 1, 8, 32, 32),        # the 3-D ResNet stem: four-channel image, eight-tap rows (m3t_conv3d_fwd_taps4)
    (3, 64, (3, 3, 3), (1, 2, 2), (1, 0, 0), 2, 8, 17, 17),        # VGG-M conv1
    (1, 128, (1, 8, 8), (1, 1, 1), (0, 4, 4), 2, 4, 15, 15),       # one channel, all eight taps used, 128-wide tile
    (64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), 3, 1, 5, 5),         # round 6: 75 rows -- one ragged row tile, a ragged reduction tile in the weight gradient
    (64, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), 3, 1, 15, 15),      # stride 2 on an odd grid, 192 output rows: the parity classes have 8 x 8, 8 x 7, 7 x 8, 7 x 7 positions
    (64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1), 2, 5, 9, 7),        # strides on all three axes (eight parity classes), ragged everywhere
])
def test_conv3d_without_a_patch_matrix(Ci, Co, k, stride, pad, N, T, H, W):
    """Round 5, second half: forward (m3t_conv3d_fwd_taps on operands split once, any stride, bias in the epilogue) and weight gradient
    (m3t_conv3d_wgrad_taps: the reduction over dy's rows, taps picked per output row) as tap walks over the channels-last input kept from
    the forward pass (reference models/backbone.py:73-103,179-271, models/resnet.py:40-45) -- against float64 autograd on the CPU and
    against the patch-matrix path (M3T_CONV3D_IMPLICIT=0); reruns bit-identical (deterministic split-K)"""
    from m3t import ops
    rs = np.random.RandomState(Ci + Co + H)
    xn, wn, bn_ = draw(rs, (N, Ci, T, H, W)), draw(rs, (Co, Ci) + k) * 0.2, draw(rs, (Co,))
    ctn = None
    res = []
    To, Ho, Wo = ((d + 2 * p_ - k_) // s_ + 1 for d, p_, k_, s_ in zip((T, H, W), pad, k, stride))
    whole_tiles = (N * To * Ho * Wo) % 128 == 0            # (the patch-matrix GEMMs still need whole 128-row tiles: no third run otherwise)
    for implicit in ((True, True, False) if whole_tiles else (True, True)):
        saved = ops.CONV3D_IMPLICIT[0]
        ops.CONV3D_IMPLICIT[0] = implicit
        n0 = dict(ops.CONV3D_CALLS)
        try:
            x, w, b = dev(xn, True), dev(wn, True), dev(bn_, True)
            y = ops.conv3d(x, w, b, stride, pad)
            if ctn is None:
                ctn = draw(rs, tuple(y.shape))
            (y * dev(ctn)).sum().backward()
        finally:
            ops.CONV3D_IMPLICIT[0] = saved
        assert ops.CONV3D_CALLS["walk" if implicit else "patch"] == n0["walk" if implicit else "patch"] + 1, "not the path this test means"
        res.append((y.detach(), w.grad, b.grad, x.grad))
    if Ci % 32 == 0:
        # round 6: the weight gradient's walk read the m3t_f16x3_split IMAGES of x and dy (M3T_CONV_IMAGES); splitting the fp32 operands in its
        # loop (M3T_WGRAD_IMAGES=0) gives the same sums bit for bit
        saved = ops.WGRAD_IMAGES[0]
        ops.WGRAD_IMAGES[0] = False
        try:
            x, w, b = dev(xn, True), dev(wn, True), dev(bn_, True)
            (ops.conv3d(x, w, b, stride, pad) * dev(ctn)).sum().backward()
        finally:
            ops.WGRAD_IMAGES[0] = saved
        assert torch.equal(res[0][1], w.grad), "weight gradient: operand images vs operands split in the loop"
        assert torch.equal(res[0][3], x.grad)
    x64, w64, b64 = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (xn, wn, bn_))
    y64 = torch.conv3d(x64, w64, b64, stride, pad)
    (y64 * torch.tensor(ctn, dtype=torch.float64)).sum().backward()
    close(res[0][0], y64.detach().numpy(), 1e-4, "y")
    close(res[0][1], w64.grad.numpy(), 2e-4, "dw")
    close(res[0][2], b64.grad.numpy(), 2e-4, "db")
    close(res[0][3], x64.grad.numpy(), 2e-4, "dx")
    for a, b_, what in zip(res[0], res[1], ("y", "dw", "db", "dx")):
        if what != "dx" or tuple(stride) == (1, 1, 1) or (Ci % 64 == 0 and Co % 32 == 0):      # (strided layers: parity-class walks since round 6, else MIOpen's)
            assert torch.equal(a, b_), "reruns differ: " + what
    if whole_tiles:
        for a, b_, what in zip(res[0], res[2], ("y", "dw", "db", "dx")):
            close(a, b_.cpu().numpy(), 2e-4, what + " vs the patch-matrix path")


def test_planes_to_image_in_one_pass():
    """Round 6, the per-frame ResNet's planes path (reference models/resnet.py:40-56: conv -> BatchNorm2d -> ReLU -> conv ...): BatchNorm's apply
    and dx kernels and the residual add + ReLU raise the magnitude slot of what they write, and the convolution behind (in front, in backward)
    turns those planes into the channels-last fp16x3 IMAGE in one pass (m3t_bct_to_btc_img) instead of transpose + measure, then split.  The
    kernel against the two-pass route, and a conv -> BN+ReLU -> conv -> BN -> add+ReLU -> conv pipeline with the switch on and off: bit-equal."""
    from m3t import ops, _lib
    import ctypes as C
    lib = ops.lib()
    rs = np.random.RandomState(3)
    B, Cc, S = 3, 64, 50                                      # (S is no multiple of 32: ragged tiles)
    src = dev(draw(rs, (B, Cc, S)) * 3.0)
    rows = torch.empty(B * S, Cc, device=DEV)
    slot = ops.amax_slots(1, src.device)
    ops.amax_out(slot.data_ptr())
    _lib.check(lib.m3t_bct_to_btc(ops._p(src), ops._p(rows), B, Cc, S, ops._stream()), "m3t_bct_to_btc")
    img2 = torch.empty_like(rows)
    _lib.check(lib.m3t_f16x3_split(ops._p(rows), B * S, Cc, Cc, ops._p(img2), Cc, slot.data_ptr(), ops._stream()), "m3t_f16x3_split")
    img1 = torch.empty_like(rows)
    part = torch.empty(B * ((S + 31) // 32), Cc, device=DEV)
    _lib.check(lib.m3t_bct_to_btc_img(ops._p(src), ops._p(img1), B, Cc, S, slot.data_ptr(), ops._p(part), ops._stream()), "m3t_bct_to_btc_img")
    assert torch.equal(img1.view(torch.int32), img2.view(torch.int32))
    close(part.sum(0), src.sum((0, 2)).cpu().numpy(), 1e-5, "channel sums riding along")
    # the pipeline
    xn, w1n, w2n, w3n = draw(rs, (6, 64, 9, 9)), draw(rs, (64, 64, 3, 3)) * 0.1, draw(rs, (64, 64, 3, 3)) * 0.1, draw(rs, (128, 64, 3, 3)) * 0.1
    gam, bet = 1.0 + 0.1 * draw(rs, (64,)), 0.1 * draw(rs, (64,))
    res = []
    saved = ops.TRANSPOSE_IMAGES[0]
    try:
        for on in (True, False):
            ops.TRANSPOSE_IMAGES[0] = on
            x, w1, w2, w3 = dev(xn, True), torch.nn.Parameter(dev(w1n)), torch.nn.Parameter(dev(w2n)), torch.nn.Parameter(dev(w3n))
            g1, b1, g2, b2 = (dev(a, True) for a in (gam, bet, gam, bet))
            rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
            n0 = ops.CONV3D_CALLS["walk"]
            h = ops.bn_planes(ops.conv2d(x, w1, None, (1, 1), (1, 1)), g1, b1, rm.clone(), rv.clone(), True, 0.1, 1e-5, True)
            assert (ops._tagged_amax(h) is not None) == on
            h = ops.bn_planes(ops.conv2d(h, w2, None, (1, 1), (1, 1)), g2, b2, rm.clone(), rv.clone(), True, 0.1, 1e-5, False)
            h = ops.add_relu(h, x)
            y = ops.conv2d(h, w3, None, (2, 2), (1, 1))
            assert ops.CONV3D_CALLS["walk"] == n0 + 3
            (y * y).sum().backward()
            res.append([y.detach(), x.grad, w1.grad, w2.grad, w3.grad, g1.grad, b1.grad, g2.grad, b2.grad])
    finally:
        ops.TRANSPOSE_IMAGES[0] = saved
    for a, b_ in zip(*res):
        assert torch.equal(a, b_)
    x64 = torch.tensor(xn, dtype=torch.float64, requires_grad=True)
    ws64 = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (w1n, w2n, w3n)]
    g64, b64 = torch.tensor(gam, dtype=torch.float64), torch.tensor(bet, dtype=torch.float64)
    F = torch.nn.functional
    h = torch.relu(F.batch_norm(F.conv2d(x64, ws64[0], None, 1, 1), None, None, g64, b64, True))
    h = torch.relu(F.batch_norm(F.conv2d(h, ws64[1], None, 1, 1), None, None, g64, b64, True) + x64)
    y64 = F.conv2d(h, ws64[2], None, 2, 1)
    (y64 * y64).sum().backward()
    close(res[0][0], y64.detach().numpy(), 2e-4, "y")
    close(res[0][1], x64.grad.numpy(), 5e-4, "dx")
    close(res[0][4], ws64[2].grad.numpy(), 5e-4, "dw3")
    close(res[0][2], ws64[0].grad.numpy(), 5e-4, "dw1")


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("N,C_,T,H,W,training", [(3, 64, 2, 7, 9, True), (2, 128, 4, 12, 12, True), (2, 512, 3, 1, 1, False), (1, 256, 1, 5, 5, True),
                                                 (2, 64, 3, 8, 6, False)])
def test_channels_last_stem_operators(N, C_, T, H, W, training, fused):
    """csrc/stem_cl.hip (round 6): BatchNorm3d + ReLU and MaxPool3d((1, k, k)) of the 3-D stems (reference models/backbone.py:73-103,179-191) on
    channels-last rows [N T H W][C], and the convolution between them as a tap walk that reads and writes that layout (ops.conv3d_cl) --
    outputs, running statistics, input and parameter gradients against float64 torch on the CPU; the kernels' magnitude slots cover what
    they wrote"""
    from m3t import ops
    rs = np.random.RandomState(N * 100 + C_ + H)
    xn = draw(rs, (N, C_, T, H, W)) * 2.0 + 0.3
    gam, bet = 1.0 + 0.1 * draw(rs, (C_,)), 0.1 * draw(rs, (C_,))
    rm, rv = 0.1 * draw(rs, (C_,)), np.abs(draw(rs, (C_,))) * 0.5 + 0.5
    to_cl = lambda a: np.ascontiguousarray(np.transpose(a, (0, 2, 3, 4, 1))).reshape(-1, a.shape[1])
    x = dev(to_cl(xn), True)
    g_, b_ = dev(gam, True), dev(bet, True)
    rmd, rvd = dev(rm.copy()), dev(rv.copy())
    xc = ops.CLTensor(x, N, T, H, W, None)
    geo = [((2, 2), (2, 2), (0, 0)), ((3, 3), (2, 2), (1, 1))][(H + W) % 2] if min(H, W) >= 3 else None
    if fused and (geo is None or geo[0] != (2, 2)):
        pytest.skip("the fused BatchNorm + ReLU + pooling operator needs tiling windows")
    # fused: BatchNorm + ReLU stay PENDING on the tensor and the pooling applies them in its window loop (m3t_bn_pool_cl_*); odd H / W: the
    # last row / column lies in no window
    y = ops.bn_cl(xc, g_, b_, rmd, rvd, training, 0.1, 1e-5, True, lazy=fused)
    z = ops.pool_cl(y, *geo) if geo else y
    if fused:
        assert y._pending is not None, "the pooling did not take the fused operator"
    ct = draw(rs, tuple(z.data.shape))
    (z.data * dev(ct)).sum().backward()
    # magnitude slots
    torch.cuda.synchronize()
    if not fused:
        assert int(y.slot.item()) & 0xffffffff == int(torch.tensor([float(y.data.abs().max())]).view(torch.int32).item())
    if geo:
        assert int(z.slot.item()) & 0xffffffff == int(torch.tensor([float(z.data.abs().max())]).view(torch.int32).item())
    # float64 torch
    x64 = torch.tensor(xn, dtype=torch.float64, requires_grad=True)
    g64, b64 = torch.tensor(gam, dtype=torch.float64, requires_grad=True), torch.tensor(bet, dtype=torch.float64, requires_grad=True)
    rm64, rv64 = torch.tensor(rm, dtype=torch.float64), torch.tensor(rv, dtype=torch.float64)
    y64 = torch.relu(torch.nn.functional.batch_norm(x64, rm64, rv64, g64, b64, training, 0.1, 1e-5))
    z64 = torch.nn.functional.max_pool3d(y64, (1,) + geo[0], (1,) + geo[1], (0,) + geo[2]) if geo else y64
    ct64 = torch.tensor(ct, dtype=torch.float64).view(z64.shape[0], z64.shape[2], z64.shape[3], z64.shape[4], C_).permute(0, 4, 1, 2, 3)
    (z64 * ct64).sum().backward()
    if not fused:
        close(y.data, to_cl(y64.detach().numpy()), 2e-5, "bn + relu")
    close(z.data, to_cl(z64.detach().numpy()), 2e-5, "pool")
    close(x.grad, to_cl(x64.grad.numpy()), 1e-4, "dx")
    close(g_.grad, g64.grad.numpy(), 2e-4, "dgamma")
    close(b_.grad, b64.grad.numpy(), 2e-4, "dbeta")
    if training:
        close(rmd, rm64.numpy(), 1e-5, "running_mean")
        close(rvd, rv64.numpy(), 1e-5, "running_var")


@pytest.mark.parametrize("Ci,Co,k,stride,pad,N,T,H,W", [(64, 128, (3, 3, 3), (1, 1, 1), (1, 0, 0), 2, 4, 9, 9), (3, 64, (3, 3, 3), (1, 2, 2), (1, 0, 0), 2, 3, 17, 17),
                                                       (128, 64, (1, 3, 3), (1, 2, 2), (0, 1, 1), 3, 1, 7, 7)])
def test_conv3d_on_channels_last_rows(Ci, Co, k, stride, pad, N, T, H, W):
    """ops.conv3d_cl: the stems' convolutions reading and writing channels-last rows (no transpose on either side): forward, weight / bias
    gradient and the data gradient (strided: parity-class walks scattered into the channels-last grid) against float64 autograd; a first
    layer takes the video planes"""
    from m3t import ops
    rs = np.random.RandomState(Ci + Co + H)
    xn, wn, bn_ = draw(rs, (N, Ci, T, H, W)), draw(rs, (Co, Ci) + k) * 0.2, draw(rs, (Co,))
    to_cl = lambda a: np.ascontiguousarray(np.transpose(a, (0, 2, 3, 4, 1))).reshape(-1, a.shape[1])
    first = Ci <= 4
    w, b = dev(wn, True), dev(bn_, True)
    if first:
        x = dev(xn)
        y = ops.conv3d_cl(x, w, b, stride, pad)
    else:
        x = dev(to_cl(xn), True)
        y = ops.conv3d_cl(ops.CLTensor(x, N, T, H, W, None), w, b, stride, pad)
    ct = draw(rs, tuple(y.data.shape))
    (y.data * dev(ct)).sum().backward()
    x64, w64, b64 = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (xn, wn, bn_))
    y64 = torch.conv3d(x64, w64, b64, stride, pad)
    assert (y.N, y.T, y.H, y.W) == (y64.shape[0], y64.shape[2], y64.shape[3], y64.shape[4])
    ct64 = torch.tensor(ct, dtype=torch.float64).view(y.N, y.T, y.H, y.W, Co).permute(0, 4, 1, 2, 3)
    (y64 * ct64).sum().backward()
    close(y.data, to_cl(y64.detach().numpy()), 1e-4, "y")
    close(w.grad, w64.grad.numpy(), 2e-4, "dw")
    close(b.grad, b64.grad.numpy(), 2e-4, "db")
    if not first:
        close(x.grad, to_cl(x64.grad.numpy()), 2e-4, "dx")
    close(y.planes(), y64.detach().numpy(), 1e-4, "planes()")


def test_gradient_sinks_of_the_visual_stack_change_nothing(monkeypatch):
    """Under m3t.ddp.FlatGradDDP the convolutions' weight / bias gradients, BatchNorm's dgamma / dbeta and CBAM's seven parameter gradients
    are written by their backward kernels straight into the flat gradient buffer (gradient sinks: no AccumulateGrad add per parameter --
    ~100 small launches per ResNet3D step).  The flat buffer must equal the one autograd accumulates (M3T_GRAD_SINKS=0): bit for bit on
    the VGG-M stem (Conv3d with bias, BatchNorm3d), to the rerun noise of MIOpen's strided data gradient on the 3-D ResNet with CBAM"""
    from m3t import ops
    from m3t.ddp import FlatGradDDP
    from models.backbone import VA_3DResNet, VA_3DVGGM
    rs = np.random.RandomState(21)
    for make, shape, exact in ((lambda: VA_3DResNet(frameLen=4, resnet_ver="v1", use_cbam=True, nClasses=2, nFCs=2), (2, 3, 4, 112, 112), False),
                               (lambda: VA_3DVGGM(frameLen=4, nClasses=2, backend="gru"), (2, 3, 4, 112, 112), True)):
        xn = draw(rs, shape)
        flats, calls = [], []
        for sinks in ("1", "0"):
            monkeypatch.setenv("M3T_GRAD_SINKS", sinks)
            m = fill_module(make(), 5).to(DEV).train()
            ddp = FlatGradDDP(m, max_norm=0.0)
            try:
                taken = [0]
                real = ops._take_sink
                monkeypatch.setattr(ops, "_take_sink", lambda p_, real=real, taken=taken: (lambda v: (taken.__setitem__(0, taken[0] + (v is not None)), v)[1])(real(p_)))
                ddp.zero_grad()
                y = m(dev(xn))
                y.square().mean().backward()
                ddp.finish()
                torch.cuda.synchronize()
                flats.append(ddp.flat.clone())
                calls.append(taken[0])
            finally:
                monkeypatch.setattr(ops, "_take_sink", real)
                ddp.close()
        assert calls[0] >= 20 and calls[1] == 0, calls          # the sinks were really used / really off
        if exact:
            assert torch.equal(flats[0], flats[1]), float((flats[0] - flats[1]).abs().max())
        else:               # (MIOpen's data gradient of the strided ResNet layers accumulates with atomics: reruns differ in the last bits upstream of them)
            assert float((flats[0] - flats[1]).abs().max()) <= 2e-5 * float(flats[1].abs().max())


def test_conv_walk_entry_points_planes_output_and_refusals():
    """the walks' C entry points called directly (include/m3t_hip.h): the planes output (written by the epilogue in one K pass, by reduction +
    transpose under split-K) equals the channels-last output transposed -- bit for bit --, and shapes no tile fits are refused with
    M3T_EINVAL instead of being mis-tiled (the host layer then takes the patch-matrix GEMMs or torch)"""
    from m3t import ops, _lib
    lib = ops.lib()
    rs = np.random.RandomState(3)
    for (N, Ci, Co, T, H, W, k, pd) in ((8, 64, 64, 1, 8, 8, (1, 3, 3), (0, 1, 1)),          # one K pass: the epilogue writes planes
                                        (4, 128, 256, 8, 4, 4, (3, 3, 3), (1, 1, 1)),        # deep layer: split-K slabs, reduction, transpose
                                        (3, 64, 64, 1, 5, 5, (1, 3, 3), (0, 1, 1)),          # round 6: 75 rows -- a ragged (and only) row tile, planes of 25
                                        (3, 128, 256, 3, 5, 7, (3, 3, 3), (1, 1, 1))):       # 315 rows: two whole tiles and a ragged one, split-K
        x = dev(draw(rs, (N * T * H * W, Ci)))
        w = dev(draw(rs, (Co, k[0] * k[1] * k[2] * Ci)) * 0.1)
        b = dev(draw(rs, (Co,)))
        sl = ops.amax_slots(2, x.device)
        ops.measure_amax([(x, sl.data_ptr()), (w, sl.data_ptr() + 8)])
        xi, wi = torch.empty_like(x), torch.empty_like(w)
        st = ops._stream()
        _lib.check(lib.m3t_f16x3_split(ops._p(x), x.shape[0], Ci, Ci, ops._p(xi), Ci, sl.data_ptr(), st), "split")
        _lib.check(lib.m3t_f16x3_split(ops._p(w), Co, w.shape[1], w.shape[1], ops._p(wi), w.shape[1], sl.data_ptr() + 8, st), "split")
        ws = ops.workspace(x.device)
        rows = N * T * H * W                                  # (stride 1, "same" padding: the output grid is the input grid)
        y_cl, y_cl2 = torch.empty(rows, Co, device=DEV), torch.empty(rows, Co, device=DEV)
        y_pl = torch.empty(N, Co, T * H * W, device=DEV)
        geo = (N, Ci, Co, T, H, W, k[0], k[1], k[2], 1, 1, 1, pd[0], pd[1], pd[2], sl.data_ptr(), sl.data_ptr() + 8, ops._p(ws), ws.numel() * 4)
        _lib.check(lib.m3t_conv3d_fwd_taps(ops._p(xi), ops._p(wi), ops._p(b), ops._p(y_cl), *geo, None, st), "fwd_taps")
        _lib.check(lib.m3t_conv3d_fwd_taps(ops._p(xi), ops._p(wi), ops._p(b), ops._p(y_cl2), *geo, ops._p(y_pl), st), "fwd_taps")
        torch.cuda.synchronize()
        assert torch.equal(y_pl, y_cl.view(N, T * H * W, Co).transpose(1, 2).contiguous())
        ref = torch.conv3d(x.view(N, T, H, W, Ci).permute(0, 4, 1, 2, 3).double().cpu(),
                           w.view(Co, k[0], k[1], k[2], Ci).permute(0, 4, 1, 2, 3).double().cpu(), b.double().cpu(), 1, pd)
        close(y_pl.view(N, Co, T, H, W), ref.numpy(), 1e-4, "y planes")
    # refusals
    bad = (N, Ci, 96, T, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 1, sl.data_ptr(), sl.data_ptr() + 8, None, 0)       # C_out % 64 != 0
    assert lib.m3t_conv3d_fwd_taps(ops._p(xi), ops._p(wi), None, ops._p(y_cl), *bad, None, st) == _lib.M3T_EINVAL
    bad = (N, 40, Co, T, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 1, sl.data_ptr(), sl.data_ptr() + 8, None, 0)       # C_in % 32 != 0 (and not a first layer)
    assert lib.m3t_conv3d_fwd_taps(ops._p(xi), ops._p(wi), None, ops._p(y_cl), *bad, None, st) == _lib.M3T_EINVAL
    # (rows need not fill tiles any more -- the two ragged cases above; the weight gradient over 75 reduction rows: test_conv3d_without_a_patch_matrix)
    assert lib.m3t_conv3d_fwd_taps4(ops._p(xi), ops._p(wi), None, ops._p(y_cl), 2, 64, 4, 16, 16, 1, 3, 9, 1, 1, 1, 0, 1, 4, sl.data_ptr(),
                                    sl.data_ptr() + 8, None, 0, None, st) == _lib.M3T_EINVAL              # nine taps in a row: the image holds eight
    assert lib.m3t_planes_to_cl4(ops._p(x), ops._p(y_cl), 2, 5, 64, st) == _lib.M3T_EINVAL                # more than four channels


@pytest.mark.parametrize("N,C_,T,H,W,training", [(3, 16, 5, 7, 9, True), (2, 64, 4, 12, 12, True), (2, 8, 3, 5, 5, False), (4, 24, 1, 1, 1, True)])
def test_batchnorm3d_relu_on_channel_planes(N, C_, T, H, W, training):
    """models.backbone.BatchNorm3dReLU (nn.BatchNorm3d + nn.ReLU of the 3-D stems, reference models/backbone.py:73-103,179-191) on
    csrc/bn.hip's channel-plane kernels: planes of 315 floats (scalar sweeps), of 576 (float4), eval mode, one value per plane --
    output, running statistics, input and parameter gradients against float64 torch on the CPU"""
    from models.backbone import BatchNorm3dReLU
    rs = np.random.RandomState(N + C_ + H)
    m = BatchNorm3dReLU(C_).to(DEV)
    ref = torch.nn.BatchNorm3d(C_).double()
    with torch.no_grad():
        for p_, q_ in ((m.weight, ref.weight), (m.bias, ref.bias), (m.running_mean, ref.running_mean), (m.running_var, ref.running_var)):
            v = rs.uniform(0.5, 1.5, C_) if p_ is m.weight or p_ is m.running_var else rs.uniform(-0.5, 0.5, C_)
            p_.copy_(torch.from_numpy(v.astype(np.float32))); q_.copy_(torch.from_numpy(v.astype(np.float32)).double())
    m.train(training); ref.train(training)
    xn = draw(rs, (N, C_, T, H, W))
    x = dev(xn, True)
    y = m(x)
    ctn = draw(rs, tuple(y.shape))
    (y * dev(ctn)).sum().backward()
    x64 = torch.tensor(xn, dtype=torch.float64, requires_grad=True)
    y64 = torch.relu(ref(x64))
    (y64 * torch.tensor(ctn, dtype=torch.float64)).sum().backward()
    close(y, y64.detach().numpy(), TOL, "y")
    close(x.grad, x64.grad.numpy(), TOL, "dx")
    close(m.weight.grad, ref.weight.grad.numpy(), 2e-4, "dgamma")
    close(m.bias.grad, ref.bias.grad.numpy(), 2e-4, "dbeta")
    close(m.running_mean, ref.running_mean.numpy(), 1e-5, "running_mean")
    close(m.running_var, ref.running_var.numpy(), 1e-5, "running_var")
    assert int(m.num_batches_tracked) == int(ref.num_batches_tracked)
    assert sorted(m.state_dict().keys()) == sorted(ref.state_dict().keys())


@pytest.mark.parametrize("N,C_,H,W,training,relu", [(37, 64, 28, 28, True, True), (21, 128, 14, 14, True, False), (19, 256, 7, 7, True, True),
                                                     (70, 512, 4, 4, True, True), (5, 24, 3, 5, False, True), (9, 16, 1, 1, True, False)])
def test_plane_batchnorm2d_on_the_resnet_maps(N, C_, H, W, training, relu):
    """models.resnet.PlaneBatchNorm2d (nn.BatchNorm2d [+ ReLU] of the per-frame ResNet, reference models/resnet.py:18-60) on csrc/bn.hip's
    small-plane kernels: 784 / 196 floats per plane (float4 units, 64 lanes per plane), 49 (scalar units), 16 (four lanes per plane, 16
    planes per wave), ragged plane counts, eval mode -- against float64 torch on the CPU"""
    from models.resnet import PlaneBatchNorm2d
    rs = np.random.RandomState(N + C_ + H)
    m = PlaneBatchNorm2d(C_, fuse_relu=relu).to(DEV)
    ref = torch.nn.BatchNorm2d(C_).double()
    with torch.no_grad():
        for p_, q_ in ((m.weight, ref.weight), (m.bias, ref.bias), (m.running_mean, ref.running_mean), (m.running_var, ref.running_var)):
            v = rs.uniform(0.5, 1.5, C_) if p_ is m.weight or p_ is m.running_var else rs.uniform(-0.5, 0.5, C_)
            p_.copy_(torch.from_numpy(v.astype(np.float32))); q_.copy_(torch.from_numpy(v.astype(np.float32)).double())
    m.train(training); ref.train(training)
    xn = draw(rs, (N, C_, H, W))
    x = dev(xn, True)
    y = m(x)
    ctn = draw(rs, tuple(y.shape))
    (y * dev(ctn)).sum().backward()
    x64 = torch.tensor(xn, dtype=torch.float64, requires_grad=True)
    y64 = ref(x64)
    if relu:
        y64 = torch.relu(y64)
    (y64 * torch.tensor(ctn, dtype=torch.float64)).sum().backward()
    close(y, y64.detach().numpy(), TOL, "y")
    close(x.grad, x64.grad.numpy(), TOL, "dx")
    close(m.weight.grad, ref.weight.grad.numpy(), 2e-4, "dgamma")
    close(m.bias.grad, ref.bias.grad.numpy(), 2e-4, "dbeta")
    close(m.running_mean, ref.running_mean.numpy(), 1e-5, "running_mean")
    close(m.running_var, ref.running_var.numpy(), 1e-5, "running_var")
    assert sorted(m.state_dict().keys()) == sorted(ref.state_dict().keys())


@pytest.mark.parametrize("shape,k,s,p", [((2, 5, 3, 55, 55), (1, 2, 2), (1, 2, 2), (0, 0, 0)), ((2, 4, 3, 56, 56), (1, 3, 3), (1, 2, 2), (0, 1, 1)),
                                         ((1, 3, 2, 7, 9), (1, 3, 2), (1, 1, 2), (0, 1, 0)), ((3, 2, 1, 4, 4), (1, 2, 2), (1, 2, 2), (0, 0, 0)), ((2, 3, 2, 11, 10), (1, 2, 1), (1, 3, 2), (0, 0, 0)),
                                         ((2, 3, 1, 6, 9), (1, 2, 2), (1, 4, 4), (0, 0, 0))])      # (k < s, last stride box past the map's rows while columns are left over: ADVICE r4)
def test_spatial_max_pooling_of_the_stems(shape, k, s, p):
    """models.backbone.SpatialMaxPool3d (nn.MaxPool3d((1, k, k)) of the 3-D stems, reference models/backbone.py:80,86,92,182) on
    csrc/bn.hip's plane kernels: odd maps whose last row / column no window covers, overlapping padded windows (gather backward),
    post-ReLU inputs full of ties, NaN -- values and gradients equal to torch's own kernels exactly"""
    from models.backbone import SpatialMaxPool3d
    rs = np.random.RandomState(sum(shape))
    xn = np.maximum(draw(rs, shape), 0.0).astype(np.float32)          # many exact zeros: ties
    xn[0, 0, 0, 1, 2] = np.nan
    m = SpatialMaxPool3d(k, s, p)
    x1, x2 = dev(xn, True), dev(xn, True)
    y1 = m(x1)
    y2 = torch.nn.functional.max_pool3d(x2, k, s, p)
    ct = dev(draw(rs, tuple(y2.shape)))
    y1.backward(ct); y2.backward(ct)
    assert y1.shape == y2.shape and torch.equal(torch.nan_to_num(y1, nan=-7.0), torch.nan_to_num(y2, nan=-7.0))
    # a position that won several overlapping windows sums their gradients: same terms, another order than torch's atomics
    d = (x1.grad - x2.grad).abs()
    assert float(d.max()) <= 1e-6, (float(d.max()), int((d > 1e-6).sum()))
    if k[1] <= s[1] and k[2] <= s[2]:
        assert torch.equal(x1.grad, x2.grad)


def test_c5_affwild_av_golden():
    """Full AffWild2VA audiovisual/attention/v2p_split on raw frames (conv stem on MIOpen)."""
    from models.model import AffWild2VA
    g = load_golden("c5_affwild_av")
    seed = int(g["seed"])
    B, T = [int(v) for v in g["dims"]]
    m = fill_module(AffWild2VA(_hp(modality="audiovisual", fusion_type="attention", loss="ccc_mtl", window=T)),
                    seed + 1).to(DEV).eval()
    batch = _affwild_batch(np.random.RandomState(seed), B, T, video=True)
    close(m(batch), g["y"], TOL, "y")                    # north_star bar; measured 1e-7 end to end (round 3: the MIOpen stem's output
    out = m.training_step(batch, 0)                      # is within 9e-6 of the reference's, test_c5_temporal_part_on_the_references_stem_features)
    close(out["loss"], g["loss"], TOL, "loss")
    out["loss"].backward()
    check_digests([(n, p.grad) for n, p in m.named_parameters() if p.grad is not None], g, tol=C5_DIGEST_TOL)


def test_c5_resnet3d_cbam_golden():
    from models.backbone import VA_3DResNet
    g = load_golden("c5_resnet3d_cbam")
    seed = int(g["seed"])
    B, T = [int(v) for v in g["dims"]]
    m = fill_module(VA_3DResNet(frameLen=T, resnet_ver="v1", use_cbam=True, nClasses=2, nFCs=2), seed + 1).to(DEV).eval()
    rs = np.random.RandomState(seed)
    x = dev(rs.randint(0, 256, (B, 3, T, 112, 112)).astype(np.float32))
    x = ((x - 127.5) / 127.5).requires_grad_(True)
    y = m(x)
    close(y, g["y"], 2e-4, "y")
    (y * dev(g["ct"])).sum().backward()
    check_digests([(n, p.grad) for n, p in m.named_parameters() if p.grad is not None], g, tol=2e-3)


@pytest.mark.parametrize("name,backend", [("vggm_tcn_eval", "tcn"), ("vggm_gru_eval", "gru")])
def test_vggm_end_to_end_golden(name, backend):
    """VA_3DVGGM(...).forward(video) from raw frames (reference models/backbone.py:134-145): the unsplit VGG-M stem (MIOpen) ->
    TemporalConvNet + Linear(512,2) on the HIP conv kernels (backend 'tcn': the one TemporalConvNet user of the reference) or
    the BiGRU head; outputs and every parameter gradient against the reference's own run"""
    from models.backbone import VA_3DVGGM
    g = load_golden(name)
    seed = int(g["seed"])
    B, T = [int(v) for v in g["dims"]]
    m = fill_module(VA_3DVGGM(frameLen=T, backend=backend, nClasses=2, nFCs=2), seed + 1).to(DEV).eval()
    rs = np.random.RandomState(seed)
    x = dev(rs.randint(0, 256, (B, 3, T, 112, 112)).astype(np.float32))
    x = ((x - 127.5) / 127.5).requires_grad_(True)
    y = m(x)
    assert tuple(y.shape) == tuple(g["y"].shape)
    close(y, g["y"], 2e-4, "y")
    (y * dev(g["ct"])).sum().backward()
    check_digests([(n, p.grad) for n, p in m.named_parameters() if p.grad is not None], g, tol=2e-3)
    check_digests([("dx", x.grad)], {"gd.dx": g["dx"]}, tol=2e-3)


def test_c5_affwild_av_t16_golden():
    """the full A+V model on 16-frame clips (longer scans than the T=4 golden; BASELINE's 64-frame size: the t64 golden and the
    property test below)"""
    from models.model import AffWild2VA
    g = load_golden("c5_affwild_av_t16")
    seed = int(g["seed"])
    B, T = [int(v) for v in g["dims"]]
    m = fill_module(AffWild2VA(_hp(modality="audiovisual", fusion_type="attention", loss="ccc_mtl", window=T)),
                    seed + 1).to(DEV).eval()
    batch = _affwild_batch(np.random.RandomState(seed), B, T, video=True)
    close(m(batch), g["y"], TOL, "y")                    # north_star bar; measured 1e-7 end to end (round 3: the MIOpen stem's output
    out = m.training_step(batch, 0)                      # is within 9e-6 of the reference's, test_c5_temporal_part_on_the_references_stem_features)
    close(out["loss"], g["loss"], TOL, "loss")
    out["loss"].backward()
    check_digests([(n, p.grad) for n, p in m.named_parameters() if p.grad is not None], g, tol=C5_DIGEST_TOL)


def test_c5_affwild_av_t64_golden():
    """the full A+V model at BASELINE's window of the end-to-end config (64 frames per clip, the length bench.py's C5 leg times; VERDICT r4:
    "C5 at T=64 is property-only"): outputs, loss and every parameter's gradient digest against the reference run on the same seeded batch
    (tests/golden/gen_golden.py c5t64)"""
    from models.model import AffWild2VA
    g = load_golden("c5_affwild_av_t64")
    seed = int(g["seed"])
    B, T = [int(v) for v in g["dims"]]
    assert T == 64
    m = fill_module(AffWild2VA(_hp(modality="audiovisual", fusion_type="attention", loss="ccc_mtl", window=T)),
                    seed + 1).to(DEV).eval()
    batch = _affwild_batch(np.random.RandomState(seed), B, T, video=True)
    close(m(batch), g["y"], TOL, "y")
    out = m.training_step(batch, 0)
    close(out["loss"], g["loss"], TOL, "loss")
    out["loss"].backward()
    check_digests([(n, p.grad) for n, p in m.named_parameters() if p.grad is not None], g, tol=C5_DIGEST_TOL)


def _check_bn_state(m, g, tol):
    """BatchNorm buffers after ONE training step against the reference's (running statistics: relative to their own scale)"""
    n_checked = 0
    for n, b in m.named_buffers():
        leaf = n.split(".")[-1]
        if leaf not in ("running_mean", "running_var", "num_batches_tracked"):
            continue
        ref = g["bn." + n]
        got = b.detach().cpu().double().numpy()
        if leaf == "num_batches_tracked":
            assert int(got) == int(ref), (n, got, ref)
        else:
            assert float(np.abs(got - ref).max()) <= tol * max(1.0, float(np.abs(ref).max())), (n, float(np.abs(got - ref).max()))
        n_checked += 1
    return n_checked


def test_c5_affwild_av_train_mode_golden():
    """TRAIN mode end to end (what bench.py's C5 leg times): AffWild2VA A+V training_step on raw frames with BatchNorm3d on
    BATCH statistics through the five stem groups (reference models/backbone.py:179-271, models/model.py:146-218) -- outputs, loss, every
    parameter-gradient digest and every BatchNorm buffer after the step against the reference's own run (tests/golden/gen_golden.py c5train)"""
    from models.model import AffWild2VA
    from m3t import ops
    g = load_golden("c5_affwild_av_t16_train")
    seed = int(g["seed"])
    B, T = [int(v) for v in g["dims"]]
    m = fill_module(AffWild2VA(_hp(modality="audiovisual", fusion_type="attention", loss="ccc_mtl", window=T)), seed + 1).to(DEV).train()
    batch = _affwild_batch(np.random.RandomState(seed), B, T, video=True)
    ys = {}
    fwd = m.forward

    def tap(b):
        ys["y"] = fwd(b)
        return ys["y"]
    m.forward = tap
    n_torch = ops.CONV3D_CALLS["torch"]
    out = m.training_step(batch, 0)
    del m.forward
    assert ops.CONV3D_CALLS["torch"] == n_torch, "a convolution of the C5 training step took the stock operator"
    close(ys["y"], g["y"], TOL, "y")
    close(out["loss"], g["loss"], TOL, "loss")
    out["loss"].backward()
    check_digests([(n, p.grad) for n, p in m.named_parameters() if p.grad is not None], g, tol=C5_DIGEST_TOL)
    assert _check_bn_state(m, g, 2e-4) >= 15


def test_c5_resnet3d_cbam_train_mode_golden():
    """VA_3DResNet(use_cbam=True) in TRAIN mode (bench.py's aux.cbam_resnet3d): BatchNorm3d of the stem, the per-frame ResNet-18's 20
    BatchNorm2d and the 8 CBAM gates' BatchNorm2d(1) on batch statistics (reference models/backbone.py:327-355, models/resnet.py:40-56,
    models/cbam.py:84-93): outputs, parameter- and input-gradient digests, buffers after the step"""
    from models.backbone import VA_3DResNet
    g = load_golden("c5_resnet3d_cbam_train")
    seed = int(g["seed"])
    B, T = [int(v) for v in g["dims"]]
    m = fill_module(VA_3DResNet(frameLen=T, resnet_ver="v1", use_cbam=True, nClasses=2, nFCs=2), seed + 1).to(DEV).train()
    rs = np.random.RandomState(seed)
    x = dev(rs.randint(0, 256, (B, 3, T, 112, 112)).astype(np.float32))
    x = ((x - 127.5) / 127.5).requires_grad_(True)
    from m3t import ops
    stock0 = dict(ops.STOCK_FALLBACKS)
    y = m(x)
    close(y, g["y"], 2e-4, "y")
    (y * dev(g["ct"])).sum().backward()
    # round 6: no stock (MIOpen) convolution kernel in this training step -- forward, weight gradient and data gradient of every layer,
    # the six stride-2 layers' data gradient included (parity-class walks), run on the library's tap walks
    # (the stem's gradient w.r.t. the VIDEO exists only because this test asks for dx: a training step's input needs none)
    if ops.CONV3D_IMPLICIT[0] and ops.CONV3D_GEMM[0] and ops.CONV3D_TAPS[0]:       # (not under the A/B switches that ask for the other paths)
        assert {k: v for k, v in ops.STOCK_FALLBACKS.items() if v != stock0.get(k, 0) and k.startswith("conv") and "k(64, 3, " not in k} == {}, \
            ops.STOCK_FALLBACKS
    check_digests([(n, p.grad) for n, p in m.named_parameters() if p.grad is not None], g, tol=2e-3)
    check_digests([("dx", x.grad)], {"gd.dx": g["dx"]}, tol=2e-3)
    assert _check_bn_state(m, g, 2e-4) >= 60


def test_c5_no_grad_forward_takes_the_same_walks():
    """validation_step / test_step (reference models/model.py:226-246,320-337: self.forward under no_grad, eval mode) and --freeze_enc training
    (model.py:376-386) run the stems' convolutions on the SAME tap walks as a training step (round 6: one path whatever the grad mode;
    until round 5 they fell back to MIOpen): the eval golden's outputs under torch.no_grad(), with no stock-operator convolution, and the
    frozen-encoder step's remaining gradients equal to the unfrozen step's"""
    from models.model import AffWild2VA
    from m3t import ops
    g = load_golden("c5_affwild_av_t16")
    seed = int(g["seed"])
    B, T = [int(v) for v in g["dims"]]
    m = fill_module(AffWild2VA(_hp(modality="audiovisual", fusion_type="attention", loss="ccc_mtl", window=T)), seed + 1).to(DEV).eval()
    batch = _affwild_batch(np.random.RandomState(seed), B, T, video=True)
    before = dict(ops.CONV3D_CALLS)
    with torch.no_grad():
        y = m(batch)
    assert ops.CONV3D_CALLS["torch"] == before["torch"] and ops.CONV3D_CALLS["walk"] > before["walk"], (before, ops.CONV3D_CALLS)
    close(y, g["y"], TOL, "y (no_grad)")
    # frozen encoders (reference model.py:376-386 freezes self.visual / self.audio parameters): the walks run, the heads' gradients are unchanged
    out = m.training_step(batch, 0)
    out["loss"].backward()
    live = ("fusion.", "proj_v.", "att_fuse.")           # configure_optimizers with --freeze_enc: everything else is frozen
    ref = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None and n.startswith(live)}
    for p in m.parameters():
        p.grad = None
    for n, p in m.named_parameters():
        if not n.startswith(live):
            p.requires_grad_(False)
    before = dict(ops.CONV3D_CALLS)
    out = m.training_step(batch, 0)
    out["loss"].backward()
    assert ops.CONV3D_CALLS["torch"] == before["torch"] and ops.CONV3D_CALLS["walk"] > before["walk"]
    assert len(ref) > 20
    for n, p in m.named_parameters():
        if not n.startswith(live):
            assert p.grad is None, n
        elif n in ref:
            assert torch.equal(p.grad, ref[n]), n


def test_cached_magnitudes_follow_in_place_changes():
    """the per-step weight magnitude table and the tags riding on tensors (|h| <= 1 of a GRU output, a TemporalBlock's measured slot)
    are dropped when the tensor is written in place after the measurement -- a stale maximum would overflow the fp16x3 scale."""
    from m3t import ops
    w = torch.nn.Parameter(torch.randn(64, 32, device=DEV))
    ops.measure_weight_amax([w], owner=None)
    try:
        assert ops.weight_amax(w) is not None
        with torch.no_grad():
            w.mul_(1e4)                                    # e.g. optimizer.step() placed after zero_grad(), load_state_dict, a clamp
        assert ops.weight_amax(w) is None
        x = torch.randn(128, 32, device=DEV)
        y = ops.linear(x, w, None, 0)                      # measures for itself again: finite and right
        ref = x.double() @ w.detach().double().t()
        assert float((y.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    finally:
        ops.drop_weight_amax(None)
    # round 6: under no_grad (validation / test steps) a weight is measured ONCE and its slot kept for as long as the parameter is that tensor
    # at that version; grad mode on never reads that cache
    w2 = torch.nn.Parameter(torch.randn(64, 32, device=DEV))
    x = torch.randn(128, 32, device=DEV)
    with torch.no_grad():
        a1, a2 = ops.weight_amax(w2), ops.weight_amax(w2)
        assert a1 is not None and a1 == a2
        y1 = ops.linear(x, w2, None, 0)
        w2.mul_(1e4)
        a3 = ops.weight_amax(w2)
        assert a3 is not None and a3 != a1                 # measured again, into a slot of its own
        y2 = ops.linear(x, w2, None, 0)
    ref = x.double() @ w2.detach().double().t()
    assert float((y2.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float((y1.double() * 1e4 - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert ops.weight_amax(w2) is None
    h = torch.rand(4, 8, 16, device=DEV)
    h._m3t_unit = h._version + 1
    assert ops._is_unit(h)
    h.mul_(3.0)
    assert not ops._is_unit(h)


def test_c1_affwild_audio_db_scale_golden():
    """The reference's REAL audio input scale: un-normalised power_to_db log-Mel values, |x| ~ 40 (reference
    process/extract_melspec.py:13-20, models/dataset.py:83-95; SURVEY 8(d) "secondary run U(-80, 0)").  Inputs that large
    saturate the gates and scale the absolute error of the bf16x6 input projection by |x|: the north_star bars (y within 1e-4,
    loss terms within 1e-4) must hold there too.  Golden: the reference's AffWild2VA(modality='audio') = GRU(200,256,2,9,2)."""
    from models.model import AffWild2VA
    g = load_golden("c1_affwild_audio_db")
    seed = int(g["seed"])
    m = fill_module(AffWild2VA(_hp(modality="audio", loss="ccc_mtl")), seed + 1).to(DEV).eval()
    rs = np.random.RandomState(seed)                       # drawn in gen_golden.case_affwild_audio's order
    batch = {"audio": dev(rs.uniform(-80.0, 0.0, (4, 100, 200)).astype(np.float32)),
             "label_valence": dev(draw(rs, (4, 100), "uniform_pm1")), "label_arousal": dev(draw(rs, (4, 100), "uniform_pm1")),
             "class_expr": dev(rs.randint(0, 7, (4, 100)).astype(np.int64)), "expr_valid": dev(rs.uniform(size=(4, 100)) < 0.7)}
    assert float(batch["audio"].abs().mean()) > 30.0
    y = m(batch)
    err = float((y.detach().cpu().double() - torch.from_numpy(g["y"]).double()).abs().max())
    print("c1 audio, dB-scale input: |y - reference| = %.2e (|y| max %.2f)" % (err, float(np.abs(g["y"]).max())))
    close(y, g["y"], TOL, "y")
    out = m.training_step(batch, 0)
    close(out["loss"], g["loss"], TOL, "loss")
    close(out["log"]["loss_v"], g["loss_v"], TOL, "loss_v")
    close(out["log"]["loss_a"], g["loss_a"], TOL, "loss_a")
    close(out["log"]["loss_expr"], g["loss_expr"], TOL, "loss_expr")
    out["loss"].backward()
    check_digests(list((n, p.grad) for n, p in m.named_parameters()), g)


@pytest.mark.parametrize("name", ["c5_affwild_av", "c5_affwild_av_t16"])
def test_c5_temporal_part_on_the_references_stem_features(name):
    """Where do the 2e-4 of the end-to-end C5 tests come from?  The golden now carries the reference's conv-stem OUTPUT (both
    private towers, [B,512,T,1,1]); fed into this repo's model in place of its own (MIOpen) stem, everything BEHIND the stem --
    feature concat, the three encoders, proj_v, AttFusion, the fusion GRU, the loss, and their gradients -- is the HIP path
    alone and must meet the north_star bar: y and loss within 1e-4, gradient digests within 2e-4.  What the end-to-end tests
    add on top is then MIOpen's convolution arithmetic, measured and printed here."""
    from models.model import AffWild2VA
    from models.backbone import _squeeze_hw
    g = load_golden(name)
    seed = int(g["seed"])
    B, T = [int(v) for v in g["dims"]]
    m = fill_module(AffWild2VA(_hp(modality="audiovisual", fusion_type="attention", loss="ccc_mtl", window=T)),
                    seed + 1).to(DEV).eval()
    batch = _affwild_batch(np.random.RandomState(seed), B, T, video=True)
    with torch.no_grad():
        own_v = m.visual.v_private(m.visual.shared((batch["video"] - 127.5) / 127.5))
        if not isinstance(own_v, torch.Tensor):          # round 6: the stem is a channels-last chain (m3t.ops.CLTensor)
            own_v = own_v.planes()
    stem_err = float((own_v.cpu().double() - torch.from_numpy(g["feat_v"]).double()).abs().max())
    y_e2e = m(batch)
    fv, fa = dev(g["feat_v"]), dev(g["feat_a"])
    m.visual.features = lambda x, se, au: (torch.cat((_squeeze_hw(fv), se), dim=1), torch.cat((_squeeze_hw(fa), au), dim=1))
    y = m(batch)
    e_free = float((y.detach().cpu().double() - torch.from_numpy(g["y"]).double()).abs().max())
    e_e2e = float((y_e2e.detach().cpu().double() - torch.from_numpy(g["y"]).double()).abs().max())
    print("%s: |y - ref| stem-free %.2e, end to end %.2e; MIOpen stem output vs the reference's %.2e (features of scale %.2f)"
          % (name, e_free, e_e2e, stem_err, float(np.abs(g["feat_v"]).max())))
    close(y, g["y"], TOL, "y (stem-free)")
    out = m.training_step(batch, 0)
    close(out["loss"], g["loss"], TOL, "loss (stem-free)")
    out["loss"].backward()
    behind = [(n, p.grad) for n, p in m.named_parameters()
              if p.grad is not None and not n.startswith(("visual.shared", "visual.v_private", "visual.a_private"))]
    assert len(behind) > 60
    check_digests(behind, g, tol=2e-4)


def test_c5_full_size_properties_t64():
    """BASELINE configs[4] size: 64-frame clips of raw 112x112 frames through the whole AffWild2VA A+V model (eval mode: BatchNorm
    on running statistics).  Size-independent properties: finite outputs of the right shape, run-to-run determinism of forward
    and of every parameter gradient, clip independence (a permuted batch gives the permuted outputs; MIOpen may pick another
    conv algorithm per call, hence a rounding-level tolerance there), and the loss equals the loss kernel on the outputs."""
    from models.model import AffWild2VA
    from m3t import ops
    B, T = 8, 64                                  # bench.py's own C5 batch (was 4 x 64)
    torch.manual_seed(12345)
    m = AffWild2VA(_hp(modality="audiovisual", fusion_type="attention", loss="ccc_mtl", window=T)).to(DEV).eval()
    batch = _affwild_batch(np.random.RandomState(7), B, T, video=True)
    y1 = m(batch)
    assert tuple(y1.shape) == (B, T, 9) and torch.isfinite(y1).all()
    y2 = m(batch)
    assert torch.equal(y1, y2), "forward is not deterministic"
    perm = torch.tensor([2, 0, 3, 1, 6, 7, 5, 4], device=DEV)
    pb = {k: (v[perm].contiguous() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    y3 = m(pb)
    assert float((y3 - y1[perm]).abs().max()) <= 1e-5, "clips are not independent"
    grads = []
    for _ in range(2):
        m.zero_grad()
        out = m.training_step(batch, 0)
        out["loss"].backward()
        grads.append([p.grad.clone() for p in m.parameters() if p.grad is not None])
    assert torch.isfinite(out["loss"]) and all(torch.isfinite(g).all() for g in grads[0])
    loss2, _ = ops.va_loss(y1.detach(), batch["label_valence"], batch["label_arousal"], batch["class_expr"], batch["expr_valid"],
                           iv=7, ia=8, n_expr=7)
    assert abs(float(loss2) - float(out["loss"])) <= 1e-6
    worst = max(float((a - b).abs().max()) / max(1e-12, float(b.abs().max())) for a, b in zip(grads[0], grads[1]))
    assert worst <= 1e-5, "gradients differ run to run by %.2e" % worst
    # TRAIN mode at the bench's size (BatchNorm3d on batch statistics): finite, deterministic, and no convolution on a stock operator
    m.train()
    n_torch, n_walk = ops.CONV3D_CALLS["torch"], ops.CONV3D_CALLS["walk"]
    tg = []
    for _ in range(2):
        m.zero_grad()
        out = m.training_step(batch, 0)
        out["loss"].backward()
        tg.append([p.grad.clone() for p in m.parameters() if p.grad is not None])
    assert torch.isfinite(out["loss"]) and all(torch.isfinite(g).all() for g in tg[0])
    assert ops.CONV3D_CALLS["torch"] == n_torch and ops.CONV3D_CALLS["walk"] > n_walk
    worst = max(float((a - b).abs().max()) / max(1e-12, float(b.abs().max())) for a, b in zip(tg[0], tg[1]))
    assert worst <= 1e-5, "train-mode gradients differ run to run by %.2e" % worst
    ops.poll_scan_error(sync=True)


# ------------------------------------------------------------------------------ full-size properties
def test_full_size_properties_c3():
    """BASELINE size (B=32, T=300): determinism, clip independence (permutation equivariance),
    exact linearity of the backward pass in the cotangent."""
    from m3t.workloads import AVFeatureGraph
    torch.manual_seed(12345)
    m = AVFeatureGraph().to(DEV)
    xa = torch.randn(32, 300, 128, device=DEV)
    xv = torch.randn(32, 300, 256, device=DEV, requires_grad=True)
    y1 = m(xa, xv)
    y2 = m(xa, xv)
    assert torch.equal(y1, y2), "forward is not deterministic"
    assert torch.isfinite(y1).all()
    perm = torch.randperm(32, device=DEV)
    y3 = m(xa[perm].contiguous(), xv[perm].contiguous())
    assert torch.equal(y3, y1[perm]), "clips are not independent"
    ct = torch.randn_like(y1)
    (g1,) = torch.autograd.grad(y1, xv, ct, retain_graph=True)
    (g2,) = torch.autograd.grad(y1, xv, 2 * ct)
    assert torch.equal(2 * g1, g2), "backward is not linear in the cotangent"


def test_time_reversal_symmetry_of_bigru():
    """swap forward/reverse weights + reverse time => output halves swap and time reverses."""
    from models.rnn import GRU
    torch.manual_seed(5)
    a = GRU(16, 32, 1, -1).to(DEV)
    b = GRU(16, 32, 1, -1).to(DEV)
    with torch.no_grad():
        for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            getattr(b.gru, k + "_l0").copy_(getattr(a.gru, k + "_l0_reverse"))
            getattr(b.gru, k + "_l0_reverse").copy_(getattr(a.gru, k + "_l0"))
    x = torch.randn(3, 21, 16, device=DEV)
    ya, yb = a(x), b(x.flip(1).contiguous())
    assert torch.equal(ya[..., :32], yb.flip(1)[..., 32:]) and torch.equal(ya[..., 32:], yb.flip(1)[..., :32])


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("B,T", [(5, 7), (32, 300), (1, 2)])
def test_solo_scans_h128_through_the_c_abi(B, T, bf16):
    """gru_solo.hip (H = 128 levels: one workgroup per (scan, clip), W_hh in registers, DPP-broadcast FMA chains) through
    m3t_gru_scan_fwd / _bwd directly: a forward and a reverse scan in one call, with h_n / dh_n, both weight layouts of the
    backward ABI (W_hh^T, and the untransposed parameter under M3T_SCAN_WHH), fp32 and the bf16 mode -- against the
    launch-per-step kernels (M3T_SCAN_NO_PERSIST) on the same buffers."""
    import ctypes as C
    from m3t import _lib, ops
    from m3t._lib import GruFwdDesc, GruBwdDesc
    lib = _lib.load()
    H = 128
    torch.manual_seed(B * 1000 + T)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = ops.workspace(torch.device(DEV))
    wsp, wsb = C.c_void_p(ws.data_ptr()), ws.numel() * 4
    mode = _lib.M3T_BF16 if bf16 else 0
    xproj = torch.randn(B, T, 6 * H, device=DEV) * 0.7
    w = [torch.randn(3 * H, H, device=DEV) / H ** 0.5 for _ in (0, 1)]
    bias = [torch.randn(3 * H, device=DEV) * 0.1 for _ in (0, 1)]

    def fwd(flags):
        out, gates, hn = torch.zeros(B, T, 2 * H, device=DEV), torch.zeros(2, B, T, 4 * H, device=DEV), torch.zeros(2, B, H, device=DEV)
        descs = [GruFwdDesc(xproj.data_ptr(), w[d].data_ptr(), bias[d].data_ptr(), out.data_ptr(), gates[d].data_ptr(), hn[d].data_ptr(),
                            H, d, 6 * H, d * 3 * H, 2 * H, d * H) for d in (0, 1)]
        n0 = lib.m3t_gru_persist_count()
        rc = lib.m3t_gru_scan_fwd((GruFwdDesc * 2)(*descs), 2, B, T, wsp, wsb, flags | mode, s)
        assert rc == 0, rc
        torch.cuda.synchronize()
        return out, gates, hn, lib.m3t_gru_persist_count() - n0

    out1, gates1, hn1, n1 = fwd(0)
    out0, gates0, hn0, n0 = fwd(_lib.M3T_SCAN_NO_PERSIST)
    assert n1 == 1 and n0 == 0
    # bf16 mode: the two paths sum in different orders, so a rounding of h_t to bf16 can flip (2^-9 of that element) and the
    # recurrence carries it on: fp32-rounding agreement on short clips, bf16-class agreement over 300 steps
    tol = (2e-5 if T < 10 else 5e-3) if bf16 else 3e-6
    close(out1, out0, tol, "out"); close(gates1, gates0, tol, "gates"); close(hn1, hn0, tol, "h_n")
    assert torch.equal(hn1[0], out1[:, T - 1, :H]) and torch.equal(hn1[1], out1[:, 0, H:])

    dout = torch.randn(B, T, 2 * H, device=DEV) * 0.2
    dhn = torch.randn(2, B, H, device=DEV) * 0.2
    wt = [x.t().contiguous() for x in w]

    def bwd(flags, direct):
        dgx, dgh = torch.zeros(B, T, 6 * H, device=DEV), torch.zeros(2, B, T, 3 * H, device=DEV)
        dh, dbp = torch.zeros(2, B, H, device=DEV), torch.zeros(2, B, 4, H, device=DEV)
        dbi, dbh = torch.zeros(2, 3 * H, device=DEV), torch.zeros(2, 3 * H, device=DEV)
        descs = [GruBwdDesc(dout.data_ptr(), out0.data_ptr(), gates0[d].data_ptr(), (w[d] if direct else wt[d]).data_ptr(), dhn[d].data_ptr(),
                            dgx.data_ptr(), dgh[d].data_ptr(), dh[d].data_ptr(), dbp[d].data_ptr(), dbi[d].data_ptr(), dbh[d].data_ptr(),
                            H, d, 2 * H, d * H, 6 * H, d * 3 * H) for d in (0, 1)]
        n0_ = lib.m3t_gru_persist_count()
        rc = lib.m3t_gru_scan_bwd((GruBwdDesc * 2)(*descs), 2, B, T, wsp, wsb, flags | mode | (_lib.M3T_SCAN_WHH if direct else 0), s)
        assert rc == 0, rc
        torch.cuda.synchronize()
        return (dgx, dgh, dh, dbi, dbh), lib.m3t_gru_persist_count() - n0_

    ref, nr = bwd(_lib.M3T_SCAN_NO_PERSIST, False)
    assert nr == 0
    for direct in (False, True):
        got, n = bwd(0, direct)
        assert n == 1
        for a, b_, name in zip(got, ref, ("dgx", "dgh", "dh", "db_ih", "db_hh")):
            close(a, b_, (3e-5 if T < 10 else 5e-3) if bf16 else 5e-6, "%s (direct=%d)" % (name, direct))
    ops.poll_scan_error()


@pytest.mark.parametrize("switch", ["M3T_CONV_X6", "M3T_CBAM_FUSED", "M3T_CBAM_RESIDENT", "M3T_BN_PLANES", "M3T_CONV3D_IMPLICIT", "M3T_STEM_CL",
                                    "M3T_BN_POOL_FUSED", "M3T_WGRAD_IMAGES", "M3T_CONV_WGRAD_STREAM", "M3T_TRANSPOSE_IMAGES"])
def test_conv_and_cbam_kernel_switches(switch):
    """README's switch table, the entries the C3 step does not exercise (read once per process, hence a child):
    M3T_CONV_X6=0 -- the TCN / tcn_simple convolutions on the fp32-MFMA kernel instead of the bf16x6 implicit GEMM;
    M3T_CBAM_FUSED=0 -- CBAM as channel gate + spatial gate instead of the fused operator;
    M3T_CBAM_RESIDENT=0 -- small frames (7 x 7, 4 x 4 ...) on the fused operator's general kernels F1 / B2 instead of the
    frame-resident F1L / B2L;
    M3T_CONV3D_IMPLICIT=0 -- Conv3d / the ResNet's Conv2d forward and weight gradient on the patch-matrix GEMMs (first half of round 5) instead
    of the tap walks over channels-last activations: the stems' goldens and the layer tests again on that path;
    M3T_BN_PLANES=0 -- BatchNorm3d / BatchNorm2d (+ReLU) of the stems and the per-frame ResNet on the stock ops instead of the channel-plane
    kernels;
    M3T_STEM_CL=0 -- the VGG-M stems on the planes operators (a transpose on each side of every convolution) instead of the channels-last chain
    of round 6; M3T_BN_POOL_FUSED=0 -- the chain with BatchNorm + ReLU and the pooling as two operators instead of one; M3T_WGRAD_IMAGES=0 -- the
    convolutions' weight-gradient walk on fp32 operands split in its loop instead of the images the other walks made; M3T_CONV_WGRAD_STREAM=0 -- that
    walk on its layer's stream instead of a weight-gradient stream; M3T_TRANSPOSE_IMAGES=0 -- the planes path's inputs and output gradients as fp32
    rows first (transpose + measure, then split) even where their producer raised the slot.  Same arithmetic: the convolution, TemporalBlock, CBAM, ResNet and C5 parity tests must pass unchanged."""
    import subprocess
    import sys
    env = dict(os.environ, M3T_SCAN_LOCK="0", **{switch: "0"})
    pick = {"M3T_CONV_X6": "conv1d_on_the_bf16x6_pipe or tcn_train_mode or tcn_golden",
            "M3T_CONV3D_IMPLICIT": "c5_resnet3d or c5_affwild_av_golden or conv3d_weight_gradient or conv3d_forward_on_the_patch",
            "M3T_BN_PLANES": "resnet_cbam or c5_resnet3d or c5_affwild_av_t16",
            "M3T_STEM_CL": "c5_affwild_av_train or vggm_end_to_end",
            "M3T_BN_POOL_FUSED": "c5_affwild_av_train or vggm_end_to_end",
            "M3T_WGRAD_IMAGES": "c5_resnet3d_cbam_train or conv3d_on_channels_last",
            "M3T_CONV_WGRAD_STREAM": "gradient_sinks_of_the_visual or c5_affwild_av_train_mode",
            "M3T_TRANSPOSE_IMAGES": "c5_resnet3d_cbam_train or resnet_cbam"}.get(switch, "cbam_golden or resnet_cbam or cbam_stage")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k", pick], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:]
    assert " passed" in r.stdout
