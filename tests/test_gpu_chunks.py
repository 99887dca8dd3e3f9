"""Direction-split, time-chunked hand-offs around the persistent BiGRU scans (round 5).

Layer l+1's input projection is out_fwd W_ih[:, :H]^T + out_rev W_ih[:, H:]^T (reference models/rnn.py:17,75: nn.GRU(bidirectional=True)
feeds [out_fwd | out_rev] to the next layer) and each half is final for the frames its direction's scan has passed; in backward the data
gradient is dgx_fwd W_ih_fwd + dgx_rev W_ih_rev.  The scan launches publish progress marks, gate kernels hold the consumer stream, and
m3t_sgemm_window multiplies one direction's half over one time window while the scan is still running.  Checked here:
  * the windowed contraction against fp64 (rows outside the window untouched, accumulate, bias, both B layouts);
  * the chunked schedule against the unchunked one (ops.CHUNKS) on FRESH inputs every iteration at recycled addresses -- a consumer that
    read a window before its producer had written it would read the previous iteration's values -- and bit-identical reruns;
  * that the chunked path really ran (calls counted through the window GEMM's wrapper).  The schedule is OFF by default (measured slower:
    NOTEBOOK.md R5.1) and stays tested as the M3T_SCAN_CHUNKS switch; the start marks of M3T_SCAN_FIRST and the fragments prepared during
    forward are pinned bit-identical to the schedule without them.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32)).to(DEV)


@pytest.mark.parametrize("transB", [1, 0])
def test_sgemm_window_against_fp64(transB):
    from m3t import ops
    rs = np.random.RandomState(5)
    B, T, K, N, ldA, ldC = 32, 300, 512, 1536, 1024, 3072
    A = _rand(rs, B, T, ldA)
    W = _rand(rs, N, 2 * K, scale=0.05) if transB else _rand(rs, K, N, scale=0.05)
    bias = _rand(rs, N)
    a_off, c_off = 512, 1536                                     # a column slice of A (one direction's half), a column block of C
    for (t0, t1) in ((0, 76), (76, 152), (224, 300)):
        Cm = torch.full((B, T, ldC), 7.0, device=DEV)
        with ops.precision("fp32"):
            ops.sgemm_window(transB, B, t1 - t0, T, t0, N, K, A, a_off, ldA, W, K if transB else 0, 2 * K if transB else N, Cm, c_off, ldC,
                             bias=bias)
        a64 = A[:, t0:t1, a_off:a_off + K].double()
        w64 = (W[:, K:2 * K].double().t() if transB else W.double())
        ref = a64 @ w64 + bias.double()
        got = Cm[:, t0:t1, c_off:c_off + N].double()
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        assert err < 2e-6, err
        keep = torch.ones(B, T, ldC, dtype=torch.bool, device=DEV)
        keep[:, t0:t1, c_off:c_off + N] = False
        assert bool((Cm[keep] == 7.0).all()), "rows / columns outside the window were written"
        # second arriver: accumulates onto what is there, no bias
        with ops.precision("fp32"):
            ops.sgemm_window(transB, B, t1 - t0, T, t0, N, K, A, a_off, ldA, W, K if transB else 0, 2 * K if transB else N, Cm, c_off, ldC,
                             accumulate=True)
        got2 = Cm[:, t0:t1, c_off:c_off + N].double()
        err2 = float((got2 - (2 * ref - bias.double())).abs().max()) / float(ref.abs().max())
        assert err2 < 4e-6, err2
    # the six-product form takes the same path
    Cm = torch.zeros(B, T, ldC, device=DEV)
    with ops.precision("x6"):
        ops.sgemm_window(transB, B, 76, T, 76, N, K, A, a_off, ldA, W, K if transB else 0, 2 * K if transB else N, Cm, c_off, ldC)
    ref = A[:, 76:152, a_off:a_off + K].double() @ (W[:, K:2 * K].double().t() if transB else W.double())
    assert float((Cm[:, 76:152, c_off:c_off + N].double() - ref).abs().max()) / float(ref.abs().max()) < 2e-6


def test_sgemm_window_refuses_what_it_cannot_tile():
    from m3t import ops
    from m3t._lib import M3THipError
    A = torch.zeros(32, 300, 512, device=DEV)
    W = torch.zeros(1536, 512, device=DEV)
    Cm = torch.zeros(32, 300, 1536, device=DEV)
    with pytest.raises(M3THipError):
        ops.sgemm_window(1, 32, 75, 300, 0, 1536, 512, A, 0, 512, W, 0, 512, Cm, 0, 1536)       # 32 x 75 rows: no whole 128-row tiles
    with pytest.raises(M3THipError):
        ops.sgemm_window(1, 32, 76, 300, 228, 1536, 512, A, 0, 512, W, 0, 512, Cm, 0, 1536)      # the window leaves the clip


def _stacks(rs, specs, B, T, L=2):
    """specs: [(I, H)] -> ([x], [flat parameter list]) with leaf tensors"""
    xs, prms = [], []
    for I, H in specs:
        xs.append(_rand(rs, B, T, I).requires_grad_(True))
        p = []
        for l in range(L):
            il = I if l == 0 else 2 * H
            for _ in (0, 1):
                p += [_rand(rs, 3 * H, il, scale=(2.0 / (il + H)) ** 0.5).requires_grad_(True),
                      _rand(rs, 3 * H, H, scale=H ** -0.5).requires_grad_(True),
                      _rand(rs, 3 * H, scale=0.1).requires_grad_(True), _rand(rs, 3 * H, scale=0.1).requires_grad_(True)]
        prms.append(p)
    return xs, prms


def _run(ops, xs, prms, douts, cat=None, L=2):
    for t in xs + [q for p in prms for q in p]:
        t.grad = None
    res = ops.multi_bigru([(x, p, L) for x, p in zip(xs, prms)], cat=cat)
    outs = [o for o, _ in res if o.numel()]
    loss = sum((o * d).sum() for o, d in zip(outs, douts))
    loss.backward()
    torch.cuda.synchronize()
    ops.poll_scan_error()
    return [o.detach().clone() for o in outs], [t.grad.detach().clone() for t in xs + [q for p in prms for q in p]]


@pytest.mark.parametrize("specs,cat", [([(128, 256), (256, 512), (256, 512)], (1, 3)),      # the C3 encoder level: audio | gru_v, gru_a
                                       ([(512, 512)], None)])                              # the fusion GRU
def test_chunked_schedule_matches_unchunked_on_fresh_inputs(specs, cat):
    from m3t import ops
    B, T = 32, 300
    saved_mode = ops.CHUNKS[0]
    ops.CHUNKS[0] = True
    assert ops._chunk_bounds(B, T) is not None
    calls = []
    real = ops.sgemm_window_batch
    ops.sgemm_window_batch = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        for it in range(3):
            rs = np.random.RandomState(100 + it)
            xs, prms = _stacks(rs, specs, B, T)
            widths = [2 * specs[0][1]] + ([sum(2 * h for _, h in specs[1:])] if cat else [2 * h for _, h in specs[1:]])
            douts = [_rand(rs, B, T, w, scale=1e-2) for w in widths]
            ops.CHUNKS[0] = True
            n0 = len(calls)
            y1, g1 = _run(ops, xs, prms, douts, cat)
            assert len(calls) > n0, "the chunked path did not run"
            y1b, g1b = _run(ops, xs, prms, douts, cat)
            ops.CHUNKS[0] = False
            n1 = len(calls)
            y0, g0 = _run(ops, xs, prms, douts, cat)
            assert len(calls) == n1
            for a, b in zip(y1 + g1, y1b + g1b):
                assert torch.equal(a, b), "chunked reruns differ: the result depends on timing"
            for a, b in zip(y1, y0):
                assert float((a - b).abs().max()) < 2e-6
            for a, b in zip(g1, g0):
                sc = float(b.abs().max()) + 1e-30
                assert float((a - b).abs().max()) / sc < 2e-5, float((a - b).abs().max()) / sc
    finally:
        ops.CHUNKS[0] = saved_mode
        ops.sgemm_window_batch = real


@pytest.mark.parametrize("specs,cat", [([(128, 256), (256, 512), (256, 512)], (1, 3)), ([(512, 512)], None)])
def test_prepared_fragments_and_scan_first_change_nothing(specs, cat):
    """m3t_gru_bwd_prepare (the backward scans' W_hh fragments written during forward, on an idle stream) and M3T_SCAN_FIRST (the weight
    gradients of level l held back until the scan of level l - 1 is resident: start marks + gate kernels) move work in time only: every
    output and gradient bit-identical to the schedule without them"""
    from m3t import ops
    B, T = 32, 300
    rs = np.random.RandomState(7)
    xs, prms = _stacks(rs, specs, B, T)
    widths = [2 * specs[0][1]] + ([sum(2 * h for _, h in specs[1:])] if cat else [2 * h for _, h in specs[1:]])
    douts = [_rand(rs, B, T, w, scale=1e-2) for w in widths]
    saved = (ops.CHUNKS[0], ops.PREP_AHEAD[0], ops.SCAN_FIRST[0])
    try:
        ops.CHUNKS[0] = False
        res = {}
        for pa in (True, False):
            for sf in (True, False):
                ops.PREP_AHEAD[0], ops.SCAN_FIRST[0] = pa, sf
                res[(pa, sf)] = _run(ops, xs, prms, douts, cat)
        # round 6: the light stack's scans aligned with the heavy level's (forward only by default; M3T_ALIGN_LIGHT=2: backward too / =0: not at
        # all) and its trailing weight gradients on weight-gradient stream 1 (M3T_LIGHT_DW_STREAM1) -- scheduling only, too
        ops.PREP_AHEAD[0], ops.SCAN_FIRST[0] = saved[1], saved[2]
        saved6 = (ops.ALIGN_LIGHT[0], ops._ALIGN_BWD, ops.LIGHT_DW_STREAM1[0])
        try:
            for al, alb, dw1 in ((False, False, False), (True, True, True), (True, False, False)):
                ops.ALIGN_LIGHT[0], ops._ALIGN_BWD, ops.LIGHT_DW_STREAM1[0] = al, alb, dw1
                res[("r6", al, alb, dw1)] = _run(ops, xs, prms, douts, cat)
        finally:
            ops.ALIGN_LIGHT[0], ops._ALIGN_BWD, ops.LIGHT_DW_STREAM1[0] = saved6
        y0, g0 = res[(False, False)]
        for k, (y, g) in res.items():
            for a, b in zip(y + g, y0 + g0):
                assert torch.equal(a, b), k
    finally:
        ops.CHUNKS[0], ops.PREP_AHEAD[0], ops.SCAN_FIRST[0] = saved


def test_weight_magnitude_table_lives_for_one_step():
    """FlatGradDDP.zero_grad() measures every weight matrix once (m3t.ops.measure_weight_amax); forward calls of that step take the slots
    from the table; finish() drops them (the optimizer is about to change the weights) -- and a weight changed IN PLACE between zero_grad and
    forward is the documented exception, so the test pins what is promised: same results with and without the table"""
    from m3t import ops
    from m3t.workloads import AVFeatureGraph, make_c3_step
    from golden.recipe import fill_module, draw
    rs = np.random.RandomState(11)
    B, T = 32, 300
    model = fill_module(AVFeatureGraph(128, 256, 512), 3).to(DEV)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    batch = dict(x_a=f(draw(rs, (B, T, 128))), x_v=f(draw(rs, (B, T, 256))), valence=f(draw(rs, (B, T), "uniform_pm1")),
                 arousal=f(draw(rs, (B, T), "uniform_pm1")), class_expr=f(rs.randint(0, 7, (B, T)).astype(np.int64)),
                 expr_valid=f(rs.uniform(size=(B, T)) < 0.7))
    ddp, step = make_c3_step(model, batch)
    try:
        loss1, _, y1 = step()
        g1 = ddp.flat.clone()
        assert not ops._W_AMAX, "finish() must drop the step's table"
        ddp.zero_grad()
        assert len(ops._W_AMAX) >= 20                       # every W_ih / W_hh / Linear weight of the graph
        ops.drop_weight_amax(ddp)
        ops._W_AMAX_ON_SAVED = ops._W_AMAX_ON
        ops._W_AMAX_ON = False                              # the same step with every call measuring for itself
        loss0, _, y0 = step()
        g0 = ddp.flat.clone()
        torch.cuda.synchronize()
        assert torch.equal(y0, y1) and torch.equal(g0, g1) and float(loss0) == float(loss1)
    finally:
        ops._W_AMAX_ON = getattr(ops, "_W_AMAX_ON_SAVED", True)
        ddp.close()


@pytest.mark.parametrize("M,N,K", [(9600, 1536, 1024), (1280, 512, 2048), (128, 256, 32)])
def test_weight_image_by_lds_dma_is_bit_identical(M, N, K):
    """m3t_sgemm_bimg: the NT product of the input projections (reference models/rnn.py:17: nn.GRU's x W_ih^T) with the weight operand as a
    staged image (m3t_f16x3_image_b) that the 128 x 256 tile kernel fetches by LDS-DMA -- no registers, no conversion, no ds_write for the B
    tile.  Same arithmetic as m3t_sgemm_scaled -- bit-identical where the planner picks the same tile and K passes for both --, bias / accumulate included (NOTEBOOK R5.4b: +0-5 % alone,
    not wired into the step)"""
    from m3t import ops, _lib
    rs = np.random.RandomState(M + N + K)
    A, W, bias = _rand(rs, M, K), _rand(rs, N, K, scale=0.05), _rand(rs, N)
    sl = ops.amax_slots(2, A.device)
    ops.measure_amax([(A, sl.data_ptr()), (W, sl.data_ptr() + 8)])
    img = torch.empty_like(W)
    lib = ops.lib()
    _lib.check(lib.m3t_f16x3_image_b(ops._p(W), N, K, K, ops._p(img), sl.data_ptr() + 8, ops._stream()), "m3t_f16x3_image_b")
    ws = ops.workspace(A.device)
    for accumulate in (False, True):
        C0 = torch.full((M, N), 0.5, device=DEV)
        C1 = torch.full((M, N), 0.5, device=DEV)
        with ops.precision("fp32"):
            ops.sgemm(0, 1, M, N, K, A, 0, K, W, 0, K, C0, 0, N, bias=bias, accumulate=accumulate, amax=(sl.data_ptr(), sl.data_ptr() + 8))
        _lib.check(lib.m3t_sgemm_bimg(M, N, K, ops._p(A), K, ops._p(img), ops._p(C1), N, ops._p(bias), 0, int(accumulate), ops._p(ws),
                                      ws.numel() * 4, sl.data_ptr(), sl.data_ptr() + 8, ops._stream()), "m3t_sgemm_bimg")
        torch.cuda.synchronize()
        if (M, N, K) == (9600, 1536, 1024):          # the planner gives m3t_sgemm_scaled the same tile and one K pass: the same sums in the same order
            assert torch.equal(C0, C1), float((C0 - C1).abs().max())
        else:                                        # (another tile / another number of split-K slabs there: fp32 rounding apart)
            assert float((C0 - C1).abs().max()) <= 2e-6 * float(C0.abs().max())
    ref = A.double() @ W.double().t() + bias.double() + 0.5
    assert float((C1.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    with pytest.raises(_lib.M3THipError):
        _lib.check(lib.m3t_sgemm_bimg(M, N + 64, K, ops._p(A), K, ops._p(img), ops._p(C1), N, None, 0, 0, None, 0, sl.data_ptr(),
                                      sl.data_ptr() + 8, ops._stream()), "m3t_sgemm_bimg")       # N % 256 != 0


@pytest.mark.parametrize("tA,tB,M,N,K,splits,seg", [(0, 1, 9600, 1536, 1024, 1, False), (0, 1, 9472, 1024, 1536, 1, False), (0, 1, 1000, 512, 2048, 1, False),
                                                     (0, 1, 1536, 256, 9600, 8, False), (0, 1, 200, 256, 64, 1, False),
                                                     (1, 0, 1536, 1024, 9600, 10, False), (1, 0, 384, 256, 4800, 3, False), (1, 0, 1536, 512, 9568, 4, True),
                                                     (0, 0, 9600, 1024, 1536, 1, False), (0, 0, 1000, 256, 768, 1, False)])
@pytest.mark.parametrize("variant", [0, 3])
def test_ring_gemm_is_bit_identical(tA, tB, M, N, K, splits, seg, variant):
    """m3t_sgemm_ring (round 6, csrc/gemm_ring.hip): the products of nn.Linear / the GRU input projections and their gradients (reference
    models/rnn.py:17,22-55,75: NT forward, NN data gradient, TN weight gradient, segmented TN for dW_hh) on the 256 x 256 tile kernels whose
    operands reach LDS as raw fp32 by LDS-DMA (a ring of four 16-k stages, counted vmcnt) and are split on the fragment read.  The same
    arithmetic as m3t_sgemm_scaled bit for bit where both take the same K passes (one, or the same slabs) -- for NT the plain loop (variant 0)
    and the pipelined one with the v_fma_mix low terms (variant 3) --, ragged M, bias / accumulate; against fp64 everywhere.  Wide dynamic
    range in A: the low terms' subnormal range is exercised."""
    from m3t import ops, _lib
    if (tA, tB) != (0, 1) and variant != 0:
        pytest.skip("one build of the row-contiguous kernels")
    rs = np.random.RandomState(M + N + K + variant + 2 * tA + tB)
    sg = (299, 300, 1, 0) if seg else (0, 0, 0, 0)
    if seg:
        A, W = _rand(rs, 9600, M), _rand(rs, 9600, 2 * N, scale=0.05)
    else:
        A, W = _rand(rs, *((K, M) if tA else (M, K))), _rand(rs, *((N, K) if tB else (K, N)), scale=0.05)
    bias = _rand(rs, N)
    A = A * torch.exp(torch.from_numpy(rs.standard_normal(tuple(A.shape)).astype(np.float32) * 4.0).to(A.device))      # elements down to 1e-7 of the maximum
    sl = ops.amax_slots(2, A.device)
    ops.measure_amax([(A, sl.data_ptr()), (W, sl.data_ptr() + 8)])
    lib = ops.lib()
    ws = ops.workspace(A.device)
    kern, _ = ops.sgemm_plan(tA, M, N, K, seg_len=sg[0])
    for accumulate in (False, True):
        C0 = torch.full((M, N), 0.5, device=DEV)
        C1 = torch.full((M, N), 0.5, device=DEV)
        C2 = torch.full((M, N), 0.5, device=DEV)
        with ops.precision("fp32"):
            # ONE K pass on both sides (use_ws=False: m3t_sgemm_scaled without split-K slabs): the same sums in the same order
            ops.sgemm(tA, tB, M, N, K, A, 0, A.shape[1], W, 0, W.shape[1], C0, 0, N, bias=bias, accumulate=accumulate, seg=sg, use_ws=False,
                      amax=(sl.data_ptr(), sl.data_ptr() + 8))
        args = (tA, tB, M, N, K, ops._p(A), A.shape[1], ops._p(W), W.shape[1])
        tail = (sg[0], sg[1], sg[2], sg[3], ops._p(ws), ws.numel() * 4)
        _lib.check(lib.m3t_sgemm_ring(*args, ops._p(C1), N, ops._p(bias), 0, int(accumulate), *tail, 1, sl.data_ptr(), sl.data_ptr() + 8, variant,
                                      ops._stream()), "m3t_sgemm_ring")
        _lib.check(lib.m3t_sgemm_ring(*args, ops._p(C2), N, ops._p(bias), 0, int(accumulate), *tail, splits, sl.data_ptr(), sl.data_ptr() + 8, variant,
                                      ops._stream()), "m3t_sgemm_ring")
        torch.cuda.synchronize()
        if kern == 1 and M % 128 == 0:      # m3t_sgemm_scaled ran an fp16x3 tile kernel
            assert torch.equal(C0, C1), float((C0 - C1).abs().max())
        else:
            assert float((C0 - C1).abs().max()) <= 4e-6 * float(C0.abs().max())
        assert float((C0 - C2).abs().max()) <= 4e-6 * float(C0.abs().max())          # deterministic slabs: fp32 rounding apart from the single pass
        C1 = C2
    if seg:
        Ad = torch.cat([A[c * 300 + 1:c * 300 + 300] for c in range(32)]).double()
        Wd = torch.cat([W[c * 300:c * 300 + 299, :N] for c in range(32)]).double()
        ref = Ad.t() @ Wd + bias.double() + 0.5
    else:
        ref = (A.double().t() if tA else A.double()) @ (W.double().t() if tB else W.double()) + bias.double() + 0.5
    assert float((C1.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    with pytest.raises(_lib.M3THipError):
        _lib.check(lib.m3t_sgemm_ring(tA, tB, M, N + 64, K, ops._p(A), A.shape[1], ops._p(W), W.shape[1], ops._p(C1), N, None, 0, 0, 0, 0, 0, 0,
                                      None, 0, 1, sl.data_ptr(), sl.data_ptr() + 8, 0, ops._stream()), "m3t_sgemm_ring")       # N % 256 != 0
