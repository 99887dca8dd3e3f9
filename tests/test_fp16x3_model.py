"""CPU model of the fp16x3 product (M3T_GEMM_F16X3, DESIGN.md section 7 / NOTEBOOK.md section 5e): numpy restatement of what the GEMM kernels do to their
operands -- power-of-two scale from the operand's largest magnitude (m3t_f16_scale in csrc/common.h), two fp16 terms, three exact
products, fp32 accumulation per 16-deep MFMA -- against fp64, next to the same model of the six-product bf16 form and of a sequential
fp32 FMA chain.  The GPU tests check the kernels; this pins the ARITHMETIC the design relies on, on the CPU."""
import numpy as np


def f16_scale(amax):
    """(scale, 1 / scale) as m3t_f16_scale computes them from the bit pattern of max |x|"""
    bits = np.float32(amax).view(np.uint32)
    e = int((bits >> 23) & 0xFF)
    es = min(max(268 - e, 1), 254)
    s = np.uint32(es << 23).view(np.float32)
    inv = np.uint32((254 - es) << 23).view(np.float32)
    return np.float32(s), np.float32(inv)


def split_f16(x, s):
    xs = (x * s).astype(np.float32)
    hi = xs.astype(np.float16).astype(np.float32)
    lo = (xs - hi).astype(np.float32).astype(np.float16).astype(np.float32)
    return hi, lo


def bf16_rn(x):
    u = x.view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split_bf16(x):
    a1 = bf16_rn(x); r1 = (x - a1).astype(np.float32)
    a2 = bf16_rn(r1); r2 = (r1 - a2).astype(np.float32)
    return a1, a2, bf16_rn(r2)


def mfma_acc(terms, kstep=16):
    """fp32 accumulator, one exact product block of `kstep` k per MFMA, the products of `terms` in the listed order"""
    A0, B0 = terms[0]
    acc = np.zeros((A0.shape[0], B0.shape[1]), np.float32)
    for k0 in range(0, A0.shape[1], kstep):
        for A, B in terms:
            acc = (acc + A[:, k0:k0 + kstep].astype(np.float64) @ B[k0:k0 + kstep].astype(np.float64)).astype(np.float32)
    return acc


def rel(x, ref):
    return float(np.linalg.norm(x - ref) / np.linalg.norm(ref))


def test_scale_puts_the_maximum_below_fp16_range_and_handles_the_edges():
    for amax in (1.0, 3.7e-9, 80.0, 1e30, 65504.0, 2.0 ** -100, np.float32(1.0) - np.float32(2 ** -24)):
        s, inv = f16_scale(amax)
        assert 2.0 ** 14 <= float(np.float32(amax) * s) < 2.0 ** 15, (amax, s)
        assert float(s) * float(inv) == 1.0
    s, inv = f16_scale(0.0)                      # an all-zero operand: the scale saturates, its inverse flushes to zero
    assert np.isfinite(s) and float(inv) == 0.0
    s, inv = f16_scale(2.0 ** -140)              # denormal maximum: same
    assert np.isfinite(s) and float(inv) == 0.0


def test_three_fp16_products_are_no_less_accurate_than_six_bf16_products():
    rs = np.random.RandomState(0)
    for K, sa, sb in ((1024, 1.0, 1.0), (4096, 1e-6, 0.03), (1536, 80.0, 0.05)):
        M = N = 48
        A = (rs.standard_normal((M, K)) * sa).astype(np.float32)
        A *= (10.0 ** rs.uniform(-4, 0, (M, 1))).astype(np.float32)          # rows four orders of magnitude apart
        B = (rs.standard_normal((K, N)) * sb).astype(np.float32)
        ref = A.astype(np.float64) @ B.astype(np.float64)
        s_a, i_a = f16_scale(np.abs(A).max()); s_b, i_b = f16_scale(np.abs(B).max())
        ah, al = split_f16(A, s_a); bh, bl = split_f16(B, s_b)
        h3 = mfma_acc([(al, bh), (ah, bl), (ah, bh)]) * i_a * i_b
        a = split_bf16(A); b = split_bf16(B)
        x6 = mfma_acc([(a[2], b[0]), (a[1], b[1]), (a[0], b[2]), (a[1], b[0]), (a[0], b[1]), (a[0], b[0])])
        seq = np.zeros((M, N), np.float32)
        for k in range(K):
            seq = (seq + A[:, k:k + 1] * B[k:k + 1]).astype(np.float32)
        e3, e6, es = rel(h3, ref), rel(x6, ref), rel(seq, ref)
        rows = (np.linalg.norm(h3 - ref, axis=1) / np.linalg.norm(ref, axis=1)).max()
        assert e3 <= 1.2 * e6 and e3 <= es and e3 <= 2e-6 and rows <= 4e-6, (K, sa, sb, e3, e6, es, rows)


def test_elements_far_below_the_maximum_keep_an_absolute_not_a_relative_error():
    """the stated limit of the mode: an element 2^-r below its operand's maximum keeps 22 bits while its low term is a normal fp16
    number and an absolute error of ~2^-40 of the maximum below that"""
    x = np.float32(1.2345678)
    for r, bits in ((0, 21), (10, 21), (17, 20), (25, 13), (30, 8)):
        v = np.array([x, x * np.float32(2.0 ** -r)], np.float32)
        s, inv = f16_scale(np.abs(v).max())
        hi, lo = split_f16(v, s)
        back = (hi.astype(np.float64) + lo.astype(np.float64)) * float(inv)
        err = abs(back[1] - float(v[1])) / float(v[1])
        assert err <= 2.0 ** -bits, (r, err)


def test_the_per_cell_bound_of_the_producer_split_backward_scan_holds():
    """gru_persist_bwd3q_kernel scales what a cell publishes -- (dr~, dz~, dn~ r) -- by a power of two taken from
    |dht| max(1, |W_hn h + b_hn| / 4), known before the rest of the cell math (csrc/gru_persist_bwd3q_step.inc); the cell formulas are
    those of gru_cell_bwd (csrc/gru_common.h).  In fp32, over wide ranges, no published value exceeds the bound by more than rounding --
    the scale leaves a factor two."""
    rs = np.random.RandomState(3)
    n_ = 200000
    f = np.float32
    r = (1.0 / (1.0 + np.exp(-rs.standard_normal(n_) * 4))).astype(f)
    z = (1.0 / (1.0 + np.exp(-rs.standard_normal(n_) * 4))).astype(f)
    n = np.tanh(rs.standard_normal(n_) * 2).astype(f)
    hprev = np.tanh(rs.standard_normal(n_) * 2).astype(f)
    ghn = (rs.standard_normal(n_) * (10.0 ** rs.uniform(-2, 2, n_))).astype(f)
    dht = (rs.standard_normal(n_) * (10.0 ** rs.uniform(-12, 6, n_))).astype(f)
    dn = (dht * (f(1) - z) * (f(1) - n * n)).astype(f)
    dz = (dht * (hprev - n) * z * (f(1) - z)).astype(f)
    dr = (dn * ghn * r * (f(1) - r)).astype(f)
    dnr = (dn * r).astype(f)
    bound = (np.abs(dht) * np.maximum(f(1), f(0.25) * np.abs(ghn))).astype(f)
    worst = np.maximum(np.abs(dr), np.maximum(np.abs(dz), np.abs(dnr))) / np.maximum(bound, f(1e-37))
    assert float(worst.max()) <= 1.0 + 1e-5, float(worst.max())
    # and with the scale taken from the bound, every published value fits fp16 with room to spare
    for i in rs.randint(0, n_, 2000):
        s, _ = f16_scale(bound[i])
        assert max(abs(float(dr[i])), abs(float(dz[i])), abs(float(dnr[i]))) * float(s) < 40000.0
