#!/usr/bin/env python3
"""The 256 x 256 LDS-DMA ring kernel (m3t_sgemm_ring, csrc/gemm_ring.hip) against m3t_sgemm_scaled: bit-identity and time, interleaved rounds in
one process (guide 5.4 rule 24).  python tools/ring_bench.py [variants, comma separated] [rounds]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from m3t import ops, _lib

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
lib = ops.lib()
shapes = [(0, 1, 9600, 1536, 1024), (0, 1, 9600, 1536, 512), (0, 1, 9600, 512, 2048), (0, 1, 8192, 2048, 2048),
          (1, 0, 1536, 1024, 9600), (1, 0, 1536, 512, 9600), (1, 0, 1536, 256, 9600), (1, 0, 768, 512, 9600), (0, 0, 9600, 1024, 1536), (0, 0, 9600, 512, 1536),
          (0, 0, 9600, 256, 1536), (1, 0, 1536, 512, 9568, "seg")]
if os.environ.get("RING_SHAPES"):
    shapes = [tuple(int(v) if v != "seg" else v for v in s.split("x")) for s in os.environ["RING_SHAPES"].split(",")]


def digest(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:12]


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


torch.manual_seed(1)
for shp in shapes:
    tA, tB, m, n, k = shp[:5]
    seg = (299, 300, 1, 0) if len(shp) > 5 else (0, 0, 0, 0)       # dW_hh = sum_t dgh_t^T h_{t-1}: 32 clips x 299 rows, shifted by one row
    A = torch.randn((k, m) if tA else (m, k), device=dev)
    Bm = torch.randn((n, k) if tB else (k, n), device=dev) * 0.05
    if seg[0]:
        A = torch.randn(9600, m, device=dev); Bm = torch.randn(9600, 2 * n, device=dev) * 0.05
    bias = torch.randn(n, device=dev)
    sl = ops.amax_slots(2, dev)
    ops.measure_amax([(A, sl.data_ptr()), (Bm, sl.data_ptr() + 8)])
    Cr = torch.empty(m, n, device=dev); Cn = torch.full((m, n), float("nan"), device=dev)
    ws = ops.workspace(dev)
    ref = lambda: ops.sgemm(tA, tB, m, n, k, A, 0, A.shape[1], Bm, 0, Bm.shape[1], Cr, 0, n, bias=bias, seg=seg, amax=(sl.data_ptr(), sl.data_ptr() + 8))

    def ring(var, splits=1):
        _lib.check(lib.m3t_sgemm_ring(tA, tB, m, n, k, ops._p(A), A.shape[1], ops._p(Bm), Bm.shape[1], ops._p(Cn), n, ops._p(bias), 0, 0,
                                      seg[0], seg[1], seg[2], seg[3], ops._p(ws), ws.numel() * 4, splits, sl.data_ptr(), sl.data_ptr() + 8,
                                      var if (tA, tB) == (0, 1) else 0, ops._stream()), "m3t_sgemm_ring")
    ref(); torch.cuda.synchronize()
    dr = digest(Cr)
    kern, spl = ops.sgemm_plan(tA, m, n, k, seg_len=seg[0], ws_bytes=ws.numel() * 4)
    line = "tA%d tB%d M%5d N%5d K%5d%s ref(splits %d)" % (tA, tB, m, n, k, " seg" if seg[0] else "", spl)
    ok = {}
    for v in variants:
        Cn.fill_(float("nan"))
        ring(v, spl); torch.cuda.synchronize()
        ok[v] = digest(Cn) == dr
        if not ok[v]:
            d = (Cn.double() - Cr.double()).abs()
            print("   variant %d differs: max |d| %.3e at %s, nan %d, ref max %.3e" % (v, float(torch.nan_to_num(d, nan=1e30).max()),
                  tuple(int(i) for i in torch.nonzero(torch.nan_to_num(d, nan=1e30) == torch.nan_to_num(d, nan=1e30).max())[0]),
                  int(torch.isnan(Cn).sum()), float(Cr.abs().max())))
    reps = 20 if m * n * k > 1e9 else 5
    rsplits = int(os.environ.get("RING_SPLITS", spl if tA else 1))
    tr = []; tv = {v: [] for v in variants}
    for _ in range(rounds):
        tr.append(timed(ref, reps))
        for v in variants:
            tv[v].append(timed(lambda: ring(v, rsplits), reps))
    fl = 2.0 * m * n * k
    line += " %7.1f us %5.0f TF | ring splits %d:" % (min(tr), fl / min(tr) / 1e6, rsplits)
    for v in variants:
        line += " v%d %s %7.1f us (med %7.1f) %5.0f TF |" % (v, "bit-identical" if ok[v] else "DIFFERS", min(tv[v]), sorted(tv[v])[len(tv[v]) // 2],
                                                             fl / min(tv[v]) / 1e6)
    print(line, flush=True)
