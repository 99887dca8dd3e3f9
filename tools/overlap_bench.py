#!/usr/bin/env python3
"""Does a weight-gradient GEMM chain hide beside a persistent scan?  Times (a) the 4x512 backward scan alone,
(b) a chain of TN GEMMs (dW shapes) alone, (c) both at once on two streams."""
import os
import sys
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from m3t import _lib, ops
from m3t._lib import GruBwdDesc

os.environ.setdefault("ONLY", "none")
import scan_bench as sb      # noqa: E402  (reuses bwd_group; its module-level loop is skipped by ONLY=none)

lib = _lib.load()
dev = torch.device("cuda:0")
B, T = sb.B, sb.T
FWD = os.environ.get("SCAN") == "fwd"          # SCAN=fwd: the forward scan (124 VGPRs per wave) instead of the backward one (144)
descs, keep = (sb.fwd_group if FWD else sb.bwd_group)([512, 512])
arr = ((_lib.GruFwdDesc if FWD else GruBwdDesc) * len(descs))(*descs)
ws = ops.workspace(dev)
side = torch.cuda.Stream()
M = B * T
dy = torch.randn(M, 1536, device=dev)
x = torch.randn(M, 1024, device=dev)
dw = torch.empty(1536, 1024, device=dev)
ws2 = torch.empty(64 << 18, device=dev)


def scan(stream):
    with torch.cuda.stream(stream):
        rc = (lib.m3t_gru_scan_fwd if FWD else lib.m3t_gru_scan_bwd)(arr, len(descs), B, T, C.c_void_p(ws.data_ptr()), ws.numel() * 4, 0,
                                                                     C.c_void_p(stream.cuda_stream))
        assert rc == 0


def gemms(stream, n):
    with torch.cuda.stream(stream):
        for _ in range(n):
            rc = lib.m3t_sgemm(1, 0, 1536, 1024, M, C.c_void_p(dy.data_ptr()), 1536, C.c_void_p(x.data_ptr()), 1024,
                               C.c_void_p(dw.data_ptr()), 1024, None, 0, 0, 0, 0, 0, 0, C.c_void_p(ws2.data_ptr()), ws2.numel() * 4, 0,
                               C.c_void_p(stream.cuda_stream))
            assert rc == 0


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


main = torch.cuda.current_stream()
NG = 6
t_scan = timed(lambda: scan(main))
t_gemm = timed(lambda: gemms(main, NG))


def both():
    side.wait_stream(main)
    scan(main)
    gemms(side, NG)
    main.wait_stream(side)


t_both = timed(both)
print("scan alone %.3f ms | %d dW GEMMs alone %.3f ms | together %.3f ms (sum %.3f)" % (t_scan, NG, t_gemm, t_both, t_scan + t_gemm))
