#!/usr/bin/env python3
"""Micro-benchmark of the grouped GRU scan launches through the C ABI (per level of the C3 graph)."""
import os
import sys
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "m3f.pytorch_amd"))
import torch
from m3t import _lib, ops
from m3t._lib import GruFwdDesc, GruBwdDesc

dev = "cuda:0"
B, T = int(os.environ.get("B", 32)), int(os.environ.get("T", 300))


def fwd_group(Hs):
    descs, keep = [], []
    for H in Hs:
        xproj = torch.randn(B, T, 6 * H, device=dev) * 0.5
        out = torch.empty(B, T, 2 * H, device=dev)
        gates = torch.empty(2, B, T, 4 * H, device=dev)
        for d in (0, 1):
            w = torch.randn(3 * H, H, device=dev) / H ** 0.5
            b = torch.zeros(3 * H, device=dev)
            keep += [w, b]
            descs.append(GruFwdDesc(xproj.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(),
                                    gates.data_ptr() + 4 * d * B * T * 4 * H, None, H, d, 6 * H, d * 3 * H, 2 * H, d * H))
        keep += [xproj, out, gates]
    return descs, keep


def bwd_group(Hs):
    descs, keep = [], []
    for H in Hs:
        dout = torch.randn(B, T, 2 * H, device=dev) * 0.1
        out = torch.randn(B, T, 2 * H, device=dev) * 0.3
        gates = torch.rand(2, B, T, 4 * H, device=dev) * 0.8 + 0.1
        dgx = torch.empty(B, T, 6 * H, device=dev)
        dgh = torch.empty(2, B, T, 3 * H, device=dev)
        dh = torch.empty(2, B, H, device=dev)
        for d in (0, 1):
            wt = torch.randn(H, 3 * H, device=dev) / H ** 0.5
            keep.append(wt)
            descs.append(GruBwdDesc(dout.data_ptr(), out.data_ptr(), gates.data_ptr() + 4 * d * B * T * 4 * H, wt.data_ptr(), None,
                                    dgx.data_ptr(), dgh.data_ptr() + 4 * d * B * T * 3 * H, dh.data_ptr() + 4 * d * B * H, None, None, None,
                                    H, d, 2 * H, d * H, 6 * H, d * 3 * H))
        keep += [dout, out, gates, dgx, dgh, dh]
    return descs, keep


def time_call(fn, descs, cls, reps=5):
    arr = (cls * len(descs))(*descs)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = ops.workspace(torch.device(dev)) if os.environ.get("NOWS") is None else None
    wsp, wsb = (C.c_void_p(ws.data_ptr()), ws.numel() * 4) if ws is not None else (None, 0)
    flags = int(os.environ.get('SCAN_FLAGS', '0'))
    rc = fn(arr, len(descs), B, T, wsp, wsb, flags, s)
    assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(arr, len(descs), B, T, wsp, wsb, flags, s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps / T * 1e3     # us per step launch
    if os.environ.get("M3T_SCAN_PROF") in ("1", "2"):
        buf = (C.c_ulonglong * 6)()
        if lib.m3t_gru_persist_profile(buf) == 0:
            tot = float(sum(buf)) or 1.0
            names = ["top", "gather", "mfma", "barrier", "gates+publish", "stores"]
            print("      stamps (share of the step, x %.2f us): " % us + "  ".join("%s %.0f%%" % (n, 100.0 * v / tot) for n, v in zip(names, buf)))
    return us


lib = _lib.load()
levels = {"enc (4x512+2x256)": [512, 512, 256], "fusion (2x512)": [512], "scorers (4x128)": [128, 128],
          "one 512 pair": [512], "4x512": [512, 512], "8x512": [512, 512, 512, 512], "2x256": [256], "tiny 2x16": [16], "2x64": [64]}
print("B=%d T=%d" % (B, T))
only = os.environ.get('ONLY')
for name, Hs in levels.items():
    if only and name not in only.split(","):
        continue
    d, k = fwd_group(Hs)
    tf = time_call(lib.m3t_gru_scan_fwd, d, GruFwdDesc)
    flops = sum(2.0 * B * 3 * H * H * 2 for H in Hs)
    d2, k2 = bwd_group(Hs)
    tb = time_call(lib.m3t_gru_scan_bwd, d2, GruBwdDesc)
    print("%-22s fwd %6.2f us/step (%5.1f TF/s)   bwd %6.2f us/step (%5.1f TF/s)" % (name, tf, flops / tf / 1e6, tb, flops / tb / 1e6))
