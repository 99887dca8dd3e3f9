"""torch.autograd Functions over the C ABI of libm3t_hip.so.

torch is plumbing here: device memory (caching allocator), the current HIP stream and
autograd bookkeeping.  All arithmetic of the hot path happens in the HIP library; there
is no CPU or eager fallback -- a CPU tensor or a missing library raises.
"""
import ctypes as C
import os
import sys

import torch

from . import _lib
from ._lib import GruFwdDesc, GruBwdDesc, M3T_MAX_SCANS, M3THipError

_WS = {}
_WS_MIN = 64 << 20

# Optional launch-stream timing (bench.py): when PROFILE_ON[0] is set, every scan / GEMM call is
# bracketed by HIP events on the stream the kernels are launched on.
PROFILE = []

# ---- gradient sinks: where a parameter's gradient lives (a view of the flat gradient buffer of m3t.ddp.FlatGradDDP).
# A backward that finds a sink for a parameter writes the gradient THERE and returns None for it, so autograd launches no
# AccumulateGrad add (48 small kernels per C3 step).  Only the first gradient of a parameter per step goes this way (a
# second use of the same parameter falls back to a returned tensor, which autograd adds onto it); a sink is only used
# while the registered parameter object is alive and its .grad still IS the registered view (someone who resets the
# grads to None gets the ordinary path).  Armed per step by FlatGradDDP.zero_grad(), after it has zeroed the buffer.
import weakref

_GRAD_SINKS = {}            # id(parameter) -> [weakref(parameter), view, armed, id(owner)]


def register_grad_sink(param, view, owner=None):
    """owner: the FlatGradDDP instance the view belongs to.  Several owners coexist (a student and an EMA teacher, two
    trainers in one process): each arms and clears only its own entries."""
    _GRAD_SINKS[id(param)] = [weakref.ref(param), view, False, id(owner) if owner is not None else None]


def clear_grad_sinks(owner=None):
    """owner=None: every sink of the process; else only that owner's"""
    if owner is None:
        _GRAD_SINKS.clear()
        return
    for k in [k for k, e in _GRAD_SINKS.items() if e[3] == id(owner)]:
        del _GRAD_SINKS[k]


def arm_grad_sinks(owner=None):
    for e in _GRAD_SINKS.values():
        if owner is None or e[3] == id(owner):
            e[2] = True


def _take_sink(param):
    """the sink view for the parameter object `param` if this is its first gradient of the step, else None"""
    e = _GRAD_SINKS.get(id(param))
    if e is None or not e[2]:
        return None
    ref = e[0]()
    if ref is not param or param.grad is None or param.grad.data_ptr() != e[1].data_ptr() or param.grad.shape != param.shape:
        return None
    e[2] = False
    return e[1]
PROFILE_ON = [False]
PROFILE_GEMM = [False]      # ~90 extra event pairs per step: off unless asked for


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL = _Null()


class _Timed:
    """HIP-event timing of a launch on the current stream (bench.py).  kernel_side=True: the events are handed to the C
    call (m3t_gru_scan_events), which records them right around its scan kernel(s) instead of around the whole call."""

    def __init__(self, kernel, launches, flops, kernel_side=False, nbytes=0):
        """nbytes: algorithmic HBM bytes of an HBM-bound launch (bench.py reports bytes / time against the HBM peak)"""
        self.rec = None
        self.kernel_side = kernel_side
        if PROFILE_ON[0]:
            self.rec = {"kernel": kernel, "launches": launches, "flops": float(flops), "bytes": float(nbytes),
                        "start": torch.cuda.Event(enable_timing=True), "end": torch.cuda.Event(enable_timing=True)}

    def __enter__(self):
        if self.rec is not None:
            st = cur_stream()
            self.rec["start"].record(st)
            if self.kernel_side:
                self.rec["end"].record(st)          # materialises the handle; the C call records both again
                _lib.check(lib().m3t_gru_scan_events(C.c_void_p(self.rec["start"].cuda_event), C.c_void_p(self.rec["end"].cuda_event)),
                           "m3t_gru_scan_events")
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            if not self.kernel_side:
                self.rec["end"].record(cur_stream())
            PROFILE.append(self.rec)
        return False


def lib():
    return _lib.load()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)       # the current stream's handle without building a Stream object


_HOST_SLOW = os.environ.get("M3T_HOST_FAST", "1") == "0"      # A/B (tools/host_profile.py): torch's own device / stream helpers and stream context
_CUR_DEV = (getattr(torch._C, "_cuda_getDevice", None) if not _HOST_SLOW else None) or torch.cuda.current_device      # (torch.cuda.current_device() walks _lazy_init: ~3 us a call)


def _stream():
    """the calling thread's current HIP stream as a C pointer (every launch of the path passes it: ~600 calls per training step)"""
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(_CUR_DEV()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_SET_STREAM = getattr(torch._C, "_cuda_setStream", None)


class on_stream:
    """`with on_stream(st)` for the schedule's ~35 stream switches per step: torch's StreamContext builds Stream objects and walks the
    device-index helpers on entry and exit (~20 us a switch); this one keeps the previous stream's ids and makes the two C calls"""
    __slots__ = ("st", "prev")

    def __init__(self, st):
        self.st = st

    def __enter__(self):
        st = self.st
        if st is None:
            return None
        if _SET_STREAM is None or st.device_index != _CUR_DEV():
            # another device than the current one: _cuda_setStream would switch the device and leave it switched -- torch's
            # own context restores both
            self.prev = torch.cuda.stream(st)
            self.prev.__enter__()
            return st
        self.prev = cur_stream(st.device)
        _SET_STREAM(stream_id=st.stream_id, device_index=st.device_index, device_type=st.device_type)
        return st

    def __exit__(self, *exc):
        if self.st is None:
            return False
        p = self.prev
        if isinstance(p, torch.cuda.StreamContext):
            p.__exit__(*exc)
        else:
            _SET_STREAM(stream_id=p.stream_id, device_index=p.device_index, device_type=p.device_type)
        return False


if _HOST_SLOW:
    on_stream = torch.cuda.stream


_STREAM_OBJ = {}      # (device index, raw handle) -> torch.cuda.Stream: the object is kept, so the handle stays this stream's


def cur_stream(device=None):
    """cur_stream(device) without building a Stream object per call (~15 us each through torch's device-index helpers;
    the schedule asks ~50 times per step for wait_stream / record / record_stream)"""
    if _RAW_STREAM is None or _HOST_SLOW:
        return torch.cuda.current_stream(device)
    idx = device.index if (device is not None and getattr(device, "index", None) is not None) else (device if isinstance(device, int) else _CUR_DEV())
    h = _RAW_STREAM(idx)
    st = _STREAM_OBJ.get((idx, h))
    if st is None:
        st = _STREAM_OBJ[(idx, h)] = torch.cuda.current_stream(idx)
    return st


def _req(t, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise M3THipError("%s must be a device (HIP) tensor: the M3T hot path has no CPU fallback" % name)
    if t.dtype != torch.float32:
        raise M3THipError("%s must be float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise M3THipError("%s must be contiguous" % name)
    return t


def _p(t, off=0):
    """device address of element `off` of a float tensor as a plain int (every pointer parameter of the C ABI is declared c_void_p in
    m3t._lib: ctypes converts ints itself; building a c_void_p object per argument was ~350 objects per C3 step)"""
    if t is None:
        return None
    return t.data_ptr() + 4 * off


# ---- streams of the step's schedule --------------------------------------------------------------------------------
# Round 3 retired the A/B switches of rounds 1-2 whose outcome is decided (NOTEBOOK.md section 5c lists each with its measured
# result): the schedule below is THE schedule.  What is left as a process-wide switch is listed in README.md (<= 10).
#   * side stream: independent light-weight stacks run beside the main chain (_stream_groups); with two H-groups of persistent
#     scans (gru_v|gru_a at H=512, audio at H=256) the light group's scans and GEMMs go there, fenced by events so that no two
#     persistent scans ever overlap, and run beside the heavy group's GEMMs; two-layer stacks run the light group's two scans
#     back to back between the heavy group's (forward H0 L0 L1 H1, backward H1 L1 H0 L0: the pass ends with a light scan);
#   * weight-gradient streams: the weight-gradient GEMMs of a GRU level, of nn.Linear and of the TemporalBlocks feed nothing but
#     the optimizer: they leave the scan -> dX -> scan chain for streams of their own, start when the level's data gradients
#     are done, and are joined in FlatGradDDP.finish() when every gradient went into a gradient sink.  Two streams; the second
#     one takes half of the LAST level's GEMMs only (the tail of backward, when no scan and no data gradient is left).
# M3T_LIGHT_DW_STREAM1=0: the light stack's trailing weight gradients on weight-gradient stream 0 behind the heavy levels' (as until round 6)
LIGHT_DW_STREAM1 = [os.environ.get("M3T_LIGHT_DW_STREAM1", "1") != "0"]
# M3T_ALIGN_LIGHT=0: the light stack's deeper FORWARD scans start as soon as their own inputs are ready (beside the heavy level's GEMMs, as until round 6)
ALIGN_LIGHT = [os.environ.get("M3T_ALIGN_LIGHT", "1") != "0"]
_ALIGN_BWD = os.environ.get("M3T_ALIGN_LIGHT", "1") == "2"      # "2": in backward too (measured: forward only 11.74, both 11.77, backward only 11.92 ms)
_SIDE = {}
_PERSIST_ENABLED = os.environ.get("M3T_SCAN_PERSIST", "1") != "0"
_WGRAD = {}
_N_WGRAD = 2


def _wait_slot_fill(st, key):
    """a stream created AFTER the live chunk of the magnitude-slot pool was zero-filled is ordered behind that fill (amax_slots only
    made the streams that existed then wait)"""
    pool = _SLOT_POOL.get(key)
    if pool is not None and len(pool) > 2:
        st.wait_event(pool[2])


def side_stream(device):
    key = (device.type, device.index)
    st = _SIDE.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE[key] = st
        _ROLE_OF_HANDLE[(st.device.index, st.cuda_stream)] = "side"
        _wait_slot_fill(st, key)
    return st


_WGRAD_PENDING = {}


def join_wgrad(device=None):
    """make the current stream wait for weight-gradient GEMMs still running on their own stream (left unjoined by a GRU
    backward whose gradients all went into gradient sinks); FlatGradDDP.finish() calls it before it touches the buffer"""
    for key, pending in list(_WGRAD_PENDING.items()):
        if pending and (device is None or key == (device.type, device.index)):
            for st in _WGRAD.get(key) or []:
                cur_stream(torch.device(key[0], key[1])).wait_stream(st)
            _WGRAD_PENDING[key] = False


def wgrad_joined(device):
    """the caller has made its stream wait for every live stream of `device` itself (FlatGradDDP.finish): nothing is pending any more"""
    _WGRAD_PENDING[(device.type, device.index)] = False


def wgrad_stream(device, i=0):
    key = (device.type, device.index)
    sts = _WGRAD.get(key)
    if sts is None:
        sts = [torch.cuda.Stream(device=device) for _ in range(_N_WGRAD)]
        _WGRAD[key] = sts
        for i_, st_ in enumerate(sts):
            _ROLE_OF_HANDLE[(st_.device.index, st_.cuda_stream)] = "wgrad%d" % i_
            _wait_slot_fill(st_, key)
    return sts[i % min(2, len(sts))]          # (streams 2.. only ever take tail GEMMs, see _MultiBiGRU.backward.level_dw)


def wgrad_streams(device):
    wgrad_stream(device)
    return _WGRAD[(device.type, device.index)]


def live_streams(device):
    """the library's own streams that exist on `device` (side stream, weight-gradient streams) -- without creating any: a caller that is about
    to read what backward wrote (an overlapped gradient all-reduce) makes its stream wait for them"""
    key = (device.type, device.index)
    return ([_SIDE[key]] if key in _SIDE else []) + list(_WGRAD.get(key) or [])


_ROLE_OF_HANDLE = {}      # (device index, raw stream handle) -> "side" / "wgrad<i>": filled when the streams are created


def _ws_tag(device):
    """role of the calling thread's current stream on `device` (called once per GEMM / scan: raw handles, no Stream objects)"""
    if _RAW_STREAM is not None and device.index is not None:
        return _ROLE_OF_HANDLE.get((device.index, _RAW_STREAM(device.index)), "main")
    cur = cur_stream(device)
    st = _SIDE.get((device.type, device.index))
    if st is not None and cur == st:
        return "side"
    for i, st in enumerate(_WGRAD.get((device.type, device.index)) or []):
        if cur == st:
            return "wgrad%d" % i
    return "main"


def workspace(device, nbytes=_WS_MIN, tag=None):
    """Per-device, per-stream scratch (split-K slabs, partial sums, scan fragments).  One buffer per stream role
    (main / side / wgrad), so stream order serialises reuse."""
    key = (device.type, device.index, tag or _ws_tag(device))
    ws = _WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(max(nbytes, _WS_MIN) // 4, dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


# ----------------------------------------------------------------------------- raw wrappers
# ---- arithmetic mode of the dense contractions --------------------------------------------------------------------
# "fp32" (default): fp32-accurate everywhere (what the reference computes): the GEMMs and implicit-GEMM convolutions form every
# product from two fp16 terms per (scaled) operand -- three MFMAs, M3T_GEMM_F16X3, DESIGN.md section 7 / NOTEBOOK.md section 5e -- the recurrent scans from
# three bf16 terms (six MFMAs).  "x6": the GEMMs and convolutions on the six-product bf16 form too (the default until round 3; the
# library's own default when called with flags = 0; env M3T_GEMM_F16X3=0 forces it).  "high": opt-in, the GEMMs and convolutions
# treat each fp32 operand as the sum of two bfloat16 numbers (four products, ~2^-16 relative error per term -- what
# torch.set_float32_matmul_precision('high') means); the recurrent scans stay fp32-accurate.  "bf16": BASELINE.json config C2 -- every
# matmul / conv / recurrent-product operand (activation, weight, gradient) is rounded to bf16, accumulation, state,
# biases, gate math, normalisation and the loss stay fp32, parameters stay fp32 ("master weights").  An autograd
# Function records the mode of its forward and uses it for its backward.
_PREC = [_lib.M3T_GEMM_F16X3]


class precision:
    """`with ops.precision("bf16"): y = model(x)` -- context manager selecting the arithmetic mode."""

    def __init__(self, mode):
        if mode not in ("fp32", "x6", "high", "bf16"):
            raise ValueError("precision mode must be 'fp32', 'x6', 'high' or 'bf16'")
        self.flag = {"fp32": _lib.M3T_GEMM_F16X3, "x6": 0, "high": _lib.M3T_GEMM_HIGH, "bf16": _lib.M3T_BF16}[mode]

    def __enter__(self):
        self.prev = _PREC[0]
        _PREC[0] = self.flag
        return self

    def __exit__(self, *exc):
        _PREC[0] = self.prev
        return False


def sgemm(transA, transB, M, N, K, A, a_off, lda, B, b_off, ldb, Cm, c_off, ldc, bias=None, act=0,
          accumulate=False, seg=(0, 0, 0, 0), use_ws=True, prec=None, exclusive=False, amax=(None, None)):
    """amax = (slot of A, slot of B): device addresses of magnitude slots for the fp16x3 products (amax_slots / amax_one /
    measure_amax below, or a backward scan's); None: the library measures that operand itself (one more launch)"""
    ws = workspace(Cm.device) if use_ws else None
    flags = (_PREC[0] if prec is None else prec) | (_lib.M3T_GEMM_EXCLUSIVE if exclusive else 0)
    if _FENCED[0]:
        flags |= _lib.M3T_GEMM_BESIDE_SCAN       # issued inside the interleaved schedule of _MultiBiGRU: scans of another stream run beside it
    with _Timed("sgemm_kernel", 1, 2.0 * M * N * K) if PROFILE_GEMM[0] else _NULL:
        rc = lib().m3t_sgemm_scaled(transA, transB, M, N, K, _p(A, a_off), lda, _p(B, b_off), ldb, _p(Cm, c_off), ldc,
                                    _p(bias), act, int(accumulate), seg[0], seg[1], seg[2], seg[3],
                                    _p(ws), (ws.numel() * 4) if ws is not None else 0, flags, amax[0], amax[1], _stream())
    _lib.check(rc, "m3t_sgemm")


# ---- magnitude slots of the fp16x3 products (include/m3t_hip.h: m3t_sgemm_scaled) ----------------------------------------------
_AMAX_ONE = {}


def amax_one(device):
    """address of a slot that says |x| <= 1 (GRU outputs: h is a convex combination of tanh values and the previous h)"""
    key = (device.type, device.index)
    t = _AMAX_ONE.get(key)
    if t is None:
        t = _AMAX_ONE[key] = torch.full((1,), 0x3F800000, dtype=torch.int64, device=device)
    return t.data_ptr()


_SLOT_POOL = {}
_SLOT_CHUNK = 8192


def amax_slots(n, device):
    """n fresh (zero) slots; slot i lives at .data_ptr() + 8 i.  Views of a pre-zeroed chunk (round 5: one fill kernel per 8192 slots instead
    of one per call -- ~15 small launches per C3 step, several of them on the chain).  A chunk is zeroed on the stream that is current when
    it is created and the library's other streams are made to wait for that fill; a slot is handed out once."""
    key = (device.type, device.index)
    st = _SLOT_POOL.get(key)
    if st is None or st[1] + n > st[0].numel():
        st = _SLOT_POOL[key] = [torch.zeros(max(_SLOT_CHUNK, n), dtype=torch.int64, device=device), 0]
        if device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(cur_stream(device))
            others = ([_SIDE.get(key)] if _SIDE.get(key) is not None else []) + list(_WGRAD.get(key) or [])
            for other in others:
                other.wait_event(ev)
            st.append(ev)
            st.append({cur_stream(device).cuda_stream} | {o.cuda_stream for o in others})      # raw handles already ordered behind the fill
    if len(st) > 3 and device.type == "cuda":
        # a consumer on another stream than the one that filled the chunk (a caller's own stream, FlatGradDDP's early-bucket stream): wait once
        h = _RAW_STREAM(device.index if device.index is not None else _CUR_DEV()) if _RAW_STREAM is not None else None
        if h is not None and h not in st[3]:
            cur_stream(device).wait_event(st[2])
            st[3].add(h)
    v = st[0][st[1]:st[1] + n]
    st[1] += n
    return v


# ---- weight magnitudes measured once per step, off the hot calls (round 5) ---------------------------------------------------------
# The fp16x3 contractions scale each operand by its magnitude; until round 4 every grouped BiGRU / Linear forward measured its weights
# itself (one launch each, ~0.2 ms of the C3 step's chain).  FlatGradDDP.zero_grad() -- the point of a step at which the weights are
# final (the optimizer has run) -- now measures EVERY weight matrix of the model in one go; forward calls find the slot here.  Entries
# die at FlatGradDDP.finish() (the optimizer is about to change the weights): a forward outside a step (validation) measures for itself
# as before.  M3T_WEIGHT_AMAX=0 turns the table off.
_W_AMAX = {}              # id(parameter) -> (weakref(parameter), slot address, owner id, slots tensor, parameter._version, data_ptr)
_W_AMAX_ON = os.environ.get("M3T_WEIGHT_AMAX", "1") != "0"


def measure_weight_amax(params, owner=None):
    """one measuring pass over every float32 matrix among `params` (16 per launch); called by FlatGradDDP.zero_grad()"""
    if not _W_AMAX_ON:
        return
    # (N-D weights -- the stems' convolutions -- as [C_out, the rest])
    ws = [p for p in params if p.dim() >= 2 and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
          and (p.numel() // p.shape[0]) % 4 == 0 and p.data_ptr() % 16 == 0]
    if not ws:
        return
    slots = amax_slots(len(ws), ws[0].device)
    if measure_amax([(p.detach().view(p.shape[0], -1), slots.data_ptr() + 8 * i) for i, p in enumerate(ws)]):
        for i, p in enumerate(ws):
            _W_AMAX[id(p)] = (weakref.ref(p), slots.data_ptr() + 8 * i, id(owner), slots, p._version, p.data_ptr())


def drop_weight_amax(owner=None):
    for k in [k for k, e in _W_AMAX.items() if owner is None or e[2] == id(owner)]:
        del _W_AMAX[k]


def weight_amax(w, keep=None):
    """address of this step's magnitude slot of the parameter object `w`, or None (not measured this step: the caller measures); keep: a
    list that receives the tensor holding the slot (an autograd context keeps it alive for its backward).  An entry is only trusted while the
    parameter is the tensor that was measured: an in-place change since (optimizer.step() after zero_grad(), load_state_dict, an EMA copy, a
    clamp -- anything that bumps ._version) or a re-pointed .data drops it and the caller measures again."""
    e = _W_AMAX.get(id(w))
    if e is None or e[0]() is not w:
        return _frozen_weight_amax(w, keep)
    if e[4] != w._version or e[5] != w.data_ptr():
        del _W_AMAX[id(w)]
        return None
    if keep is not None and not any(k is e[3] for k in keep):
        keep.append(e[3])
    return e[1]


_W_AMAX_FROZEN = {}      # id(parameter) -> (weakref, slot tensor, version, data_ptr): weights measured under no_grad


def _frozen_weight_amax(w, keep):
    """Inference (torch.no_grad(): validation_step / test_step, reference models/model.py:226-246,320-337): nothing changes the weights between
    calls, so a weight matrix is measured ONCE and the slot is kept for as long as the parameter is that tensor at that version (an optimizer
    step, load_state_dict or any in-place write bumps it: measured again).  Training steps never come here with a hit -- FlatGradDDP measures
    per step (measure_weight_amax), plain autograd training measures per call as before (grad mode on)."""
    if torch.is_grad_enabled() or not _W_AMAX_ON or not torch.is_tensor(w) or not w.is_cuda or w.dtype != torch.float32 or w.dim() < 2:
        return None
    e = _W_AMAX_FROZEN.get(id(w))
    if e is not None and e[0]() is w and e[2] == w._version and e[3] == w.data_ptr():
        slot = e[1]
    else:
        if not w.is_contiguous() or (w.numel() // w.shape[0]) % 4 != 0 or w.data_ptr() % 16 != 0:
            return None
        if len(_W_AMAX_FROZEN) > 4096:
            _W_AMAX_FROZEN.clear()
        slot = torch.zeros(1, dtype=torch.int64, device=w.device)      # (its own allocation: the slot pool's chunks are recycled per step)
        if not measure_amax([(w.detach().view(w.shape[0], -1), slot.data_ptr())]):
            return None
        _W_AMAX_FROZEN[id(w)] = (weakref.ref(w), slot, w._version, w.data_ptr())
    if keep is not None and not any(k is slot for k in keep):
        keep.append(slot)
    return slot.data_ptr()


def _is_unit(x):
    """x carries the |x| <= 1 tag of a GRU output AND has not been written in place since (out.mul_(), a clamp ...: ADVICE r5)"""
    v = getattr(x, "_m3t_unit", None)
    return v is not None and v is not False and v == x._version + 1


def _tagged_amax(x):
    """the magnitude slot a producer attached to x (temporal_block), or None when x was modified in place after the measurement"""
    e = getattr(x, "_m3t_amax", None)
    if e is None:
        return None
    return e[0] if e[1] == x._version else None


def measure_amax(items):
    """items: (tensor viewed as [rows, cols] rows-contiguous, slot address) pairs -- one launch per 16 of them; returns False (and
    measures nothing) unless every tensor is 16-B aligned with cols % 4 == 0"""
    for t, _ in items:
        if t.shape[-1] % 4 != 0 or t.data_ptr() % 16 != 0 or not t.is_contiguous():
            return False
    for i in range(0, len(items), 16):
        chunk = items[i:i + 16]
        n = len(chunk)
        xs = (C.c_void_p * n)(*[t.data_ptr() for t, _ in chunk])
        rows = (C.c_size_t * n)(*[t.numel() // t.shape[-1] for t, _ in chunk])
        cols = (C.c_int * n)(*[t.shape[-1] for t, _ in chunk])
        lds = (C.c_size_t * n)(*[t.shape[-1] for t, _ in chunk])
        sl = (C.c_void_p * n)(*[a for _, a in chunk])
        _lib.check(lib().m3t_absmax(n, xs, rows, cols, lds, sl, _stream()), "m3t_absmax")
    return True


def sgemm_plan(transA, M, N, K, seg_len=0, exclusive=False, prec=None, ws_bytes=_WS_MIN):
    """(kernel, splits) m3t_sgemm would use: kernel 0 fp32-MFMA, 1 the 16-bit-term tile kernels (gemm_x6.hip / gemm_x6d.hip)."""
    k, sp = C.c_int(0), C.c_int(0)
    flags = (_PREC[0] if prec is None else prec) | (_lib.M3T_GEMM_EXCLUSIVE if exclusive else 0)
    _lib.check(lib().m3t_sgemm_plan(transA, M, N, K, seg_len, ws_bytes, flags, C.byref(k), C.byref(sp)), "m3t_sgemm_plan")
    return k.value, sp.value


def colsum(X, x_off, M, N, ld, out, accumulate=False):
    ws = workspace(out.device)
    rc = lib().m3t_colsum(_p(X, x_off), M, N, ld, _p(out), int(accumulate), _p(ws), ws.numel() * 4, _stream())
    _lib.check(rc, "m3t_colsum")


def transpose2d(src):
    R, Cc = src.shape
    dst = torch.empty(Cc, R, dtype=src.dtype, device=src.device)
    _lib.check(lib().m3t_transpose(_p(src), R, Cc, Cc, _p(dst), R, _stream()), "m3t_transpose")
    return dst


def amax_out(slot):
    """the NEXT conv1d / mask_pos / weight_norm_fwd call raises the magnitude slot at address `slot` to max |its output| in the same kernel
    (include/m3t_hip.h, m3t_amax_out): the producer measures what the consuming fp16x3 contraction scales by.  Call it IMMEDIATELY in front
    of the consuming call -- after every allocation (an allocation failure between the two would leave the pointer parked for an
    unrelated later call; the wrappers below also clear it on any failure)"""
    if slot is not None:
        _lib.check(lib().m3t_amax_out(C.c_void_p(slot)), "m3t_amax_out")


def _amax_clear():
    _lib.check(lib().m3t_amax_out(None), "m3t_amax_out")


def mask_pos(s, dy, mul=None, drop=None, amax=None):
    """dy where s > 0 (x mul); drop = (p, seed): x the in-kernel dropout mask of the forward conv epilogue, regenerated;
    amax: address of a magnitude slot the kernel raises to max |out|"""
    out = torch.empty_like(dy)
    amax_out(amax)
    try:
        if drop is not None and drop[0] > 0:
            Cc = dy.shape[-1]
            _lib.check(lib().m3t_mask_pos_drop(_p(s), _p(dy), _p(out), dy.numel() // Cc, Cc, float(drop[0]), int(drop[1]), _stream()),
                       "m3t_mask_pos_drop")
            return out
        _lib.check(lib().m3t_mask_pos(_p(s), _p(dy), _p(mul), _p(out), dy.numel(), _stream()), "m3t_mask_pos")
        return out
    except BaseException:
        _amax_clear()
        raise


# ----------------------------------------------------------------------------- Linear
class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) on the fp32-MFMA GEMM; replaces nn.Linear (+ReLU) call sites
    (reference models/rnn.py:22-55, models/model.py:88, models/att_fusion.py:13)."""

    @staticmethod
    def forward(ctx, x, w, b, act, unit=False):
        """unit: |x| <= 1 is known (x is a GRU output: m3t.ops.multi_bigru marks its outputs): its magnitude slot is a constant, nothing
        is measured for it"""
        x = _req(x.contiguous(), "x"); _req(w, "weight")
        K = x.shape[-1]
        N = w.shape[0]
        M = x.numel() // K
        y = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
        ctx.prec = _PREC[0]
        # fp16x3 products: x and w are measured once here (one launch) and their slots serve the backward GEMMs too
        slots = None
        ctx.w_slot = None
        if (ctx.prec & _lib.M3T_GEMM_F16X3) and M % 128 == 0 and N % 64 == 0 and K % 32 == 0 and w.is_contiguous():
            slots = amax_slots(3, x.device)
            ctx.w_keep = []
            ctx.w_slot = weight_amax(w, ctx.w_keep)  # measured once per step by FlatGradDDP.zero_grad(), else here
            ctx.x_slot = amax_one(x.device) if unit else slots.data_ptr()
            todo = ([] if unit else [(x, slots.data_ptr())]) + ([] if ctx.w_slot is not None else [(w, slots.data_ptr() + 8)])
            if todo and not measure_amax(todo):
                slots = None
            elif ctx.w_slot is None:
                ctx.w_slot = slots.data_ptr() + 8
        sgemm(0, 1, M, N, K, x, 0, K, w, 0, K, y, 0, N, bias=b, act=act, prec=ctx.prec, exclusive=True,
              amax=(None, None) if slots is None else (ctx.x_slot, ctx.w_slot))
        ctx.save_for_backward(x, w, y if act else None, slots)
        ctx.act, ctx.has_bias = act, b is not None
        ctx.bias_ref = b if (b is not None and id(b) in _GRAD_SINKS) else None
        ctx.weight_ref = w if id(w) in _GRAD_SINKS else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, slots = ctx.saved_tensors
        dy = _req(dy.contiguous(), "dy")
        if ctx.act:
            dy = mask_pos(y, dy)
        K, N = x.shape[-1], w.shape[0]
        M = x.numel() // K
        dx = dw = db = None
        a_x = a_w = a_dy = None
        if slots is not None and measure_amax([(dy, slots.data_ptr() + 16)]):       # dy: once for both backward GEMMs
            a_x, a_w, a_dy = ctx.x_slot, ctx.w_slot, slots.data_ptr() + 16
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            sgemm(0, 0, M, K, N, dy, 0, N, w, 0, K, dx, 0, K, prec=ctx.prec, exclusive=True, amax=(a_dy, a_w))
        w_sink = _take_sink(ctx.weight_ref) if (ctx.needs_input_grad[1] and ctx.weight_ref is not None) else None
        b_sink = _take_sink(ctx.bias_ref) if (ctx.has_bias and ctx.needs_input_grad[2] and ctx.bias_ref is not None) else None
        if (ctx.needs_input_grad[1] and ctx.weight_ref is not None and w_sink is None) or \
                (ctx.has_bias and ctx.needs_input_grad[2] and ctx.bias_ref is not None and b_sink is None):
            # a sink exists but was already taken this step (the parameter is used twice in the graph, or gradients are being
            # accumulated over several backward passes): autograd will ADD the tensor returned below onto the slice on this
            # stream, while the first use may still be writing that slice on the weight-gradient stream -- order them
            join_wgrad(x.device)
        # gradients that go straight into the flat buffer feed nothing on the chain: they run on the weight-gradient stream
        # (joined by FlatGradDDP.finish(), see join_wgrad) beside whatever backward does next
        wg = wgrad_stream(x.device, _LINEAR_RR[0]) if x.is_cuda else None
        off_chain = on_stream(wg) if wg is not None else None
        if off_chain is not None and (w_sink is not None or b_sink is not None):
            wg.wait_stream(cur_stream())
        if ctx.needs_input_grad[1]:
            if w_sink is not None and off_chain is not None:
                with off_chain:
                    sgemm(1, 0, N, K, M, dy, 0, N, x, 0, K, w_sink, 0, K, prec=ctx.prec, amax=(a_dy, a_x))
                    if slots is not None:
                        slots.record_stream(wg)
            else:
                dw = w_sink if w_sink is not None else torch.empty_like(w)
                sgemm(1, 0, N, K, M, dy, 0, N, x, 0, K, dw, 0, K, prec=ctx.prec, exclusive=True, amax=(a_dy, a_x))
                if w_sink is not None:
                    dw = None                              # written in place: nothing for autograd to accumulate
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if b_sink is not None and off_chain is not None:
                with off_chain:
                    colsum(dy, 0, M, N, N, b_sink)
            else:
                db = b_sink if b_sink is not None else torch.empty(N, dtype=x.dtype, device=x.device)
                colsum(dy, 0, M, N, N, db)
                if b_sink is not None:
                    db = None
        if off_chain is not None and (w_sink is not None or b_sink is not None):
            dy.record_stream(wg)
            x.record_stream(wg)
            _WGRAD_PENDING[(x.device.type, x.device.index)] = True
            _LINEAR_RR[0] += 1
        return dx, dw, db, None, None


_LINEAR_RR = [0]      # round-robin over the weight-gradient streams


# ---- BatchNorm's num_batches_tracked (reference: nn.BatchNorm*d in train mode adds 1 per forward call) -------------------------------------
# One int64 add per BatchNorm and forward pass is one launch each: 28 on the chain of a ResNet3D+CBAM step, 5 us apiece with their gap.  Inside
# `batch_counters()` (the visual models' forward) they are collected and added by ONE multi-tensor launch when the block ends.
_NBT_PENDING = [None, 0]


def count_batch(counter):
    if counter is None:
        return
    if _NBT_PENDING[0] is not None:
        _NBT_PENDING[0].append(counter)
    else:
        counter.add_(1)


class batch_counters:
    def __enter__(self):
        if _NBT_PENDING[1] == 0:
            _NBT_PENDING[0] = []
        _NBT_PENDING[1] += 1
        return self

    def __exit__(self, *exc):
        _NBT_PENDING[1] -= 1
        if _NBT_PENDING[1] == 0:
            pending, _NBT_PENDING[0] = _NBT_PENDING[0], None
            if pending:
                if len(pending) > 1 and all(t.is_cuda for t in pending):
                    torch._foreach_add_(pending, 1)
                else:
                    for t in pending:
                        t.add_(1)
        return False


class _AddRelu(torch.autograd.Function):
    """relu(a + b) in one pass (csrc/gemm.hip add_relu_kernel): the end of a ResNet block, reference models/resnet.py:52-54"""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        slot = amax_slots(1, a.device) if TRANSPOSE_IMAGES[0] else None
        if slot is not None:
            amax_out(slot.data_ptr())
        _lib.check(lib().m3t_add_relu(_p(a), _p(b), _p(out), a.numel(), _stream()), "m3t_add_relu")
        _OUT_SLOT[0] = slot
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, = ctx.saved_tensors
        g = mask_pos(out, _req(dout.contiguous(), "dout"))
        return g, g


def add_relu(a, b):
    """relu(a + b); device fp32 tensors of one shape, both contiguous -- anything else: the stock operators"""
    if (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape and a.device == b.device
            and a.is_contiguous() and b.is_contiguous()):
        _OUT_SLOT[0] = None
        return _tag_out(_AddRelu.apply(a, b))
    return torch.relu(a + b)


class _ReluDropout(torch.autograd.Function):
    """nn.Dropout(p) behind a ReLU (the FC heads of reference models/rnn.py:24-28,40-49 with dropout=True): the mask is the
    Philox4x32-10 word of element (row, col) under `seed`, generated in the kernel forward AND backward (csrc/common.h, the same
    generator as the TCN's conv epilogues) -- no mask tensor, no torch RNG kernel.  `y` must be a ReLU output (>= 0)."""

    @staticmethod
    def forward(ctx, y, p, seed):
        y = _req(y.contiguous(), "y")
        ctx.save_for_backward(y)
        ctx.drop = (float(p), int(seed))
        return mask_pos(y, y, drop=ctx.drop)              # y * mask / (1 - p)   (y > 0; zeros stay zero)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return mask_pos(y, _req(dy.contiguous(), "dy"), drop=ctx.drop), None, None


def relu_dropout(y, p, seed=None):
    """dropout of a ReLU output with an in-kernel mask; seed: 64-bit, default drawn from torch's CPU generator (torch.manual_seed
    makes a run reproducible, as for the TCN blocks)"""
    if p <= 0:
        return y
    if seed is None:
        seed = int(torch.empty(1, dtype=torch.int64).random_().item())
    return _ReluDropout.apply(y, float(p), int(seed))


def linear(x, w, b=None, act=0):
    return _Linear.apply(x, w, b, act, _is_unit(x))


# ----------------------------------------------------------------------------- BiGRU
SCAN_PER_STEP = [False]     # tests/benchmarks: force the launch-per-step scan path
SCAN_FAULT = [False]        # tests: fault injection (M3T_SCAN_FAULT), see include/m3t_hip.h
# Range probe of the fp16x3 mode: RANGE_PROBE[0] = [] makes every grouped BiGRU backward record, per (layer, stack,
# direction), how far the gradient operands it hands to the fp16x3 GEMMs -- dgx / dgh [B T, 3H] -- spread below their maximum: rows
# (a frame: an output row of the data-gradient GEMM dX = dgx W_ih) and columns (a gate unit: an output row of the weight-gradient GEMMs
# dW = dgx^T x) whose LARGEST element is more than 2^17 below the operand's maximum are built only of elements past the mode's
# full-precision range (error model: absolute error 2^-39 of the maximum there).  Recorded: the widest spread, how many rows /
# columns are past the range, and their share of the operand's energy (sum of squares) -- what the limit can cost a gradient normwise.
# Debug aid (a dozen extra reductions per level); env M3T_F16X3_PROBE=1 arms it process-wide; range_probe_report() evaluates (synchronises).
RANGE_PROBE = [[] if os.environ.get("M3T_F16X3_PROBE") == "1" else None]


def _probe_range(tag, t2d):
    """t2d: [rows, cols] view of a gradient operand"""
    a = t2d.abs()
    rmax, cmax, gmax = a.amax(1), a.amax(0), a.max()
    big = torch.full((), float("inf"), device=t2d.device)
    lim = gmax * 2.0 ** -17
    sq = t2d.double() ** 2
    tot = sq.sum()
    past_r, past_c = rmax < lim, cmax < lim
    RANGE_PROBE[0].append((tag, gmax, torch.where(rmax > 0, rmax, big).min(), torch.where(cmax > 0, cmax, big).min(),
                           past_r.sum(), past_c.sum(), sq.sum(1)[past_r].sum() / tot, sq.sum(0)[past_c].sum() / tot))


def range_probe_report(clear=True):
    """per probed operand: dict(tag, max, row_spread, col_spread = max / smallest non-zero row / column maximum, rows_past, cols_past =
    rows / columns entirely more than 2^17 below the maximum, row_energy_past, col_energy_past = their share of the sum of squares)"""
    keys = ("tag", "max", "row_spread", "col_spread", "rows_past", "cols_past", "row_energy_past", "col_energy_past")
    out = []
    for rec in (RANGE_PROBE[0] or []):
        v = [rec[0]] + [float(x) for x in rec[1:]]
        v[2], v[3] = v[1] / v[2], v[1] / v[3]
        out.append(dict(zip(keys, v)))
    if clear and RANGE_PROBE[0] is not None:
        RANGE_PROBE[0] = []
    return out


PREP_AHEAD = [os.environ.get("M3T_SCAN_PREP_AHEAD", "1") != "0"]      # the backward scans' weight fragments during forward (A/B switch, tests)
SCAN_FIRST = [os.environ.get("M3T_SCAN_FIRST", "0") == "1"]            # opt-in (measured neutral: 12.005 vs 12.017 ms): weight gradients of level l start once the scan of level l - 1 is resident
FORCE_WIDE_FWD = [False]    # tests: every forward scan asks for the wide form (by default only the level that makes room for the audio scans does)
SCAN_FP32 = [False]         # tests: keep the persistent forward scan on fp32 MFMAs (bit-identical to the per-step kernels)


_FENCED = [False]     # set by the interleaved schedule of _MultiBiGRU: its side-stream scans are fenced by events


def _scan_flags(device):
    """A persistent scan launch needs every workgroup resident, so two of them may only run concurrently on one device when
    BOTH grids fit the chip together: scans issued on the side stream take the launch-per-step path (include/m3t_hip.h,
    m3t_gru_scan_fwd) -- unless the caller has checked that (the concurrent schedule of _MultiBiGRU, _pair_fits) or fences
    them against every other scan with events (its alternating schedule)."""
    if SCAN_PER_STEP[0] or (_ws_tag(device) == "side" and not _FENCED[0]):
        return _lib.M3T_SCAN_NO_PERSIST
    return (_lib.M3T_SCAN_FP32 if SCAN_FP32[0] else 0) | (_lib.M3T_SCAN_FAULT if SCAN_FAULT[0] else 0)


_DEFERRED = [False]      # several ranks: failures are raised only at points all ranks agree on (FlatGradDDP)


def defer_scan_errors(on=True):
    """Several ranks (FlatGradDDP with world > 1 calls this): a rank must not raise the moment ITS host sees the scan error flag --
    mid-step, from a scan call or a poll -- because its peers would then wait forever in the step's collective.  While deferred,
    scan calls launch behind a dead scan (the device-side guards skip every such step), poll_scan_error() is silent, and the
    failure is raised on ALL ranks at the same step from the all-reduced dead slot (FlatGradDDP.finish) or by
    FlatGradDDP.agree_on_scan_error() at points every rank reaches (validation, checkpoints, the end of fit)."""
    _DEFERRED[0] = bool(on)
    _lib.check(lib().m3t_gru_error_defer(1 if on else 0), "m3t_gru_error_defer")


_DEFER_OWNERS = [0]


def defer_scan_errors_acquire():
    """refcounted form for owners that come and go (FlatGradDDP instances): the process-global switch stays on while ANY owner is alive --
    an older instance that is closed or garbage-collected after a newer one was built must not turn the newer one's deferral off"""
    _DEFER_OWNERS[0] += 1
    if _DEFER_OWNERS[0] == 1:
        defer_scan_errors(True)


def defer_scan_errors_release():
    if _DEFER_OWNERS[0] > 0:
        _DEFER_OWNERS[0] -= 1
        if _DEFER_OWNERS[0] == 0:
            defer_scan_errors(False)


def poll_scan_error(sync=False, force=False):
    """raise M3THipError if a persistent scan has died (sync=True: wait for the device first, so the answer covers everything
    issued so far).  The error state is sticky on the device (every optimizer step queued behind the dead scan skips itself,
    include/m3t_hip.h); raising synchronises and clears it, so the caller can redo the step.  Silent while failures are
    deferred to rank-agreed points (defer_scan_errors) unless force=True."""
    if _DEFERRED[0] and not force:
        return
    if sync and torch.cuda.is_available():
        torch.cuda.synchronize()
    try:
        _lib.poll_scan_error()
    except M3THipError:
        reset_progress()             # a dead scan may have left its progress counters short of what the library's shadow expects
        raise


def inject_scan_error():
    """fault injection (tests): raise the scan error flag from a kernel on the current stream, as a dying scan would"""
    _lib.check(lib().m3t_gru_inject_error(_stream()), "m3t_gru_inject_error")


def persist_owner():
    """0: no persistent scan attempted yet on this device, 1: this process owns them, 2: another process does (this one runs
    the launch-per-step kernels, several times slower)"""
    return int(lib().m3t_gru_persist_owner())


def grad_poison_(flat, dead):
    """before the gradient all-reduce: a dead scan of THIS rank poisons flat[0] (NaN) and sets dead[0] (include/m3t_hip.h)"""
    _lib.check(lib().m3t_grad_poison(_p(flat), _p(dead), _stream()), "m3t_grad_poison")


def grad_dead_check_(dead):
    """after the all-reduce: a non-zero dead[0] (some rank died) raises this rank's scan error flag as well"""
    _lib.check(lib().m3t_grad_dead_check(_p(dead), _stream()), "m3t_grad_dead_check")


def _scan_after(ev):
    """the next scan call launches its scan kernels only after torch event `ev` (its preparation kernels are not held back)"""
    if ev is not None:
        _lib.check(lib().m3t_gru_scan_after(C.c_void_p(ev.cuda_event)), "m3t_gru_scan_after")


_ARENAS = {}
_ARENA_BYTES = (8 << 20) + (64 << 10)      # 4 granule formats (8 MiB) + the placement-handshake table (64 KiB)


def _scan_arena(device):
    """the exchange arena of the persistent scans issued on the current stream role (include/m3t_hip.h, m3t_gru_scan_arena):
    8 MiB + 64 KiB that nothing but scan launches ever writes -- one per (device, stream role), because launches that share an arena
    must be ordered -- handed to the next scan call"""
    key = (device.type, device.index, _ws_tag(device))
    a = _ARENAS.get(key)
    if a is None:
        a = torch.empty(_ARENA_BYTES, dtype=torch.uint8, device=device)
        _ARENAS[key] = a
        _lib.check(lib().m3t_gru_scan_arena_reset(C.c_void_p(a.data_ptr())), "m3t_gru_scan_arena_reset")
    _lib.check(lib().m3t_gru_scan_arena(C.c_void_p(a.data_ptr()), _ARENA_BYTES), "m3t_gru_scan_arena")


# ---- direction-split, time-chunked hand-offs around the persistent scans (round 5) ------------------------------------------------
# Layer l+1's input projection is out_fwd W_ih[:, :H]^T + out_rev W_ih[:, H:]^T (reference models/rnn.py:17,75) and each half is final
# for the frames its direction's scan has passed; likewise dX = dgx_fwd W_f + dgx_rev W_r in backward.  A scan launch publishes PROGRESS
# MARKS (include/m3t_hip.h, m3t_gru_scan_progress); a consumer stream waits for a mark with a one-lane gate kernel and runs the product of
# that direction's half over that TIME WINDOW (m3t_sgemm_window) while the scan is still running.  Per window the first arriver writes
# (+ bias), the second accumulates -- the order is fixed by the windows' arrival steps, so results do not depend on timing.
# M3T_SCAN_CHUNKS: "0" (default: OFF -- measured slower, NOTEBOOK.md R5.1: 12.17 ms unchunked, 12.48 forward only, 12.30 backward only,
# 12.65 both), "1" both passes, "fwd" / "bwd" one pass only; tests flip CHUNKS[0] (True / False): the unchunked schedule is the yardstick
# of the chunked one
_ce = os.environ.get("M3T_SCAN_CHUNKS", "0")
CHUNKS = [{"0": False, "1": True}.get(_ce, _ce)]
_CHUNK_WINDOWS = 4
_PROGRESS = {}


def _progress_counters(device):
    """the two progress words (up-scans, down-scans) of the scan launches issued on the current stream role: zeroed once, then written by
    scan launches only (launches that share a pair must be ordered: one pair per (device, stream role), like the exchange arenas)"""
    key = (device.type, device.index, _ws_tag(device))
    t = _PROGRESS.get(key)
    if t is None:
        t = _PROGRESS[key] = torch.zeros(2, dtype=torch.int32, device=device)
        _lib.check(lib().m3t_gru_scan_progress_reset(C.c_void_p(t.data_ptr())), "m3t_gru_scan_progress_reset")
    return t


def reset_progress():
    """after a scan error: the device's progress words and the library's shadow totals start from zero again (the raise that leads here has
    synchronised the device: nothing is in flight)"""
    for t in _PROGRESS.values():
        t.zero_()
        _lib.check(lib().m3t_gru_scan_progress_reset(C.c_void_p(t.data_ptr())), "m3t_gru_scan_progress_reset")


def _chunk_bounds(B, T, n=None, backward=False):
    """time bounds tb[0 .. n-2] that cut [0, T) into n windows whose row counts B * len are whole 128-row GEMM tiles, or None"""
    import math
    n = n or _CHUNK_WINDOWS
    q = 128 // math.gcd(B, 128)
    on = CHUNKS[0] is True or CHUNKS[0] == ("bwd" if backward else "fwd")
    if not on or (B * T) % 128 != 0 or T < 16 * n or B * T < 4096:
        return None
    tb = [int(round(k * T / n / q)) * q for k in range(1, n)]
    lo = [0] + tb
    hi = tb + [T]
    if any(b - a < 8 for a, b in zip(lo, hi)) or tb[0] < 4 or tb[-1] > T - 4 or (T - tb[-1]) * B % 128 != 0:
        return None
    return tb


def _arrivals(tb, T):
    """[(dir, k, window, second)] in the order the halves of the windows become final: dir 0 = the scans that walk time upwards, 1 = downwards;
    k = index into that direction's `need` values (None: final only when the scan launch has ended); second: the window's other half came first"""
    n = len(tb)
    ev = []
    for w in range(n + 1):
        ev.append((tb[w] if w < n else T, 0, w if w < n else None, w))
        ev.append((T - tb[w - 1] if w > 0 else T, 1, (n - w) if w > 0 else None, w))
    ev.sort(key=lambda e: (e[0], e[1]))
    seen, out = set(), []
    for step, d, k, w in ev:
        out.append((d, k, w, w in seen))
        seen.add(w)
    return out


def _wait_progress(counters, dirn, need):
    _lib.check(lib().m3t_stream_wait_progress(C.c_void_p(counters.data_ptr() + 4 * dirn), int(need) & 0xffffffff, _stream()), "m3t_stream_wait_progress")


def sgemm_window(transB, n_seg, win_len, win_stride, win_off, N, K, A, a_off, lda, Bm, b_off, ldb, Cm, c_off, ldc, bias=None, act=0,
                 accumulate=False, prec=None, amax=(None, None)):
    """C[window rows, :N] = A[window rows, :K] op(B) (+ bias) (+ C): the rows are the frames [win_off, win_off + win_len) of every clip of
    batch-major [n_seg, win_stride, *] tensors (include/m3t_hip.h, m3t_sgemm_window)"""
    flags = _PREC[0] if prec is None else prec
    rc = lib().m3t_sgemm_window(transB, n_seg, win_len, win_stride, win_off, N, K, _p(A, a_off), lda, _p(Bm, b_off), ldb, _p(Cm, c_off), ldc,
                                _p(bias), act, int(accumulate), flags, amax[0], amax[1], _stream())
    _lib.check(rc, "m3t_sgemm_window")


def sgemm_window_batch(problems, transB, n_seg, win_len, win_stride, win_off, N, K, lda, ldb, ldc, act=0, prec=None):
    """problems: [(A, a_off, B, b_off, C, c_off, bias or None, accumulate, amax_a, amax_b)] of ONE shape: one launch (<= 8 per launch)"""
    flags = _PREC[0] if prec is None else prec
    WP = _lib.WindowProblem
    for i in range(0, len(problems), _lib.M3T_WINDOW_BATCH):
        chunk = problems[i:i + _lib.M3T_WINDOW_BATCH]
        arr = (WP * len(chunk))(*[WP(_vp(A, ao), _vp(Bm, bo), _vp(Cm, co), _vp(bias), aa, ab, int(acc))
                                  for A, ao, Bm, bo, Cm, co, bias, acc, aa, ab in chunk])
        _lib.check(lib().m3t_sgemm_window_batch(len(chunk), arr, transB, n_seg, win_len, win_stride, win_off, N, K, lda, ldb, ldc, act, flags,
                                                _stream()), "m3t_sgemm_window_batch")


def _scan_fwd(descs, B, T, prec=0, after=None, progress=None):
    """progress = (counters tensor, time bounds): arm the launch's progress marks; returns the `need` table (2 x len(bounds) counter values)"""
    need = None
    if progress is not None:
        assert len(descs) <= M3T_MAX_SCANS
        ctr, tb = progress
        need = (C.c_uint * (2 * len(tb)))()
        _lib.check(lib().m3t_gru_scan_progress(C.c_void_p(ctr.data_ptr()), len(tb), (C.c_int * len(tb))(*tb), need), "m3t_gru_scan_progress")
    _scan_fwd_(descs, B, T, prec, after)
    return None if need is None else list(need)


def _scan_fwd_(descs, B, T, prec=0, after=None):
    for i in range(0, len(descs), M3T_MAX_SCANS):
        chunk = descs[i:i + M3T_MAX_SCANS]
        arr = (GruFwdDesc * len(chunk))(*chunk)
        flops = (T - 1) * sum(2.0 * B * 3 * d.H * d.H for d in chunk)
        with _Timed("gru_step_fwd_kernel", T, flops, kernel_side=True) as tm:
            dev = torch.device("cuda", _CUR_DEV())
            ws = workspace(dev)
            n0 = lib().m3t_gru_persist_count() if tm.rec is not None else 0
            _scan_after(after if i == 0 else None)
            _scan_arena(dev)
            rc = lib().m3t_gru_scan_fwd(arr, len(chunk), B, T, _p(ws), ws.numel() * 4, _scan_flags(dev) | prec, _stream())
            if tm.rec is not None and lib().m3t_gru_persist_count() != n0:      # one launch ran all T steps
                tm.rec.update(kernel="gru_persist_fwd_kernel", launches=1, steps=T)
        _lib.check(rc, "m3t_gru_scan_fwd")


def _scan_bwd(descs, B, T, prec=0, after=None, progress=None):
    need = None
    if progress is not None:
        assert len(descs) <= M3T_MAX_SCANS
        ctr, tb = progress
        need = (C.c_uint * (2 * len(tb)))()
        _lib.check(lib().m3t_gru_scan_progress(C.c_void_p(ctr.data_ptr()), len(tb), (C.c_int * len(tb))(*tb), need), "m3t_gru_scan_progress")
    _scan_bwd_(descs, B, T, prec, after)
    return None if need is None else list(need)


def _scan_bwd_(descs, B, T, prec=0, after=None):
    for i in range(0, len(descs), M3T_MAX_SCANS):
        chunk = descs[i:i + M3T_MAX_SCANS]
        arr = (GruBwdDesc * len(chunk))(*chunk)
        flops = (T - 1) * sum(2.0 * B * 3 * d.H * d.H for d in chunk)
        with _Timed("gru_step_bwd_kernel", T, flops, kernel_side=True) as tm:
            dev = torch.device("cuda", _CUR_DEV())
            ws = workspace(dev)
            n0 = lib().m3t_gru_persist_count() if tm.rec is not None else 0
            _scan_after(after if i == 0 else None)
            _scan_arena(dev)
            rc = lib().m3t_gru_scan_bwd(arr, len(chunk), B, T, _p(ws), ws.numel() * 4, _scan_flags(dev) | prec, _stream())
            if tm.rec is not None and lib().m3t_gru_persist_count() != n0:
                tm.rec.update(kernel="gru_persist_bwd_kernel", launches=1, steps=T)
        _lib.check(rc, "m3t_gru_scan_bwd")


def _vp(t, off=0):
    return t.data_ptr() + 4 * off if t is not None else None


def _stream_groups(Hs, B):
    """Split independent stacks over (main, side) streams.  When the widest stacks alone give the scan launch about
    one 16-row workgroup per CU (>= 192 of 256), narrower stacks sharing that launch would double up on CUs and
    stretch every step (measured +2.5 us/step for audio H=256 beside 4 x H=512); they run as their own launches
    on a side stream instead -- both chains are latency-bound and overlap."""
    if _PERSIST_ENABLED and not SCAN_PER_STEP[0] and len(set(Hs)) > 1 and all(h % 128 == 0 and h <= 512 for h in Hs):
        # persistent scans (one launch per level, csrc/gru_persist.hip) need one H per launch and must not overlap each
        # other: one group per H, back to back on the main stream (a persistent launch holds every CU, so a
        # launch-per-step chain on the side stream would only run before or after it anyway)
        return [("main", [i for i, h in enumerate(Hs) if h == hh]) for hh in sorted(set(Hs), reverse=True)]
    hmax = max(Hs)
    heavy = [i for i, h in enumerate(Hs) if h == hmax]
    light = [i for i, h in enumerate(Hs) if h != hmax]
    wgs = sum(2 * (Hs[i] // 16) * ((B + 15) // 16) for i in heavy)
    if light and wgs >= 192:
        return [("side", light), ("main", heavy)]
    return [("main", list(range(len(Hs))))]


def _interleaved(groups):
    return (_PERSIST_ENABLED and not SCAN_PER_STEP[0] and len(groups) == 2
            and all(kind == "main" for kind, _ in groups))


_CONCURRENT_ENABLED = os.environ.get("M3T_SCAN_CONCURRENT", "1") != "0"
_CUS = {}


def _pair_fits(dev, n_heavy, H_heavy, n_light, H_light, B, T, prec):
    """Round 4: may the heavy level's persistent scan (asked to make room: M3T_SCAN_WIDE) and the light level's run AT THE SAME TIME?
    Yes iff both grids fit the chip together, forward and backward, in total and per XCD (workgroups are dealt round-robin over the 8
    XCDs and every scan workgroup owns its CU): then both become resident whatever is still draining from the CUs and neither waits
    for the other (include/m3t_hip.h, m3t_gru_scan_workgroups).  C3: 4 x H=512 wide = 128 workgroups + 2 x H=256 = 64 of 256 CUs."""
    if not _CONCURRENT_ENABLED:
        return False
    key = (dev.type, dev.index)
    cus = _CUS.get(key)
    if cus is None:
        cus = _CUS[key] = torch.cuda.get_device_properties(dev).multi_processor_count
    base = _scan_flags(dev) | prec
    if base & _lib.M3T_SCAN_NO_PERSIST:
        return False
    for bwd in (0, 1):
        a = lib().m3t_gru_scan_workgroups(n_heavy, H_heavy, B, T, base | _lib.M3T_SCAN_WIDE, bwd)
        b = lib().m3t_gru_scan_workgroups(n_light, H_light, B, T, base | _lib.M3T_SCAN_WIDE, bwd)
        if a <= 0 or b <= 0 or a + b > cus or (a + 7) // 8 + (b + 7) // 8 > cus // 8:
            return False
    return True


class _MultiBiGRU(torch.autograd.Function):
    """Several independent stacked bidirectional GRUs (same B, T, depth) advanced together:
    per layer, one input-projection GEMM per direction, then ONE grouped scan over every
    (stack, direction) of a stream group.  Replaces nn.GRU(batch_first=True, bidirectional=True) at reference
    models/rnn.py:17,75 (and its autograd).  tensors = per stack: x, then per layer, per
    direction: w_ih, w_hh, b_ih, b_hh.  Returns per stack: out [B,T,2H], h_n [2L,B,H]."""

    @staticmethod
    def forward(ctx, n_stacks, L, cat_lo, cat_hi, unit_mask, *tensors):
        """cat_lo < cat_hi: the last-layer outputs of stacks cat_lo .. cat_hi-1 are written side by side into ONE buffer
        [B, T, sum 2H] (the `torch.cat(..., -1)` a caller would apply next, without the copy -- and without the slice copies of its
        backward): that buffer is returned in the slot of stack cat_lo, the other stacks of the group return an empty tensor."""
        per = 1 + 8 * L
        assert len(tensors) == n_stacks * per
        xs, params = [], []
        for s in range(n_stacks):
            xs.append(_req(tensors[s * per].contiguous(), "x"))
            params.append([_req(t, "gru parameter") for t in tensors[s * per + 1:(s + 1) * per]])
        B, T = xs[0].shape[0], xs[0].shape[1]
        dev = xs[0].device
        for x in xs:
            if x.shape[0] != B or x.shape[1] != T:
                raise M3THipError("grouped GRU stacks must share batch and length")
        Hs = [params[s][1].shape[1] for s in range(n_stacks)]
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        # every buffer is allocated on the caller's stream; side-stream work is fenced by wait_stream on both ends
        h_ns = [new(2 * L, B, Hs[s]) for s in range(n_stacks)]
        outs = [[new(B, T, 2 * Hs[s]) for s in range(n_stacks)] for _ in range(L)]
        cat_buf = None
        if cat_hi - cat_lo >= 2:
            cat_buf = new(B, T, sum(2 * Hs[s] for s in range(cat_lo, cat_hi)))
            off = 0
            for s in range(cat_lo, cat_hi):
                outs[L - 1][s] = cat_buf[..., off:off + 2 * Hs[s]]          # strided view: row stride = the buffer's width
                off += 2 * Hs[s]
        gates = [[new(2, B, T, 4 * Hs[s]) for s in range(n_stacks)] for _ in range(L)]
        xprojs = [[new(B, T, 6 * Hs[s]) for s in range(n_stacks)] for _ in range(L)]
        main = cur_stream()
        prec = _PREC[0]
        groups = _stream_groups(Hs, B)
        # fp16x3 products: the magnitudes of every layer-0 input and every W_ih, measured once up front (one launch; the backward
        # pass reuses them); deeper layers read GRU outputs, |h| <= 1
        fslots = None
        if (prec & _lib.M3T_GEMM_F16X3) and all(h % 4 == 0 for h in Hs):      # (narrower scans never reach the fp16x3 kernels)
            fslots = amax_slots(n_stacks * (1 + 2 * L), dev)
            # (bit s of unit_mask: stack s reads a GRU output, |x| <= 1: the constant slot, nothing to measure)
            items = [(xs[s], fslots.data_ptr() + 8 * s) for s in range(n_stacks) if not (unit_mask >> s) & 1]
            wslot = {}                                  # (l, s, d) -> slot address: this step's table (weight_amax) or measured here
            ctx.w_keep = []
            for l in range(L):
                for s in range(n_stacks):
                    for d in (0, 1):
                        a = weight_amax(params[s][(2 * l + d) * 4], ctx.w_keep)
                        if a is None:
                            a = fslots.data_ptr() + 8 * (n_stacks + (l * n_stacks + s) * 2 + d)
                            items.append((params[s][(2 * l + d) * 4], a))
                        wslot[(l, s, d)] = a
            if not measure_amax(items):
                fslots = None
        one = amax_one(dev)

        def fslot_x(l, s):
            return None if fslots is None else (fslots.data_ptr() + 8 * s if (l == 0 and not (unit_mask >> s) & 1) else one)

        def fslot_w(l, s, d):
            return None if fslots is None else wslot[(l, s, d)]

        alone = not _interleaved(groups) and all(kind == "main" for kind, _ in groups)   # nothing runs beside these GEMMs

        def level_fwd(l, idxs, scan, after=None, wide=False, progress=None):
            """scan=False: the input projections of layer l for the stacks idxs; scan=True: their grouped scan (its scan
            kernels fenced behind the event `after`; wide: the launch leaves room for a second persistent scan; progress: time bounds
            of the launch's progress marks -- returns their `need` table)"""
            descs = []
            for s in idxs:
                H = Hs[s]
                inp = xs[s] if l == 0 else outs[l - 1][s]
                I = inp.shape[-1]
                for d in (0, 1):
                    w_ih, w_hh, b_ih, b_hh = params[s][(2 * l + d) * 4:(2 * l + d) * 4 + 4]
                    if not scan:
                        sgemm(0, 1, B * T, 3 * H, I, inp, 0, I, w_ih, 0, I, xprojs[l][s], d * 3 * H, 6 * H, bias=b_ih, prec=prec,
                              exclusive=alone, amax=(fslot_x(l, s), fslot_w(l, s, d)))
                    else:
                        descs.append(GruFwdDesc(_vp(xprojs[l][s]), _vp(w_hh), _vp(b_hh), _vp(outs[l][s]),
                                                _vp(gates[l][s], d * B * T * 4 * H), _vp(h_ns[s], (2 * l + d) * B * H),
                                                H, d, 6 * H, d * 3 * H, outs[l][s].stride(1), d * H))
            if scan:
                return _scan_fwd(descs, B, T, prec | (_lib.M3T_SCAN_WIDE if (wide or FORCE_WIDE_FWD[0]) else 0), after,
                                 None if progress is None else (_progress_counters(dev), progress))

        def chunk_plan(idxs, wide):
            """time bounds for direction-split, time-chunked hand-offs between the layers of the stacks idxs (one H), or None"""
            if L < 2 or fslots is None or len({Hs[i] for i in idxs}) != 1 or 2 * len(idxs) > M3T_MAX_SCANS:
                return None
            H = Hs[idxs[0]]
            if H % 64 != 0:
                return None
            fl = _scan_flags(dev) | prec | (_lib.M3T_SCAN_WIDE if (wide or FORCE_WIDE_FWD[0]) else 0)
            if not lib().m3t_gru_scan_progress_ok(2 * len(idxs), H, B, T, fl, 0):
                return None
            return _chunk_bounds(B, T)

        def proj_pieces(l, idxs, tb, need, scan_stream):
            """layer l's input projections from layer l - 1's outputs, as (direction half x time window) pieces on weight-gradient
            stream 0 (idle in forward), each behind the progress mark of the scan whose output it reads; `scan_stream` waits for the last"""
            wg = wgrad_stream(dev, 0)
            ctr = _progress_counters(dev)
            ev_end = torch.cuda.Event()
            ev_end.record(scan_stream)                      # the scan launch of layer l - 1 has ended
            lo, hi = [0] + tb, tb + [T]
            with on_stream(wg):
                ended = False
                for dirn, k, w, second in _arrivals(tb, T):
                    if k is None:
                        if not ended:
                            wg.wait_event(ev_end)
                            ended = True
                    else:
                        _wait_progress(ctr, dirn, need[dirn * len(tb) + k])
                    H = Hs[idxs[0]]
                    lda = outs[l - 1][idxs[0]].stride(1)
                    probs = []
                    for si in idxs:
                        src = outs[l - 1][si]
                        assert src.stride(1) == lda
                        for d in (0, 1):
                            w_ih, _, b_ih, _ = params[si][(2 * l + d) * 4:(2 * l + d) * 4 + 4]
                            # (forward pass: the scans that walk time upwards are the forward direction = columns [0, H) of the output)
                            probs.append((src, dirn * H, w_ih, dirn * H, xprojs[l][si], d * 3 * H, None if second else b_ih, second,
                                          one, fslot_w(l, si, d)))
                    # ONE launch per mark: every (stack, direction) pair that reads this window
                    sgemm_window_batch(probs, 1, B, hi[w] - lo[w], T, lo[w], 3 * H, H, lda, 2 * H, 6 * H, prec=prec)
                ev = torch.cuda.Event()
                ev.record(wg)
            scan_stream.wait_event(ev)

        concurrent = (_interleaved(groups) and len({Hs[i] for i in groups[0][1]}) == 1 and len({Hs[i] for i in groups[1][1]}) == 1
                      and _pair_fits(dev, 2 * len(groups[0][1]), Hs[groups[0][1][0]], 2 * len(groups[1][1]), Hs[groups[1][1][0]], B, T, prec))
        if concurrent:
            # round 4: the heavy level runs on half the CUs (wide workgroups) and the light stack's scans run AT THE SAME TIME on
            # other CUs: two independent chains (projection -> scan -> projection -> scan), no fence between them; the rest of the
            # chip takes their GEMMs.  Enqueue order = the order in which the GPU can start things.
            heavy, light = groups[0][1], groups[1][1]
            side = side_stream(dev)
            side.wait_stream(main)
            _FENCED[0] = True
            try:
                tb = chunk_plan(heavy, True)           # round 5: the heavy stacks' deeper projections run UNDER the scan that feeds them
                if tb is not None:
                    wgrad_stream(dev, 0).wait_stream(main)      # (the magnitude slots, the parameters: everything issued so far)
                for l in range(L):
                    if l == 0 or tb is None:
                        level_fwd(l, heavy, False)
                    ev_al = None
                    if ALIGN_LIGHT[0] and l > 0:
                        # round 6: the light stack's scan of level l starts WITH the heavy one's, not as soon as its own (short) projections are
                        # done -- it then no longer runs beside the heavy level's four projection GEMMs (which took 0.60 instead of 0.42 ms
                        # with 64 of the CUs spinning in that scan) but beside the heavy scan, which leaves those CUs idle anyway
                        ev_al = torch.cuda.Event()
                        ev_al.record(main)
                    with on_stream(side):
                        level_fwd(l, light, False)
                        level_fwd(l, light, True, ev_al, True)      # (wide too: 32 workgroups, one group per XCD, L2-served exchange; -0.03 ms)
                    need = level_fwd(l, heavy, True, None, True, tb if l + 1 < L else None)
                    if tb is not None and l + 1 < L:
                        proj_pieces(l + 1, heavy, tb, need, main)
            finally:
                _FENCED[0] = False
            main.wait_stream(side)
        elif _interleaved(groups):
            # heavy group on the main stream, light group on the side stream; persistent scans strictly alternate
            # (events), so the light scan of layer l runs beside the heavy input projections of layer l+1
            heavy, light = groups[0][1], groups[1][1]
            side = side_stream(dev)
            side.wait_stream(main)
            ev_light = None
            _FENCED[0] = True
            try:
                if L == 2:
                    # persistent scans in the order H0, L0, L1, H1 (alternating H0, L0, H1, L1 in forward only: 15.77 vs 15.68 ms per
                    # step, round 3): BOTH light scans (and the light layer-1 projection between
                    # them) run beside the heavy layer-1 input projections, the longest GEMM window of the pass; strictly
                    # alternating (H0, L0, H1, L1) left the second light scan with nothing beside it
                    level_fwd(0, heavy, False)
                    with on_stream(side):
                        level_fwd(0, light, False)
                    level_fwd(0, heavy, True, None)
                    ev_h0 = torch.cuda.Event()
                    ev_h0.record(main)
                    with on_stream(side):
                        level_fwd(0, light, True, ev_h0)
                        level_fwd(1, light, False)
                        level_fwd(1, light, True)
                        ev_light = torch.cuda.Event()
                        ev_light.record(side)
                    level_fwd(1, heavy, False)
                    level_fwd(1, heavy, True, ev_light)
                else:
                  for l in range(L):
                    level_fwd(l, heavy, False)
                    with on_stream(side):
                        level_fwd(l, light, False)
                    level_fwd(l, heavy, True, ev_light)
                    ev_heavy = torch.cuda.Event()
                    ev_heavy.record(main)
                    with on_stream(side):
                        level_fwd(l, light, True, ev_heavy)
                        ev_light = torch.cuda.Event()
                        ev_light.record(side)
            finally:
                _FENCED[0] = False
            main.wait_stream(side)
        else:
            for kind, idxs in groups:
                stream = main
                if kind == "side":
                    stream = side_stream(dev)
                    stream.wait_stream(main)
                with on_stream(stream):
                    tb = chunk_plan(idxs, False) if (kind == "main" and alone) else None
                    if tb is not None:
                        wgrad_stream(dev, 0).wait_stream(stream)
                    for l in range(L):
                        if l == 0 or tb is None:
                            level_fwd(l, idxs, False)
                        need = level_fwd(l, idxs, True, progress=tb if l + 1 < L else None)
                        if tb is not None and l + 1 < L:
                            proj_pieces(l + 1, idxs, tb, need, stream)
            for kind, _ in groups:
                if kind == "side":
                    main.wait_stream(side_stream(dev))
        del xprojs
        # round 5: W_hh does not change before this step's backward pass -- the wide backward scans' weight fragments (one preparation
        # launch in front of EVERY backward scan, on the chain: ~0.12 ms of the C3 step) are written now, on weight-gradient stream 1,
        # which is idle in forward (include/m3t_hip.h, m3t_gru_bwd_prepare); a level whose backward launch is not that kernel prepares
        # for itself as before
        ctx.wfq, ctx.wfq_ev = None, None
        if PREP_AHEAD[0] and any(ctx.needs_input_grad) and fslots is not None and all(p_.is_contiguous() for prm in params for p_ in prm):
            wfq = {}
            bfl = _scan_flags(dev) | prec | _lib.M3T_SCAN_WHH | _lib.M3T_SCAN_WIDE
            jobs = []
            for _, idxs in groups:
                H = Hs[idxs[0]]
                if len({Hs[i] for i in idxs}) == 1 and H % 256 == 0 and 2 * len(idxs) <= M3T_MAX_SCANS \
                        and lib().m3t_gru_scan_progress_ok(2 * len(idxs), H, B, T, bfl, 1):
                    nfl = int(lib().m3t_gru_bwd_prepare_floats(H))
                    for l in range(L):
                        ws_, outs_ = [], []
                        for si in idxs:
                            for d in (0, 1):
                                buf = new(nfl)
                                wfq[(l, si, d)] = buf
                                ws_.append(params[si][(2 * l + d) * 4 + 1])
                                outs_.append(buf)
                        jobs.append((H, ws_, outs_))
            if jobs:
                wg1 = wgrad_streams(dev)[1]
                wg1.wait_stream(main)                 # (the buffers come from this stream's allocator: their previous users are done)
                with on_stream(wg1):
                    for H, ws_, outs_ in jobs:
                        n_ = len(ws_)
                        _lib.check(lib().m3t_gru_bwd_prepare((C.c_void_p * n_)(*[t.data_ptr() for t in ws_]), n_, H, 1,
                                                             (C.c_void_p * n_)(*[t.data_ptr() for t in outs_]), _stream()), "m3t_gru_bwd_prepare")
                    ctx.wfq_ev = torch.cuda.Event()
                    ctx.wfq_ev.record(wg1)
                ctx.wfq = wfq
        ctx.n_stacks, ctx.L, ctx.Hs, ctx.B, ctx.T, ctx.prec = n_stacks, L, Hs, B, T, prec
        ctx.unit_mask = unit_mask
        ctx.concurrent = concurrent
        ctx.cat = (cat_lo, cat_hi) if cat_buf is not None else (0, 0)
        saved = []
        for l in range(L):
            for s in range(n_stacks):
                saved += [outs[l][s], gates[l][s]]
        ctx.has_fslots = fslots is not None
        ctx.wslot = wslot if fslots is not None else None
        if fslots is not None:
            for st_ in (side_stream(dev),):
                fslots.record_stream(st_)
            saved.append(fslots)
        ctx.save_for_backward(*(list(tensors) + saved))
        result = []
        for s in range(n_stacks):
            if cat_buf is not None and cat_lo <= s < cat_hi:
                result += [cat_buf if s == cat_lo else new(0), h_ns[s]]
            else:
                result += [outs[L - 1][s], h_ns[s]]
        return tuple(result)

    @staticmethod
    def backward(ctx, *grads):
        n_stacks, L, Hs, B, T = ctx.n_stacks, ctx.L, ctx.Hs, ctx.B, ctx.T
        per = 1 + 8 * L
        st = ctx.saved_tensors
        tensors, acts = st[:n_stacks * per], st[n_stacks * per:]
        fslots = None
        if ctx.has_fslots:
            fslots, acts = acts[-1], acts[:-1]
        params = [list(tensors[s * per + 1:(s + 1) * per]) for s in range(n_stacks)]
        xs = [tensors[s * per].contiguous() for s in range(n_stacks)]

        def layer_io(l, s):   # (input, out, gates) of stack s at layer l
            out, gts = acts[(l * n_stacks + s) * 2], acts[(l * n_stacks + s) * 2 + 1]
            inp = xs[s] if l == 0 else acts[((l - 1) * n_stacks + s) * 2]
            return inp, out, gts

        dev = params[0][0].device
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        out_grads = [None] * (n_stacks * per)
        sunk = [False] * (n_stacks * per)
        douts, dhns = [], []
        cat_lo, cat_hi = ctx.cat
        dcat = None
        if cat_hi > cat_lo:                      # the group's gradient arrives as one [B, T, sum 2H] tensor: read in place, strided
            g = grads[2 * cat_lo]
            dcat = (torch.zeros(B, T, sum(2 * Hs[s] for s in range(cat_lo, cat_hi)), dtype=torch.float32, device=dev) if g is None
                    else _req(g.contiguous(), "dout"))
        off = 0
        for s in range(n_stacks):
            g, gh = grads[2 * s], grads[2 * s + 1]
            if dcat is not None and cat_lo <= s < cat_hi:
                douts.append(dcat[..., off:off + 2 * Hs[s]])
                off += 2 * Hs[s]
            else:
                douts.append(torch.zeros(B, T, 2 * Hs[s], dtype=torch.float32, device=dev) if g is None
                             else _req(g.contiguous(), "dout"))
            dhns.append(None if gh is None else _req(gh.contiguous(), "dh_n"))
        # all scratch and result buffers on the caller's stream (see forward)
        dgx = [[new(B, T, 6 * Hs[s]) for s in range(n_stacks)] for _ in range(L)]
        dgh = [[new(2, B, T, 3 * Hs[s]) for s in range(n_stacks)] for _ in range(L)]
        dh = [[new(2, B, Hs[s]) for s in range(n_stacks)] for _ in range(L)]
        dbp = [[new(2, B, 4, Hs[s]) for s in range(n_stacks)] for _ in range(L)]      # per-clip bias-gradient sums (scan output)
        # H % 16 == 0 everywhere: the scans build their weight fragments straight from w_hh (no transposes in front of them)
        direct_whh = all(h % 16 == 0 for h in Hs) and all(p.is_contiguous() for prm in params for p in prm)
        wht = [[[None if direct_whh else new(Hs[s], 3 * Hs[s]) for _ in (0, 1)] for s in range(n_stacks)] for _ in range(L)]
        need_dx = [[l > 0 or ctx.needs_input_grad[5 + s * per] for s in range(n_stacks)] for l in range(L)]
        dinp = [[torch.empty_like(layer_io(l, s)[0]) if need_dx[l][s] else None for s in range(n_stacks)] for l in range(L)]
        for s in range(n_stacks):
            for l in range(L):
                for d in (0, 1):
                    w_ih, w_hh = params[s][(2 * l + d) * 4], params[s][(2 * l + d) * 4 + 1]
                    base = s * per + 1 + (2 * l + d) * 4
                    for j, prm in enumerate(params[s][(2 * l + d) * 4:(2 * l + d) * 4 + 4]):
                        sink = _take_sink(prm)             # gradient written straight into the flat buffer
                        out_grads[base + j] = sink if sink is not None else torch.empty_like(prm)
                        sunk[base + j] = sink is not None
        main = cur_stream()
        groups = _stream_groups(Hs, B)
        cur = {s: douts[s] for s in range(n_stacks)}
        prec = ctx.prec
        wfq = ctx.wfq or {}                  # fragments of the wide backward scans, written during forward (m3t_gru_bwd_prepare)
        if ctx.wfq_ev is not None:
            main.wait_event(ctx.wfq_ev)
        # fp16x3 products: every backward scan raises one magnitude slot per direction (max |dgx|, |dgh|); the inputs' and the
        # weights' slots come from the forward pass, the recurrent states are GRU outputs (|h| <= 1)
        # (per scan: [0] the whole scan, [1 + j] the j-th time window in the order the scan walks them -- written by launches with progress marks)
        NSL = _CHUNK_WINDOWS + 1
        bslots = amax_slots(2 * L * n_stacks * NSL, dev) if fslots is not None else None
        one = amax_one(dev)

        def bslot(l, s, d, j=-1):
            return None if bslots is None else bslots.data_ptr() + 8 * (((l * n_stacks + s) * 2 + d) * NSL + 1 + j)

        def fslot_x(l, s):
            return None if fslots is None else (fslots.data_ptr() + 8 * s if (l == 0 and not (ctx.unit_mask >> s) & 1) else one)

        def fslot_w(l, s, d):
            return None if fslots is None else ctx.wslot[(l, s, d)]

        def chunk_plan(l, idxs):
            """time bounds for the direction-split, time-chunked data gradients of level l (stacks idxs, one H), or None"""
            if bslots is None or len({Hs[i] for i in idxs}) != 1 or 2 * len(idxs) > M3T_MAX_SCANS or not any(need_dx[l][s] for s in idxs):
                return None
            H = Hs[idxs[0]]
            if H % 64 != 0 or any(need_dx[l][s] and layer_io(l, s)[0].shape[-1] % 64 != 0 for s in idxs) or not direct_whh:
                return None
            fl = _scan_flags(dev) | prec | _lib.M3T_SCAN_WHH | _lib.M3T_SCAN_WIDE
            if not lib().m3t_gru_scan_progress_ok(2 * len(idxs), H, B, T, fl, 1):
                return None
            tb = _chunk_bounds(B, T, backward=True)
            return tb if (tb is not None and len(tb) + 1 <= _CHUNK_WINDOWS) else None

        def dx_pieces(l, idxs, tb, need, scan_stream):
            """level l's data gradients dX = dgx_fwd W_ih_fwd + dgx_rev W_ih_rev as (direction x time window) pieces on weight-gradient stream 1
            (idle while the chain's scans run), each behind the progress mark of the backward scan that writes its dgx; `scan_stream` waits
            for the last piece.  The backward scan of the FORWARD direction walks time downwards."""
            wg = wgs[1]
            ctr = _progress_counters(dev)
            ev_end = torch.cuda.Event()
            ev_end.record(scan_stream)
            lo, hi = [0] + tb, tb + [T]
            n = len(tb)
            with on_stream(wg):
                ended = False
                for dirn, k, w, second in _arrivals(tb, T):
                    if k is None:
                        if not ended:
                            wg.wait_event(ev_end)
                            ended = True
                    else:
                        _wait_progress(ctr, dirn, need[dirn * n + k])
                    d = 1 - dirn                                   # up-walking backward scans belong to the reverse direction
                    j = w if dirn == 0 else n - w                  # the window's index in the order that scan walks them (its magnitude slot)
                    by_shape = {}
                    for si in idxs:
                        if need_dx[l][si]:
                            I = layer_io(l, si)[0].shape[-1]
                            by_shape.setdefault(I, []).append((dgx[l][si], d * 3 * Hs[si], params[si][(2 * l + d) * 4], 0, dinp[l][si], 0, None,
                                                               second, bslot(l, si, d, j), fslot_w(l, si, d)))
                    for I, probs in by_shape.items():      # ONE launch per mark and input width
                        H = Hs[idxs[0]]
                        sgemm_window_batch(probs, 0, B, hi[w] - lo[w], T, lo[w], I, 3 * H, 6 * H, I, I, prec=prec)
                ev = torch.cuda.Event()
                ev.record(wg)
            scan_stream.wait_event(ev)
            for si in idxs:
                if need_dx[l][si]:
                    cur[si] = dinp[l][si]

        def start_marks(idxs):
            """round 5 (M3T_SCAN_FIRST): marks at step 8 of both directions -- "every workgroup of the launch is resident and stepping" -- for a
            backward launch that carries no chunk marks.  The weight gradients of the level before are held back until then: queued on their
            stream as soon as the data gradients are done, their workgroups (50 KB of LDS each) kept refilling the CUs the scan's
            workgroups (a whole CU each) were waiting for -- the scan started 0.1-0.2 ms late (HIP events around the launch vs the kernel's
            own duration)."""
            if not SCAN_FIRST[0] or bslots is None or len({Hs[i] for i in idxs}) != 1 or 2 * len(idxs) > M3T_MAX_SCANS or T < 64 or not direct_whh:
                return None
            fl = _scan_flags(dev) | prec | _lib.M3T_SCAN_WHH | _lib.M3T_SCAN_WIDE
            if not lib().m3t_gru_scan_progress_ok(2 * len(idxs), Hs[idxs[0]], B, T, fl, 1):
                return None
            return [8, T - 8]

        held = {}                          # chain (stream role) -> weight-gradient closure of the level before, waiting for the next scan's start

        def release_held(key, started):
            fn = held.pop(key, None)
            if fn is None:
                return
            if started is not None:        # (need table of a launch with start marks: up- and down-walking scans at step 8)
                ctr_, need_ = started
                with on_stream(wgs[0]):
                    _wait_progress(ctr_, 0, need_[0])
                    _wait_progress(ctr_, 1, need_[2])
            fn()

        def level_scan(l, idxs, after=None, wide=False, progress=None):
            """every BACKWARD level asks for the wide form (the library applies it where it exists: H = 512 in the fp16x3 mode), not only
            the level that makes room for the audio scans: the wide backward kernel is no slower than the narrow one on half the CUs
            (fusion level 3.15 vs 3.30 us per step, exchange served by one XCD's L2) and the weight-gradient GEMMs get the rest;
            forward, the wide form costs 0.2 us per step (2.49 vs 2.30) and only the level that must make room takes it"""
            descs = []
            for s in idxs:
                H = Hs[s]
                inp, out, gts = layer_io(l, s)
                for d in (0, 1):
                    w_hh = params[s][(2 * l + d) * 4 + 1]
                    if not direct_whh:
                        _lib.check(lib().m3t_transpose(_p(w_hh), 3 * H, H, H, _p(wht[l][s][d]), 3 * H, _stream()),
                                   "m3t_transpose")
                    base = s * per + 1 + (2 * l + d) * 4
                    descs.append(GruBwdDesc(_vp(cur[s]), _vp(out), _vp(gts, d * B * T * 4 * H),
                                            _vp(w_hh if direct_whh else wht[l][s][d]),
                                            _vp(dhns[s], (2 * l + d) * B * H) if dhns[s] is not None else None,
                                            _vp(dgx[l][s]), _vp(dgh[l][s], d * B * T * 3 * H), _vp(dh[l][s], d * B * H),
                                            _vp(dbp[l][s], d * B * 4 * H), _vp(out_grads[base + 2]), _vp(out_grads[base + 3]),
                                            H, d, out.stride(1), d * H, 6 * H, d * 3 * H, bslot(l, s, d),
                                            _vp(wfq[(l, s, d)]) if (l, s, d) in wfq else None))      # (dout and out share the layout)
            chain = _ws_tag(dev)
            sm = None
            if progress is None and chain in held:
                sm = start_marks(idxs)
            need = _scan_bwd(descs, B, T, prec | (_lib.M3T_SCAN_WHH if direct_whh else 0) | _lib.M3T_SCAN_WIDE, after,
                             None if (progress is None and sm is None) else (_progress_counters(dev), progress if progress is not None else sm))
            if progress is not None:
                pending[(l, tuple(idxs))] = (progress, need)
            release_held(chain, (_progress_counters(dev), need) if sm is not None else None)
            if RANGE_PROBE[0] is not None:
                for s in idxs:
                    for d in (0, 1):
                        _probe_range("dgx l%d s%d d%d H%d" % (l, s, d, Hs[s]), dgx[l][s].view(B * T, 6 * Hs[s])[:, d * 3 * Hs[s]:(d + 1) * 3 * Hs[s]])
                        _probe_range("dgh l%d s%d d%d H%d" % (l, s, d, Hs[s]), dgh[l][s][d].view(B * T, 3 * Hs[s]))

        pending = {}                       # (l, idxs) -> (time bounds, need) of a scan launched with progress marks

        def level_dx(l, idxs):           # on the chain: feeds the next level's scan
            pg = pending.pop((l, tuple(idxs)), None)
            if pg is not None:           # round 5: the pieces run under the scan; what is left behind it is the last window of each direction
                dx_pieces(l, idxs, pg[0], pg[1], cur_stream())
                return
            for s in idxs:
                H = Hs[s]
                inp, out, gts = layer_io(l, s)
                I = inp.shape[-1]
                if need_dx[l][s]:
                    for d in (0, 1):
                        sgemm(0, 0, B * T, I, 3 * H, dgx[l][s], d * 3 * H, 6 * H, params[s][(2 * l + d) * 4], 0, I,
                              dinp[l][s], 0, I, accumulate=(d == 1), prec=prec, amax=(bslot(l, s, d), fslot_w(l, s, d)))
                    cur[s] = dinp[l][s]

        wgs = wgrad_streams(dev)
        rr = [0]

        def _level_dw(l, idxs, pool):           # off the chain: only the optimizer reads these
            """the weight-gradient GEMMs of (l, idxs) on weight-gradient stream 0 (every stream has waited for the scan; a second
            concurrent GEMM stream beside the chain takes CUs from the data-gradient GEMMs and stretches the scans: 17.77 vs 17.21
            ms, round 2).  spread: the LAST weight gradients of the pass -- nothing else is queued on the level's own stream any
            more -- alternate between that stream and weight-gradient stream 1: FlatGradDDP.finish() then follows the last GEMM
            on the main stream without a cross-queue hop (a queue that has idled for ~2 ms answers an event ~110 us late: 15.69
            vs 15.81 ms per step, round 3; more tail streams do not help: 2 / 3 / 4 streams 15.83 / 15.83 / 15.93, the tail is
            bound by GEMM throughput, not by concurrency)"""
            for s in idxs:
                H = Hs[s]
                inp, out, gts = layer_io(l, s)
                I = inp.shape[-1]
                for d in (0, 1):
                    base = s * per + 1 + (2 * l + d) * 4
                    dw_ih, dw_hh = out_grads[base:base + 2]      # the bias gradients come out of the scan itself
                    goff = d * B * T * 3 * H
                    # tail: [the stream this level runs on, weight-gradient stream 1] -- after the last level nothing else is queued on
                    # the level's own stream, and finalize (FlatGradDDP.finish) then follows its last GEMM with no cross-queue hop
                    with (on_stream(pool[rr[0] % len(pool)]) if pool is not None else _NULL):
                        if T > 1:
                            # (K = B (T-1) need not be a multiple of the 32-deep k tile -- 8 clips x 64 frames: K = 504 -- since round 6: the
                            # segmented reduction's ragged last tile reads zeros.  Until then: a zero-padded shifted copy of the states per
                            # scan, 40 fill / copy launches per C5 step)
                            # dW_hh = sum_{b,t} dgh[b,t]^T h_prev(b,t): forward pairs (t, t-1), reverse pairs (t, t+1)
                            a_off, b_off = (1, 0) if d == 0 else (0, 1)
                            sgemm(1, 0, 3 * H, H, B * (T - 1), dgh[l][s], goff, 3 * H, out, d * H, out.stride(1), dw_hh, 0, H,
                                  seg=(T - 1, T, a_off, b_off), prec=prec, amax=(bslot(l, s, d), one if bslots is not None else None))
                        else:
                            dw_hh.zero_()
                    rr[0] += 1
                    with (on_stream(pool[rr[0] % len(pool)]) if pool is not None else _NULL):
                        sgemm(1, 0, 3 * H, I, B * T, dgx[l][s], d * 3 * H, 6 * H, inp, 0, I, dw_ih, 0, I, prec=prec,
                              amax=(bslot(l, s, d), fslot_x(l, s)))
                    rr[0] += 1

        def level_dw(l, idxs, spread=False, wg_i=0):
            # one stream switch per level (not per GEMM: ~40 context switches of ~8 us of host time per step) unless the level spreads
            if spread:
                _level_dw(l, idxs, [cur_stream(), wgs[1]])
            else:
                with on_stream(wgs[wg_i]):
                    _level_dw(l, idxs, None)

        # round 6: with two concurrent chains the LIGHT stack's trailing weight gradients go to weight-gradient stream 1 -- stream 0 carries the
        # heavy levels' and was the last to finish (7.04 ms after the loss against 6.81 / 6.88 on the other two: the clip waited for it)
        light_idxs = groups[1][1] if (LIGHT_DW_STREAM1[0] and ctx.concurrent and _interleaved(groups)) else None

        for w_ in wgs:
            w_.wait_stream(main)

        def level_gemms(l, idxs, last=False):
            """after the scan of (l, idxs) on the current stream: the data gradients in line (they are what the chain -- or the
            light scans' own chain -- waits for), then the weight gradients on their streams (started earlier they would split
            the CUs the scan beside them leaves free: 17.56 vs 17.78 ms per step, round 2)"""
            level_dx(l, idxs)
            ev = torch.cuda.Event()
            ev.record(cur_stream())
            wg_i = 1 if (light_idxs is not None and list(idxs) == list(light_idxs)) else 0
            for w_ in (wgs if last else [wgs[wg_i]]):      # (the streams level_dw uses: stream 1 carries the chunked data gradients of the chain)
                w_.wait_event(ev)
            if last or not SCAN_FIRST[0]:
                level_dw(l, idxs, spread=last, wg_i=wg_i)
            else:
                # held until the NEXT scan of this chain has been launched (level_scan -> release_held): the scan first, then the GEMMs that
                # only the optimizer waits for
                held[_ws_tag(dev)] = lambda: level_dw(l, idxs, spread=False, wg_i=wg_i)

        if ctx.concurrent and _interleaved(groups):
            # as in forward (round 4): the heavy level on half the CUs, the light stack's scans at the same time on others; two
            # independent chains scan -> data gradients -> scan, the weight gradients of both trail on their own streams
            heavy, light = groups[0][1], groups[1][1]
            side = side_stream(dev)
            side.wait_stream(main)
            _FENCED[0] = True
            try:
                for l in range(L - 1, -1, -1):
                    ev_al = None
                    if ALIGN_LIGHT[0] and _ALIGN_BWD and l < L - 1:       # (as in forward: the light scan beside the heavy scan, not beside the heavy data-gradient GEMMs)
                        ev_al = torch.cuda.Event()
                        ev_al.record(main)
                    level_scan(l, heavy, None, True, chunk_plan(l, heavy))
                    with on_stream(side):
                        level_scan(l, light, ev_al)
                    level_gemms(l, heavy, last=(l == 0))
                    with on_stream(side):
                        level_gemms(l, light, last=(l == 0))
            finally:
                _FENCED[0] = False
            main.wait_stream(side)
        elif _interleaved(groups):
            # as in forward: the light group's backward scans run on the side stream beside the heavy group's GEMMs
            # (data gradients; the weight gradients of both groups trail on their own stream), persistent scans
            # strictly alternating
            heavy, light = groups[0][1], groups[1][1]
            side = side_stream(dev)
            side.wait_stream(main)
            ev_light = None
            _FENCED[0] = True
            try:
                if L == 2:
                    # persistent scans in the order H1, L1, H0, L0: the light layer-1 scan beside the heavy layer-1 data-gradient
                    # GEMMs, and the pass ENDS with a light scan (64 workgroups), beside which the weight-gradient GEMMs that
                    # are left have three quarters of the chip -- instead of running alone after the last heavy scan
                    level_scan(1, heavy, None)
                    ev_h1 = torch.cuda.Event()
                    ev_h1.record(main)
                    level_gemms(1, heavy)
                    with on_stream(side):
                        level_scan(1, light, ev_h1)
                        ev_l1 = torch.cuda.Event()
                        ev_l1.record(side)
                        level_gemms(1, light)
                    level_scan(0, heavy, ev_l1)
                    ev_h0 = torch.cuda.Event()
                    ev_h0.record(main)
                    level_gemms(0, heavy, last=True)
                    with on_stream(side):
                        level_scan(0, light, ev_h0)
                        level_gemms(0, light, last=True)
                else:
                  for l in range(L - 1, -1, -1):
                    level_scan(l, heavy, ev_light)
                    ev_heavy = torch.cuda.Event()
                    ev_heavy.record(main)
                    level_gemms(l, heavy)
                    with on_stream(side):
                        level_scan(l, light, ev_heavy)
                        ev_light = torch.cuda.Event()
                        ev_light.record(side)
                        level_gemms(l, light)
            finally:
                _FENCED[0] = False
            main.wait_stream(side)
        else:
            for kind, idxs in groups:
                stream = main
                if kind == "side":
                    stream = side_stream(dev)
                    stream.wait_stream(main)
                with on_stream(stream):
                    for l in range(L - 1, -1, -1):
                        level_scan(l, idxs, progress=chunk_plan(l, idxs) if (kind == "main" and len(groups) == 1) else None)
                        level_gemms(l, idxs)
            for kind, _ in groups:
                if kind == "side":
                    main.wait_stream(side_stream(dev))
        for key_ in list(held):
            release_held(key_, None)
        weights_sunk = all(sunk[s * per + 1 + (2 * l + d) * 4 + j] for s in range(n_stacks) for l in range(L) for d in (0, 1) for j in (0, 1))
        if weights_sunk:
            # every weight gradient goes straight into the flat gradient buffer, which nobody reads before
            # FlatGradDDP.finish(): leave the weight-gradient streams running (join_wgrad() there) instead of waiting
            # for their tail here.  What they still read must outlive this call on THEIR stream.
            for l in range(L):
                for s in range(n_stacks):
                    for t in (dgx[l][s], dgh[l][s]) + tuple(layer_io(l, s)[:2]):
                        for w_ in wgs:
                            t.record_stream(w_)
            for t in (bslots, fslots):
                if t is not None:
                    for w_ in wgs:
                        t.record_stream(w_)
            _WGRAD_PENDING[(dev.type, dev.index)] = True
        else:
            for w_ in wgs:
                main.wait_stream(w_)
        for s in range(n_stacks):
            out_grads[s * per] = dinp[0][s]
        return (None, None, None, None, None) + tuple(None if sunk[i] else g for i, g in enumerate(out_grads))


def multi_bigru(stacks, cat=None):
    """stacks: list of (x [B,T,I], flat_params [w_ih,w_hh,b_ih,b_hh per (layer,dir)], L).
    Returns list of (out [B,T,2H], h_n [2L,B,H])."""
    L = stacks[0][2]
    flat = []
    for x, prm, l in stacks:
        if l != L:
            raise M3THipError("grouped GRU stacks must have the same depth")
        flat.append(x)
        flat.extend(prm)
    lo, hi = cat if cat is not None else (0, 0)
    unit_mask = sum(1 << i for i, (x, _, _) in enumerate(stacks) if _is_unit(x))
    res = _MultiBiGRU.apply(len(stacks), L, lo, hi, unit_mask, *flat)
    for i in range(len(stacks)):
        if res[2 * i].numel():
            res[2 * i]._m3t_unit = res[2 * i]._version + 1          # (the tag is the version it holds for, + 1: see _is_unit)  |h| <= 1: a consumer's fp16x3 contraction takes the constant magnitude slot (linear, multi_bigru)
    return [(res[2 * i], res[2 * i + 1]) for i in range(len(stacks))]


# ----------------------------------------------------------------------------- AttFusion reduction
class _AttFuse(torch.autograd.Function):
    """softmax([sigmoid(s_v), sigmoid(s_a)]) weighted sum (reference models/att_fusion.py:21-25)."""

    @staticmethod
    def forward(ctx, s_v, s_a, x_v, x_a):
        s_v, s_a = _req(s_v.contiguous(), "s_v"), _req(s_a.contiguous(), "s_a")
        x_v, x_a = _req(x_v.contiguous(), "x_v"), _req(x_a.contiguous(), "x_a")
        D = x_v.shape[-1]
        rows = x_v.numel() // D
        if x_a.shape != x_v.shape or s_v.numel() != rows or s_a.numel() != rows:
            raise M3THipError("att_fuse: shape mismatch")
        f = torch.empty_like(x_v)
        # algorithmic traffic: read x_v, x_a, write f -- 3 D floats per frame (+ the two scores)
        with _Timed("att_fuse_fwd_kernel", 1, 0, nbytes=4.0 * rows * (3 * D + 2)):
            _lib.check(lib().m3t_att_fuse_fwd(_p(s_v), _p(s_a), _p(x_v), _p(x_a), _p(f), rows, D, _stream()), "m3t_att_fuse_fwd")
        ctx.save_for_backward(s_v, s_a, x_v, x_a)
        return f

    @staticmethod
    def backward(ctx, df):
        s_v, s_a, x_v, x_a = ctx.saved_tensors
        df = _req(df.contiguous(), "df")
        D = x_v.shape[-1]
        rows = x_v.numel() // D
        ds_v, ds_a = torch.empty_like(s_v), torch.empty_like(s_a)
        dx_v, dx_a = torch.empty_like(x_v), torch.empty_like(x_a)
        # read df, x_v, x_a, write dx_v, dx_a: 5 D floats per frame (+ scores and their gradients)
        with _Timed("att_fuse_bwd_kernel", 1, 0, nbytes=4.0 * rows * (5 * D + 4)):
            _lib.check(lib().m3t_att_fuse_bwd(_p(df), _p(s_v), _p(s_a), _p(x_v), _p(x_a), _p(ds_v), _p(ds_a), _p(dx_v),
                                              _p(dx_a), rows, D, _stream()), "m3t_att_fuse_bwd")
        return ds_v, ds_a, dx_v, dx_a


def att_fuse(s_v, s_a, x_v, x_a):
    return _AttFuse.apply(s_v, s_a, x_v, x_a)


# ----------------------------------------------------------------------------- VA loss
class _VALoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_hat, valence, arousal, class_expr, expr_valid, iv, ia, n_expr, w_v, w_a, expr_w, use_mse):
        y = _req(y_hat.contiguous(), "y_hat")
        Cc = y.shape[-1]
        rows = y.numel() // Cc
        val = _req(valence.contiguous().float(), "valence")
        aro = _req(arousal.contiguous().float(), "arousal")
        cls = vld = None
        if n_expr > 0:
            cls = class_expr.contiguous().long()
            vld = expr_valid.contiguous().to(torch.uint8)
            if not cls.is_cuda or not vld.is_cuda:
                raise M3THipError("labels must be device tensors")
        stats = torch.empty(8, dtype=torch.float32, device=y.device)
        dy = torch.empty_like(y)
        ws = workspace(y.device)
        # three sweeps over y [rows, C] (sums, centred moments, gradient) + the labels each time + one write of dy
        with _Timed("va_loss_kernels", 3, 0, nbytes=4.0 * rows * (3 * Cc + 3 * 2 + Cc) + (3.0 * rows * 9 if n_expr > 0 else 0.0)):
            rc = lib().m3t_va_loss(_p(y), rows, Cc, iv, ia, _p(val), _p(aro),
                                   C.c_void_p(cls.data_ptr()) if cls is not None else None,
                                   C.c_void_p(vld.data_ptr()) if vld is not None else None,
                                   n_expr, w_v, w_a, expr_w, int(use_mse), _p(stats), _p(dy), _p(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "m3t_va_loss")
        ctx.save_for_backward(dy)
        ctx.mark_non_differentiable(stats)
        return stats[0], stats

    @staticmethod
    def backward(ctx, g_loss, g_stats):
        (dy,) = ctx.saved_tensors
        return (dy * g_loss,) + (None,) * 11


def va_loss(y_hat, valence, arousal, class_expr=None, expr_valid=None, iv=None, ia=None, n_expr=0,
            w_v=0.5, w_a=0.5, expr_w=0.8, use_mse=False):
    """Loss of AffWild2VA.training_step (reference models/model.py:146-182).  Returns
    (loss, stats[8]) with stats = loss, loss_v, loss_a, loss_expr, n_valid, n_correct, ccc_v, ccc_a."""
    Cc = y_hat.shape[-1]
    iv = Cc - 2 if iv is None else iv
    ia = Cc - 1 if ia is None else ia
    return _VALoss.apply(y_hat, valence, arousal, class_expr, expr_valid, iv, ia, n_expr, float(w_v), float(w_a),
                         float(expr_w), bool(use_mse))


# ----------------------------------------------------------------------------- TCN
class _BctToBtc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x.contiguous(), "x")
        B, Cc, T = x.shape
        y = torch.empty(B, T, Cc, dtype=x.dtype, device=x.device)
        _lib.check(lib().m3t_bct_to_btc(_p(x), _p(y), B, Cc, T, _stream()), "m3t_bct_to_btc")
        return y

    @staticmethod
    def backward(ctx, dy):
        return _BtcToBct.apply(dy)


class _BtcToBct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x.contiguous(), "x")
        B, T, Cc = x.shape
        y = torch.empty(B, Cc, T, dtype=x.dtype, device=x.device)
        _lib.check(lib().m3t_btc_to_bct(_p(x), _p(y), B, T, Cc, _stream()), "m3t_btc_to_bct")
        return y

    @staticmethod
    def backward(ctx, dy):
        return _BctToBtc.apply(dy)


def bct_to_btc(x):
    return _BctToBtc.apply(x)


def btc_to_bct(x):
    return _BtcToBct.apply(x)


def _conv(x, w_t, bias, res, mask, pre, B, T, Ci, Co, K, dil, act, anti, prec=0, drop=(0.0, 0), amax=(None, None), amax_y=None):
    """amax_y: address of a magnitude slot the epilogue raises to max |y| (armed AFTER the output allocation)"""
    y = torch.empty(B, T, Co, dtype=torch.float32, device=x.device)
    amax_out(amax_y)
    rc = lib().m3t_conv1d_fwd_scaled(_p(x), _p(w_t), _p(bias), _p(res), _p(mask), _p(y), _p(pre), B, T, Ci, Co, K, dil, 0,
                                     act, anti, float(drop[0]), int(drop[1]), prec, amax[0], amax[1], _stream())
    _lib.check(rc, "m3t_conv1d_fwd")
    return y


def _conv_wgrad(dy, x, dw_t, B, T, Ci, Co, K, dil, ws, prec, amax=(None, None)):
    _lib.check(lib().m3t_conv1d_wgrad_scaled(_p(dy), _p(x), _p(dw_t), B, T, Ci, Co, K, dil, 0, _p(ws), ws.numel() * 4, prec,
                                             amax[0], amax[1], _stream()), "m3t_conv1d_wgrad")


class _TemporalBlock(torch.autograd.Function):
    """One TemporalBlock (reference models/tcn.py:16-46) on channel-last activations:
    weight-norm -> dilated causal conv + bias + ReLU [+dropout mask] -> same again ->
    + residual (identity or 1x1 conv) -> ReLU, with bias/ReLU/residual fused in the conv epilogue."""

    @staticmethod
    def forward(ctx, x, v1, g1, b1, v2, g2, b2, wd, bd, dilation, m1, m2, drop_p=0.0, seed1=0, seed2=0):
        x = _req(x.contiguous(), "x")
        for t in (v1, g1, b1, v2, g2, b2):
            _req(t, "tcn parameter")
        B, T, Ci = x.shape
        Co, _, K = v1.shape
        dev = x.device
        w1t = torch.empty(K, Co, Ci, dtype=torch.float32, device=dev)
        w2t = torch.empty(K, Co, Co, dtype=torch.float32, device=dev)
        n1 = torch.empty(Co, dtype=torch.float32, device=dev)
        n2 = torch.empty(Co, dtype=torch.float32, device=dev)
        prec = ctx.prec = _PREC[0]
        d1, d2 = (drop_p, seed1), (drop_p, seed2)        # in-kernel Philox masks (drop_p > 0) instead of the mask tensors m1 / m2
        # fp16x3 products: magnitude slots 0 x, 1 w1t, 2 w2t, 3 wd, 4 h1, 8 y (forward), 5 ds, 6 da2, 7 da1 (backward).  Round 4: raised by
        # the PRODUCERS (m3t_amax_out: the weight-norm kernels, the conv epilogues, the mask kernels) instead of measuring launches -- 13
        # launches per C1 step, 10 % of its kernel time; what is left is one launch for a block whose input arrives
        # without a slot (x, and the raw down-sampling weight wd).  A slot is only ever handed to a kernel once something has raised it
        # (a zero slot that was never raised reads as "all-zero operand"); an operand whose slot is not in `ok` is measured
        # by the library.
        slots, ok, x_ext = None, set(), None
        if (prec & _lib.M3T_GEMM_F16X3) and (B * T) % 128 == 0 and Co % 128 == 0 and Ci % 32 == 0:
            slots = amax_slots(9, dev)
        sp = slots.data_ptr() if slots is not None else None
        amax_out(sp + 8 if slots is not None else None)
        _lib.check(lib().m3t_weight_norm_fwd(_p(v1), _p(g1), _p(w1t), _p(n1), Co, Ci, K, _stream()), "m3t_weight_norm_fwd")
        amax_out(sp + 16 if slots is not None else None)
        _lib.check(lib().m3t_weight_norm_fwd(_p(v2), _p(g2), _p(w2t), _p(n2), Co, Co, K, _stream()), "m3t_weight_norm_fwd")
        if slots is not None:
            ok.update((1, 2))
            x_ext = _X_EXT[0]                             # (slots tensor, index) left on x by its producer (the previous block's epilogue)
            todo = ([] if x_ext is not None else [(x, sp)]) + ([(wd, sp + 24)] if wd is not None else [])
            if todo and measure_amax(todo):
                ok.update(([] if x_ext is not None else [0]) + ([3] if wd is not None else []))
        sl = lambda i: ((x_ext[0].data_ptr() + 8 * x_ext[1]) if (i == 0 and x_ext is not None) else
                        (sp + 8 * i) if (slots is not None and i in ok) else None)
        ctx.x_ext = x_ext
        h1 = _conv(x, w1t, b1, None, m1, None, B, T, Ci, Co, K, dilation, 1, 0, prec, d1, amax=(sl(0), sl(1)),
                   amax_y=sp + 32 if slots is not None else None)
        if slots is not None:
            ok.add(4)
        if wd is not None:
            res = torch.empty(B, T, Co, dtype=torch.float32, device=dev)
            sgemm(0, 1, B * T, Co, Ci, x, 0, Ci, wd, 0, Ci, res, 0, Co, bias=bd, prec=prec, amax=(sl(0), sl(3)))
        else:
            res = x
        ctx.slots_ok = ok
        a2 = torch.empty(B, T, Co, dtype=torch.float32, device=dev)
        y = _conv(h1, w2t, b2, res, m2, a2, B, T, Co, Co, K, dilation, 2, 0, prec, d2, amax=(sl(4), sl(2)),
                  amax_y=sp + 64 if slots is not None else None)
        if slots is not None:
            _LAST_OUT_SLOT[0] = (slots, 8)                 # temporal_block() hands it to whoever consumes y
        ctx.save_for_backward(x, v1, g1, v2, g2, wd, w1t, w2t, n1, n2, h1, a2, y, m1, m2, slots)
        ctx.dil, ctx.drops = dilation, (d1, d2)
        # parameter objects that own a gradient sink (FlatGradDDP): their gradients can be written straight into the flat buffer
        # on the weight-gradient stream, off the chain (see backward)
        prm = (v1, g1, b1, v2, g2, b2) + ((wd, bd) if wd is not None else ())
        ctx.sink_refs = prm if all(t is not None and id(t) in _GRAD_SINKS for t in prm) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, v1, g1, v2, g2, wd, w1t, w2t, n1, n2, h1, a2, y, m1, m2, slots = ctx.saved_tensors
        dy = _req(dy.contiguous(), "dy")
        B, T, Ci = x.shape
        Co, _, K = v1.shape
        dil, dev, prec = ctx.dil, x.device, ctx.prec
        ws = workspace(dev)
        ok = set(ctx.slots_ok)                       # slots something has raised (see forward)
        x_ext = ctx.x_ext
        sp = slots.data_ptr() if slots is not None else None
        sl = lambda i: ((x_ext[0].data_ptr() + 8 * x_ext[1]) if (i == 0 and x_ext is not None) else
                        (sp + 8 * i) if (slots is not None and i in ok) else None)
        ds = mask_pos(y, dy, amax=(sp + 40 if slots is not None else None))                         # through the block's output ReLU
        d1, d2 = ctx.drops
        da2 = mask_pos(a2, ds, m2, d2, amax=(sp + 48 if slots is not None else None))               # through dropout2 + relu2
        if slots is not None:
            ok.update((5, 6))
        dh1 = _conv(da2, w2t, None, None, None, None, B, T, Co, Co, K, dil, 0, 1, prec, amax=(sl(6), sl(2)))
        da1 = mask_pos(h1, dh1, m1, d1, amax=(sp + 56 if slots is not None else None))              # h1 > 0 <=> a1 > 0 (dropout keeps the sign)
        if slots is not None:
            ok.add(7)
        dw2t = torch.empty_like(w2t)
        dw1t = torch.empty_like(w1t)
        # Off the chain: when every parameter of the block has a gradient sink, the data gradient (what the previous block waits
        # for) is computed first, and the weight gradients (6 segmented GEMMs), the bias sums and the weight-norm backward run on
        # the weight-gradient stream straight into the flat gradient buffer (joined by FlatGradDDP.finish(), like _Linear's)
        sinks = None
        if (ctx.sink_refs is not None and x.is_cuda
                and all(ctx.needs_input_grad[i] for i in range(1, 7 if wd is None else 9))):
            got = [_take_sink(t) for t in ctx.sink_refs]
            if all(g_ is not None for g_ in got):
                sinks = got
            else:
                join_wgrad(dev)                      # a sink was taken earlier this step: the returned tensors get ADDED on this stream
        if sinks is not None:
            if wd is None:
                dx = _conv(da1, w1t, None, ds, None, None, B, T, Co, Ci, K, dil, 0, 1, prec, amax=(sl(7), sl(1)))
            else:
                dx = _conv(da1, w1t, None, None, None, None, B, T, Co, Ci, K, dil, 0, 1, prec, amax=(sl(7), sl(1)))
                sgemm(0, 0, B * T, Ci, Co, ds, 0, Co, wd, 0, Ci, dx, 0, Ci, accumulate=True, use_ws=False, prec=prec, amax=(sl(5), sl(3)))
            # the two convolutions' weight gradients on the two weight-gradient streams (round 4): each is K per-tap split-K GEMMs of 300-odd
            # workgroups -- less than half a round of the chip -- and the LAST block of the backward pass has nothing else to hide behind
            # (C1: 1.84 -> ~1.7 ms per step)
            main = cur_stream()
            wg, wg1 = wgrad_stream(dev, 0), wgrad_stream(dev, 1)
            wg.wait_stream(main)
            wg1.wait_stream(main)
            with on_stream(wg):
                wsw = workspace(dev)
                _conv_wgrad(da2, h1, dw2t, B, T, Co, Co, K, dil, wsw, prec, amax=(sl(6), sl(4)))
                colsum(da2, 0, B * T, Co, Co, sinks[5])
                _lib.check(lib().m3t_weight_norm_bwd(_p(dw2t), _p(v2), _p(g2), _p(n2), _p(sinks[3]), _p(sinks[4]), Co, Co, K, _stream()),
                           "m3t_weight_norm_bwd")
            with on_stream(wg1):
                wsw1 = workspace(dev)
                _conv_wgrad(da1, x, dw1t, B, T, Ci, Co, K, dil, wsw1, prec, amax=(sl(7), sl(0)))
                colsum(da1, 0, B * T, Co, Co, sinks[2])
                _lib.check(lib().m3t_weight_norm_bwd(_p(dw1t), _p(v1), _p(g1), _p(n1), _p(sinks[0]), _p(sinks[1]), Co, Ci, K, _stream()),
                           "m3t_weight_norm_bwd")
                if wd is not None:
                    sgemm(1, 0, Co, Ci, B * T, ds, 0, Co, x, 0, Ci, sinks[6], 0, Ci, prec=prec, amax=(sl(5), sl(0)))
                    colsum(ds, 0, B * T, Co, Co, sinks[7])
            for t in (da1, da2, h1, x, dw1t, dw2t, ds, n1, n2) + ((slots,) if slots is not None else ()) + ((x_ext[0],) if x_ext is not None else ()):
                t.record_stream(wg)
                t.record_stream(wg1)
            _WGRAD_PENDING[(dev.type, dev.index)] = True
            return (dx,) + (None,) * 14
        _conv_wgrad(da2, h1, dw2t, B, T, Co, Co, K, dil, ws, prec, amax=(sl(6), sl(4)))
        _conv_wgrad(da1, x, dw1t, B, T, Ci, Co, K, dil, ws, prec, amax=(sl(7), sl(0)))
        db1 = torch.empty(Co, dtype=torch.float32, device=dev)
        db2 = torch.empty(Co, dtype=torch.float32, device=dev)
        colsum(da1, 0, B * T, Co, Co, db1)
        colsum(da2, 0, B * T, Co, Co, db2)
        dv1, dg1 = torch.empty_like(v1), torch.empty_like(g1)
        dv2, dg2 = torch.empty_like(v2), torch.empty_like(g2)
        _lib.check(lib().m3t_weight_norm_bwd(_p(dw1t), _p(v1), _p(g1), _p(n1), _p(dv1), _p(dg1), Co, Ci, K, _stream()),
                   "m3t_weight_norm_bwd")
        _lib.check(lib().m3t_weight_norm_bwd(_p(dw2t), _p(v2), _p(g2), _p(n2), _p(dv2), _p(dg2), Co, Co, K, _stream()),
                   "m3t_weight_norm_bwd")
        dwd = dbd = None
        if wd is None:
            dx = _conv(da1, w1t, None, ds, None, None, B, T, Co, Ci, K, dil, 0, 1, prec, amax=(sl(7), sl(1)))      # + identity residual
        else:
            dx = _conv(da1, w1t, None, None, None, None, B, T, Co, Ci, K, dil, 0, 1, prec, amax=(sl(7), sl(1)))
            sgemm(0, 0, B * T, Ci, Co, ds, 0, Co, wd, 0, Ci, dx, 0, Ci, accumulate=True, use_ws=False, prec=prec, amax=(sl(5), sl(3)))
            dwd = torch.empty_like(wd)
            sgemm(1, 0, Co, Ci, B * T, ds, 0, Co, x, 0, Ci, dwd, 0, Ci, prec=prec, amax=(sl(5), sl(0)))
            dbd = torch.empty(Co, dtype=torch.float32, device=dev)
            colsum(ds, 0, B * T, Co, Co, dbd)
        return dx, dv1, dg1, db1, dv2, dg2, db2, dwd, dbd, None, None, None, None, None, None


def temporal_block(x_btc, v1, g1, b1, v2, g2, b2, wd, bd, dilation, m1=None, m2=None, drop_p=0.0, seeds=(0, 0)):
    """m1 / m2: explicit pre-scaled dropout masks, or drop_p > 0 with two 64-bit seeds: masks generated inside the conv epilogues
    (Philox4x32-10, include/m3t_hip.h) and regenerated in backward"""
    _LAST_OUT_SLOT[0] = None
    _X_EXT[0] = _tagged_amax(x_btc) if x_btc.is_contiguous() else None      # (autograd hands forward() a detached alias: read it here)
    try:
        y = _TemporalBlock.apply(x_btc, v1, g1, b1, v2, g2, b2, wd, bd, dilation, m1, m2, float(drop_p), int(seeds[0]), int(seeds[1]))
    finally:
        _X_EXT[0] = None
    if _LAST_OUT_SLOT[0] is not None:                # the conv epilogue raised max |y|: the next block takes it instead of measuring x
        y._m3t_amax = (_LAST_OUT_SLOT[0], y._version)              # (slot, the version it was measured at: _tagged_amax)  travels ON the tensor and dies with it (no registry holding activations alive)
        _LAST_OUT_SLOT[0] = None
    return y


_LAST_OUT_SLOT = [None]     # (slots tensor, index) of the last _TemporalBlock forward's output
_X_EXT = [None]             # the magnitude slot found on the input tensor of the temporal_block() call in progress (x._m3t_amax)


class _ConvBnRelu(torch.autograd.Function):
    """nn.Conv1d(C_in, C_out, k, 1, pad) -> nn.BatchNorm1d -> nn.ReLU on channel-last rows: one stage of the
    `tcn_simple` back-end (reference models/backbone.py:107-112, 214-222).  The conv is the implicit-GEMM kernel of
    the TCN with `lead = pad` frames of look-ahead; BatchNorm runs over the [B*T, C] rows (column statistics).
    running_mean / running_var are updated in place in training mode, as nn.BatchNorm1d does."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, run_mean, run_var, pad, training, momentum, eps):
        x = _req(x.contiguous(), "x")
        for t in (w, gamma, beta, run_mean, run_var):
            _req(t, "tcn_simple parameter")
        B, T, Ci = x.shape
        Co, _, K = w.shape
        if 2 * pad != K - 1:
            raise M3THipError("tcn_simple conv needs 2*padding == kernel_size-1 (got k=%d, pad=%d)" % (K, pad))
        dev = x.device
        w_t = transpose2d(w.detach().view(Co * Ci, K)).view(K, Co, Ci)          # tap-major [K][Co][Ci]
        a = torch.empty(B, T, Co, dtype=torch.float32, device=dev)
        prec = _PREC[0]
        rc = lib().m3t_conv1d_fwd(_p(x), _p(w_t), _p(b), None, None, _p(a), None, B, T, Ci, Co, K, 1, pad, 0, 0, 0.0, 0, prec, _stream())
        _lib.check(rc, "m3t_conv1d_fwd")
        y = torch.empty_like(a)
        mean = torch.empty(Co, dtype=torch.float32, device=dev)
        invstd = torch.empty(Co, dtype=torch.float32, device=dev)
        ws = workspace(dev)
        rc = lib().m3t_bn_rows_fwd(_p(a), B * T, Co, _p(gamma), _p(beta), _p(run_mean), _p(run_var), float(momentum),
                                   float(eps), int(training), 1, _p(y), _p(mean), _p(invstd), _p(ws), ws.numel() * 4,
                                   _stream())
        _lib.check(rc, "m3t_bn_rows_fwd")
        ctx.save_for_backward(x, w_t, a, y, gamma, mean, invstd)
        ctx.cfg = (pad, int(training), b is not None, prec)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_t, a, y, gamma, mean, invstd = ctx.saved_tensors
        pad, training, has_bias, prec = ctx.cfg
        dy = _req(dy.contiguous(), "dy")
        B, T, Ci = x.shape
        K, Co, _ = w_t.shape
        dev = x.device
        ws = workspace(dev)
        da = torch.empty_like(a)
        dgamma = torch.empty(Co, dtype=torch.float32, device=dev)
        dbeta = torch.empty(Co, dtype=torch.float32, device=dev)
        rc = lib().m3t_bn_rows_bwd(_p(dy), _p(a), _p(y), _p(gamma), _p(mean), _p(invstd), B * T, Co, training, 1, _p(da),
                                   _p(dgamma), _p(dbeta), _p(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "m3t_bn_rows_bwd")
        dx = torch.empty_like(x)
        rc = lib().m3t_conv1d_fwd(_p(da), _p(w_t), None, None, None, _p(dx), None, B, T, Co, Ci, K, 1, pad, 0, 1, 0.0, 0, prec, _stream())
        _lib.check(rc, "m3t_conv1d_fwd (data gradient)")
        dw_t = torch.empty_like(w_t)
        rc = lib().m3t_conv1d_wgrad(_p(da), _p(x), _p(dw_t), B, T, Ci, Co, K, 1, pad, _p(ws), ws.numel() * 4, prec, _stream())
        _lib.check(rc, "m3t_conv1d_wgrad")
        dw = transpose2d(dw_t.view(K, Co * Ci)).view(Co, Ci, K)
        db = None
        if has_bias:
            db = torch.empty(Co, dtype=torch.float32, device=dev)
            colsum(da, 0, B * T, Co, Co, db)
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None, None


def simple_tcn(x_btc, seq):
    """Run the reference's `tcn_simple` nn.Sequential(Conv1d, BatchNorm1d, ReLU, Conv1d, BatchNorm1d, ReLU)
    (models/backbone.py:107-112, 214-222) on channel-last [B,T,C] activations with the HIP kernels; the
    Sequential only holds the parameters/buffers (state_dict contract)."""
    h = x_btc
    for ci in (0, 3):
        conv, bn = seq[ci], seq[ci + 1]
        if conv.stride[0] != 1 or conv.dilation[0] != 1 or conv.groups != 1:
            raise M3THipError("tcn_simple: only stride-1, undilated, ungrouped Conv1d is built")
        training = bn.training or bn.running_mean is None
        if bn.training and bn.track_running_stats:
            count_batch(bn.num_batches_tracked)
        mom = 0.1 if bn.momentum is None else bn.momentum
        h = _ConvBnRelu.apply(h, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                              conv.padding[0], training, mom, bn.eps)
    return h


# ----------------------------------------------------------------------------- CBAM
class _ChannelGate(torch.autograd.Function):
    """models.cbam.ChannelGate (reference models/cbam.py:33-58) on [N,C,H,W]."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x = _req(x.contiguous(), "x")
        for t in (w1, b1, w2, b2):
            _req(t, "cbam parameter")
        N, Cc, H, W = x.shape
        Cr = w1.shape[0]
        dev = x.device
        y = torch.empty_like(x)
        pooled = torch.empty(N, 2, Cc, dtype=torch.float32, device=dev)
        argmax = torch.empty(N, Cc, dtype=torch.int32, device=dev)
        hidden = torch.empty(N, 2, Cr, dtype=torch.float32, device=dev)
        scale = torch.empty(N, Cc, dtype=torch.float32, device=dev)
        rc = lib().m3t_cbam_channel_fwd(_p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(y), _p(pooled),
                                        C.c_void_p(argmax.data_ptr()), _p(hidden), _p(scale), N, Cc, Cr, H * W, _stream())
        _lib.check(rc, "m3t_cbam_channel_fwd")
        ctx.save_for_backward(x, w1, w2, pooled, argmax, hidden, scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, pooled, argmax, hidden, scale = ctx.saved_tensors
        dy = _req(dy.contiguous(), "dy")
        N, Cc, H, W = x.shape
        Cr = w1.shape[0]
        dev = x.device
        dx = torch.empty_like(x)
        dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
        db1 = torch.empty(Cr, dtype=torch.float32, device=dev)
        db2 = torch.empty(Cc, dtype=torch.float32, device=dev)
        ws = workspace(dev, N * (Cc + 3 * Cr) * 4 + (32 << 20))
        rc = lib().m3t_cbam_channel_bwd(_p(dy), _p(x), _p(w1), _p(w2), _p(pooled), C.c_void_p(argmax.data_ptr()),
                                        _p(hidden), _p(scale), _p(dx), _p(dw1), _p(db1), _p(dw2), _p(db2),
                                        N, Cc, Cr, H * W, _p(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "m3t_cbam_channel_bwd")
        return dx, dw1, db1, dw2, db2


class _SpatialGate(torch.autograd.Function):
    """models.cbam.SpatialGate (reference models/cbam.py:74-92): ChannelPool(max,mean) ->
    Conv2d(2,1,5,pad 2) -> BatchNorm2d(1) -> sigmoid -> scale."""

    @staticmethod
    def forward(ctx, x, conv_w, bn_w, bn_b, running_mean, running_var, training, momentum, eps):
        x = _req(x.contiguous(), "x")
        _req(conv_w, "conv weight")
        N, Cc, H, W = x.shape
        dev = x.device
        bn = torch.cat([bn_w.reshape(1), bn_b.reshape(1)]).contiguous()
        running = torch.cat([running_mean.reshape(1), running_var.reshape(1)]).contiguous()
        y = torch.empty_like(x)
        comp = torch.empty(N, 2, H * W, dtype=torch.float32, device=dev)
        cargmax = torch.empty(N, H * W, dtype=torch.int32, device=dev)
        xhat = torch.empty(N, H * W, dtype=torch.float32, device=dev)
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        scale = torch.empty(N, H * W, dtype=torch.float32, device=dev)
        ws = workspace(dev)
        rc = lib().m3t_cbam_spatial_fwd(_p(x), _p(conv_w), _p(bn), _p(running), _p(y), _p(comp),
                                        C.c_void_p(cargmax.data_ptr()), _p(xhat), _p(stats), _p(scale), N, Cc, H, W,
                                        int(training), float(momentum), float(eps), _p(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "m3t_cbam_spatial_fwd")
        if training:
            with torch.no_grad():
                running_mean.copy_(running[0:1])
                running_var.copy_(running[1:2])
        ctx.save_for_backward(x, conv_w, bn, comp, cargmax, xhat, stats, scale)
        ctx.training = bool(training)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, conv_w, bn, comp, cargmax, xhat, stats, scale = ctx.saved_tensors
        dy = _req(dy.contiguous(), "dy")
        N, Cc, H, W = x.shape
        dev = x.device
        dx = torch.empty_like(x)
        dconv = torch.empty_like(conv_w)
        dbn = torch.empty(2, dtype=torch.float32, device=dev)
        ws = workspace(dev, 2 * N * H * W * 4 + (16 << 20))
        rc = lib().m3t_cbam_spatial_bwd(_p(dy), _p(x), _p(conv_w), _p(bn), _p(comp), C.c_void_p(cargmax.data_ptr()),
                                        _p(xhat), _p(stats), _p(scale), _p(dx), _p(dconv), _p(dbn), N, Cc, H, W,
                                        int(ctx.training), _p(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "m3t_cbam_spatial_bwd")
        return dx, dconv, dbn[0:1].clone(), dbn[1:2].clone(), None, None, None, None, None


class _CBAM(torch.autograd.Function):
    """models.cbam.CBAM (reference models/cbam.py:95-111) as ONE operator, csrc/cbam_fused.hip: the intermediate
    x * channel_scale is never written to memory.  BatchNorm parameters and running statistics are handed over as they
    are (one float each): no staging copies."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, conv_w, bn_w, bn_b, running_mean, running_var, training, momentum, eps):
        x = _req(x.contiguous(), "x")
        for t in (w1, b1, w2, b2, conv_w, bn_w, bn_b, running_mean, running_var):
            _req(t, "cbam parameter")
        N, Cc, H, W = x.shape
        Cr, HW = w1.shape[0], H * W
        dev = x.device
        y = torch.empty_like(x)
        # everything backward needs besides x, in ONE allocation: cs [N,C] | pooled [N,2,C] | hidden [N,2,Cr] | comp [N,2,HW] |
        # xhat [N,HW] | ss [N,HW] | stats [4] | argmax_p [N,C] (int32) | cargmax [N,HW] (int32); every part starts 16-B aligned
        offs, tot = _cbam_layout(N, Cc, Cr, HW)
        buf = torch.empty(tot, dtype=torch.float32, device=dev)
        ws = workspace(dev, 16 * N + 256)
        base = buf.data_ptr()
        ptr = lambda i: C.c_void_p(base + 4 * offs[i])
        rc = lib().m3t_cbam_fwd(_p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(conv_w), _p(bn_w), _p(bn_b), _p(running_mean),
                                _p(running_var), _p(y), ptr(0), ptr(7), ptr(1), ptr(2), ptr(3), ptr(8), ptr(4), ptr(5), ptr(6),
                                N, Cc, Cr, H, W, int(training), float(momentum), float(eps), _p(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "m3t_cbam_fwd")
        ctx.save_for_backward(x, w1, w2, conv_w, bn_w, buf)
        # gradient sinks: the seven parameter gradients straight into the flat gradient buffer (order of backward's `sz`)
        ctx.sink_refs = tuple(t if (t is not None and id(t) in _GRAD_SINKS) else None for t in (w1, w2, b1, b2, conv_w, bn_w, bn_b))
        ctx.offs = offs
        ctx.training = bool(training)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, conv_w, bn_w, buf = ctx.saved_tensors
        offs = ctx.offs
        dy = _req(dy.contiguous(), "dy")
        N, Cc, H, W = x.shape
        Cr = w1.shape[0]
        dev = x.device
        dx = torch.empty_like(x)
        # parameter gradients in one allocation too, packed (the kernels write them as scalars): dw1 [Cr,C] | dw2 [C,Cr] | db1 [Cr] |
        # db2 [C] | dconv [50] | dbn_w [1] | dbn_b [1]; ONE split hands back the seven gradients (the host path of this operator is
        # as long as its kernels at the late ResNet stages: tools/cbam_host.py)
        sz = (Cr * Cc, Cc * Cr, Cr, Cc, 50, 1, 1)
        go = (0, sz[0], 2 * sz[0], 2 * sz[0] + Cr, 2 * sz[0] + Cr + Cc, 2 * sz[0] + Cr + Cc + 50, 2 * sz[0] + Cr + Cc + 51)
        gbuf = torch.empty(go[6] + 1, dtype=torch.float32, device=dev)
        ws = workspace(dev, _cbam_ws_bytes(N, Cc, Cr, H, W))
        base, gbase = buf.data_ptr(), gbuf.data_ptr()
        ptr = lambda i: C.c_void_p(base + 4 * offs[i])
        # input index of (w1, w2, b1, b2, conv_w, bn_w, bn_b) in forward's argument list: 1, 3, 2, 4, 5, 6, 7
        need = (1, 3, 2, 4, 5, 6, 7)
        sinks = [(_take_sink(r) if (r is not None and ctx.needs_input_grad[need[i]]) else None) for i, r in enumerate(ctx.sink_refs)]
        gp = lambda i: C.c_void_p(sinks[i].data_ptr() if sinks[i] is not None else gbase + 4 * go[i])
        rc = lib().m3t_cbam_bwd(_p(dy), _p(x), _p(w1), _p(w2), _p(conv_w), _p(bn_w), ptr(0), ptr(7), ptr(1), ptr(2), ptr(3), ptr(8),
                                ptr(4), ptr(5), ptr(6), _p(dx), gp(0), gp(2), gp(1), gp(3), gp(4), gp(5), gp(6),
                                N, Cc, Cr, H, W, int(ctx.training), _p(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "m3t_cbam_bwd")
        g = gbuf.split_with_sizes(sz)
        k = lambda i, t: None if sinks[i] is not None else t
        return (dx, k(0, g[0].view(w1.shape)), k(2, g[2]), k(1, g[1].view(w2.shape)), k(3, g[3]), k(4, g[4].view(conv_w.shape)), k(5, g[5]),
                k(6, g[6]), None, None, None, None, None)


_CBAM_LAYOUTS, _CBAM_WS = {}, {}


def _cbam_layout(N, Cc, Cr, HW):
    """offsets (floats) of the saved slabs inside _CBAM's one allocation, every part 16-B aligned; cached per shape"""
    key = (N, Cc, Cr, HW)
    hit = _CBAM_LAYOUTS.get(key)
    if hit is None:
        offs, tot = [], 0
        for n_ in (N * Cc, 2 * N * Cc, 2 * N * Cr, 2 * N * HW, N * HW, N * HW, 4, N * Cc, N * HW):
            offs.append(tot)
            tot += (n_ + 3) // 4 * 4
        hit = _CBAM_LAYOUTS[key] = (tuple(offs), tot)
    return hit


def _cbam_ws_bytes(N, Cc, Cr, H, W):
    key = (N, Cc, Cr, H, W)
    hit = _CBAM_WS.get(key)
    if hit is None:
        hit = _CBAM_WS[key] = int(lib().m3t_cbam_fused_ws_bytes(N, Cc, Cr, H, W))
    return hit


CBAM_FUSED = [os.environ.get("M3T_CBAM_FUSED", "1") != "0"]      # 0: the two gates as two operators (cbam.hip), for A/B runs


def cbam_fused_ok(x, Cr):
    """the fused CBAM operator covers this input (float32 device tensor [N,C,H,W] with H*W/4 <= 512 units, or H*W <= 512 odd pixels)"""
    return (CBAM_FUSED[0] and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and bool(lib().m3t_cbam_fused_ok(x.shape[1], Cr, x.shape[2], x.shape[3])))


def cbam(x, w1, b1, w2, b2, conv_w, bn_w, bn_b, running_mean, running_var, training, momentum=0.01, eps=1e-5):
    return _CBAM.apply(x, w1, b1, w2, b2, conv_w, bn_w, bn_b, running_mean, running_var, training, momentum, eps)


def channel_gate(x, w1, b1, w2, b2):
    return _ChannelGate.apply(x, w1, b1, w2, b2)


def spatial_gate(x, conv_w, bn_w, bn_b, running_mean, running_var, training, momentum=0.01, eps=1e-5):
    return _SpatialGate.apply(x, conv_w, bn_w, bn_b, running_mean, running_var, training, momentum, eps)


# ----------------------------------------------------------------------------- DDP helper
def grad_norm_scale_(flat, world_size, max_norm):
    """In place: flat <- flat/world * min(1, max_norm/(||flat/world|| + 1e-6)).  Returns the norm (device scalar)."""
    _req(flat, "flat gradient buffer")
    norm = torch.empty(1, dtype=torch.float32, device=flat.device)
    ws = workspace(flat.device)
    rc = lib().m3t_grad_norm_scale(_p(flat), flat.numel(), 1.0 / float(world_size), float(max_norm), _p(norm), _p(ws),
                                   ws.numel() * 4, _stream())
    _lib.check(rc, "m3t_grad_norm_scale")
    return norm


# ----------------------------------------------------------------------------- BatchNorm3d (+ReLU) of the 3-D stems
class _BNPlanes(torch.autograd.Function):
    """nn.BatchNorm3d (+ nn.ReLU) on channels-first activations [N, C, T, H, W] (reference models/backbone.py:73-103,179-191), csrc/bn.hip
    m3t_bn_planes_*: statistics sweep | apply + ReLU; backward sums sweep | dx -- the ReLU costs no pass of its own"""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, training, momentum, eps, relu):
        x = _req(x.contiguous(), "x")
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        y = torch.empty_like(x)
        stats = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
        ws = workspace(x.device, int(lib().m3t_bn_planes_ws_bytes(N, Cc, S)))
        slot = amax_slots(1, x.device) if TRANSPOSE_IMAGES[0] else None      # (raised to max |y| by the apply kernel: the next convolution's scale)
        if slot is not None:
            amax_out(slot.data_ptr())
        rc = lib().m3t_bn_planes_fwd(_p(x), N, Cc, S, _p(gamma), _p(beta), _p(run_mean), _p(run_var), float(momentum), float(eps),
                                     int(training), int(relu), _p(y), _p(stats[0]), _p(stats[1]), _p(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "m3t_bn_planes_fwd")
        _OUT_SLOT[0] = slot
        ctx.save_for_backward(x, y if relu else None, gamma, stats)
        ctx.training, ctx.relu = bool(training), bool(relu)
        # gradient sinks (top of this file): dgamma / dbeta are written by the backward kernel straight into the flat gradient buffer
        ctx.sink_refs = (gamma if (gamma is not None and id(gamma) in _GRAD_SINKS) else None,
                         beta if (beta is not None and id(beta) in _GRAD_SINKS) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, stats = ctx.saved_tensors
        dy = _req(dy.contiguous(), "dy")
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        dx = torch.empty_like(x)
        g = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
        ws = workspace(x.device, int(lib().m3t_bn_planes_ws_bytes(N, Cc, S)))
        gs = _take_sink(ctx.sink_refs[0]) if (ctx.needs_input_grad[1] and ctx.sink_refs[0] is not None) else None
        bs = _take_sink(ctx.sink_refs[1]) if (ctx.needs_input_grad[2] and ctx.sink_refs[1] is not None) else None
        slot = amax_slots(1, x.device) if TRANSPOSE_IMAGES[0] else None      # (raised to max |dx|: handed to the convolution in front, _note_grad_slot)
        if slot is not None:
            amax_out(slot.data_ptr())
        rc = lib().m3t_bn_planes_bwd(_p(dy), _p(x), _p(y), _p(gamma), _p(stats[0]), _p(stats[1]), N, Cc, S, int(ctx.training), int(ctx.relu),
                                     _p(dx), _p(gs if gs is not None else g[0]), _p(bs if bs is not None else g[1]), _p(ws), ws.numel() * 4,
                                     _stream())
        _lib.check(rc, "m3t_bn_planes_bwd")
        if slot is not None:
            _note_grad_slot(dx, slot)
        return (dx, (g[0] if (gamma is not None and gs is None) else None), (g[1] if (gamma is not None and bs is None) else None),
                None, None, None, None, None, None)


# M3T_BN_PLANES=0: BatchNorm3d / BatchNorm2d (+ReLU) of the stems and the per-frame ResNet on the stock (MIOpen) ops instead of csrc/bn.hip's
# channel-plane kernels (A/B runs; README switch table)
BN_PLANES = [os.environ.get("M3T_BN_PLANES", "1") != "0"]


def bn_planes(x, gamma, beta, run_mean, run_var, training, momentum, eps, relu=True):
    _OUT_SLOT[0] = None
    return _tag_out(_BNPlanes.apply(x, gamma, beta, run_mean, run_var, training, momentum, eps, relu))


class _PoolPlanes(torch.autograd.Function):
    """max pooling of [..., H, W] planes with a (kh, kw) window (csrc/bn.hip m3t_pool_planes_*: one byte per output for the winner,
    gather backward) -- nn.MaxPool3d((1, k, k)) of the 3-D stems (reference models/backbone.py:80,86,92,182)"""

    @staticmethod
    def forward(ctx, x, k, s, p):
        x = _req(x.contiguous(), "x")
        H, W = x.shape[-2], x.shape[-1]
        P = x.numel() // (H * W)
        Ho, Wo = (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1
        y = torch.empty(x.shape[:-2] + (Ho, Wo), dtype=torch.float32, device=x.device)
        win = torch.empty(x.shape[:-2] + (Ho, Wo), dtype=torch.uint8, device=x.device)
        rc = lib().m3t_pool_planes_fwd(_p(x), P, H, W, k[0], k[1], s[0], s[1], p[0], p[1], _p(y), C.c_void_p(win.data_ptr()), _stream())
        _lib.check(rc, "m3t_pool_planes_fwd")
        ctx.save_for_backward(win)
        ctx.geo = (tuple(x.shape), k, s, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (win,) = ctx.saved_tensors
        shape, k, s, p = ctx.geo
        dy = _req(dy.contiguous(), "dy")
        H, W = shape[-2], shape[-1]
        dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
        rc = lib().m3t_pool_planes_bwd(_p(dy), C.c_void_p(win.data_ptr()), dx.numel() // (H * W), H, W, k[0], k[1], s[0], s[1], p[0], p[1],
                                       _p(dx), _stream())
        _lib.check(rc, "m3t_pool_planes_bwd")
        return dx, None, None, None


def pool_planes(x, k, s, p):
    return _PoolPlanes.apply(x, tuple(k), tuple(s), tuple(p))


# ----------------------------------------------------------------------------- Conv3d of the visual stems
CONV3D_GEMM = [os.environ.get("M3T_CONV3D_MIOPEN", "0") != "1"]      # False: forward on MIOpen as until round 4 (tested fallback)
CONV3D_PRESPLIT = [os.environ.get("M3T_CONV3D_PRESPLIT", "1") != "0"]      # the tap walk on operands split once (m3t_f16x3_split) instead of in the kernel
CONV3D_TAPS = [os.environ.get("M3T_CONV3D_MIOPEN", "0") not in ("1", "dgrad")]      # False: the data gradient on MIOpen ("dgrad": only that)
# forward and weight gradient as tap walks over channels-last activations (no patch matrix at all) where the layer allows it; "0": the
# patch-matrix GEMMs of the first half of round 5 (kept: the layers with C_in % 32 != 0 -- the stems' first convolutions -- always take them)
CONV3D_IMPLICIT = [os.environ.get("M3T_CONV3D_IMPLICIT", "1") != "0"]
CONV3D_CALLS = {"walk": 0, "patch": 0, "torch": 0}       # forward calls by path (tests assert the path they mean to check)
_STOCK_WARNED = set()
def _dgrad_plan(ks, st, pd, dims):
    """the data gradient of a convolution as stride-1 tap walks, one per PARITY CLASS of the input grid (one class for a stride-1 layer): input
    position s h' + c receives only the taps s j + r, r = (c + p) mod s, from source row h' + (c + p - r) / s - j.  -> [(class, r, taps per
    axis, the class's sub-grid, base)] for the classes that have a tap and a position"""
    plan = []
    for ct in range(st[0]):
        for ch in range(st[1]):
            for cw_ in range(st[2]):
                cls = (ct, ch, cw_)
                r = [(cls[a] + pd[a]) % st[a] for a in range(3)]
                sub = [len(range(r[a], ks[a], st[a])) for a in range(3)]               # taps of this class per axis
                size = [len(range(cls[a], dims[a], st[a])) for a in range(3)]          # the class's sub-grid
                if min(sub) < 1 or min(size) < 1:
                    continue
                plan.append((cls, r, sub, size, [(cls[a] + pd[a] - r[a]) // st[a] for a in range(3)]))
    return plan


def _dgrad_weight_images(wd, plan, st, a_w):
    """the m3t_f16x3_split images of the K-contiguous [ci][(tap, co)] (sub-)kernels the walks of _dgrad_plan read, under the weights' slot a_w"""
    Co, Ci = wd.shape[0], wd.shape[1]
    imgs = []
    for cls, r, sub, size, base in plan:
        taps = sub[0] * sub[1] * sub[2]
        w_img = torch.empty(Ci, taps * Co, dtype=torch.float32, device=wd.device)
        if tuple(st) == (1, 1, 1) and wd.is_contiguous():       # straight from w (no permute copy)
            _lib.check(lib().m3t_f16x3_split_perm(_p(wd), Ci, taps, Co, taps, 1, Ci * taps, _p(w_img), a_w, _stream()), "m3t_f16x3_split_perm")
        else:
            w_t = wd[:, :, r[0]::st[0], r[1]::st[1], r[2]::st[2]].permute(1, 2, 3, 4, 0).contiguous().view(Ci, taps * Co)
            _lib.check(lib().m3t_f16x3_split(_p(w_t), Ci, taps * Co, taps * Co, _p(w_img), taps * Co, a_w, _stream()), "m3t_f16x3_split")
        imgs.append(w_img)
    return imgs


def _dgrad_images(ctx, w, st, pd, dims, a_w):
    """backward: (plan, weight images) of a convolution's data gradient.  (Round 6 tried to queue the images in the FORWARD pass on a
    weight-gradient stream -- they depend on the weights only: ResNet3D+CBAM 18.64 -> 18.89 ms, C5 unchanged; the stream hops cost more than
    the ~50 small launches they take off the backward chain.)"""
    plan = _dgrad_plan(tuple(w.shape[2:]), st, pd, dims)
    return plan, _dgrad_weight_images(w.detach(), plan, st, a_w)


# M3T_CONV_WGRAD_STREAM=0: the convolutions' weight-gradient walks stay on the stream their layer's backward runs on (as until round 6)
CONV_WGRAD_STREAM = [os.environ.get("M3T_CONV_WGRAD_STREAM", "1") != "0"]
_CONV_RR = [0]


def _conv_wgrad_stream(device, w_sink):
    """the weight-gradient stream for a convolution's weight-gradient walk, already waiting for the current stream -- or None (no sink: the
    gradient tensor is returned to autograd on this stream; CPU; switched off)"""
    if w_sink is None or not CONV_WGRAD_STREAM[0] or device.type != "cuda":
        return None
    wg = wgrad_stream(device, _CONV_RR[0])
    _CONV_RR[0] += 1
    wg.wait_stream(cur_stream(device))
    return wg


# M3T_TRANSPOSE_IMAGES=0: a planes-path convolution whose input (or output gradient) comes with its producer's magnitude slot -- BatchNorm's apply
# and dx kernels, the residual add + ReLU -- still transposes to fp32 rows and splits them in a second pass (as until round 6) instead of
# writing the channels-last image in one (m3t_bct_to_btc_img)
TRANSPOSE_IMAGES = [os.environ.get("M3T_TRANSPOSE_IMAGES", "1") != "0"]
_X_SLOT = [None]             # conv2d() / conv3d() -> _Conv3dGemmWgrad.forward: the slot tagged on the caller's x (autograd hands forward() a detached alias)
_OUT_SLOT = [None]           # _BNPlanes.forward / _AddRelu.forward -> their wrappers: the slot their kernel raised for the output


def _tag_out(y):
    """attach the slot the producing kernel raised to its output tensor (holds for this version of y: _tagged_amax)"""
    if _OUT_SLOT[0] is not None:
        y._m3t_amax = (_OUT_SLOT[0], y._version)
        _OUT_SLOT[0] = None
    return y


# M3T_WGRAD_IMAGES=0: the convolutions' weight-gradient walk splits its fp32 operands in its loop (as until round 6) instead of reading the images
# the forward walk (x) and the data gradient (dy) have made
WGRAD_IMAGES = [os.environ.get("M3T_WGRAD_IMAGES", "1") != "0"]
STOCK_FALLBACKS = {}          # site -> number of calls that took a stock (torch / MIOpen) operator instead of the HIP library


def stock_fallback(site, why):
    """A product module is about to run a stock torch operator instead of the HIP library (a shape no kernel covers, an A/B switch): say so
    ONCE per site on stderr and count it (which kernel ran must not be a silent function of shape or mode)."""
    STOCK_FALLBACKS[site] = STOCK_FALLBACKS.get(site, 0) + 1
    if site not in _STOCK_WARNED:
        _STOCK_WARNED.add(site)
        print("m3t: %s runs on the stock torch operator (%s)" % (site, why), file=sys.stderr, flush=True)



def _conv3d_plan(x, w, stride, padding):
    """(To, Ho, Wo, rows, Kc, Kp) of the patch-matrix form, or None when the 16-bit-term GEMM cannot tile it"""
    N_, Ci, T_, H_, W_ = x.shape
    Co, _, kt, kh, kw = w.shape
    To = (T_ + 2 * padding[0] - kt) // stride[0] + 1
    Ho = (H_ + 2 * padding[1] - kh) // stride[1] + 1
    Wo = (W_ + 2 * padding[2] - kw) // stride[2] + 1
    rows, Kc = N_ * To * Ho * Wo, Ci * kt * kh * kw
    if min(To, Ho, Wo) < 1 or Co % 64 != 0:
        return None
    # round 6: the tap walks take any number of rows (a ragged last tile reads zeros and is not stored); only the patch-matrix GEMMs
    # (M3T_CONV3D_IMPLICIT=0, C_in % 32 != 0 layers) still need whole 128-row tiles
    cw = Ci if Ci % 32 == 0 else (4 if (Ci <= 4 and kw <= 8) else 0)
    walk = bool(CONV3D_IMPLICIT[0] and (_PREC[0] & _lib.M3T_GEMM_F16X3) and cw)
    if rows % 128 != 0 and not walk:
        return None
    Kp = Kc if Kc % 64 == 0 else (Kc + 127) // 128 * 128       # (the weight gradient's rules: dW = dy^T P, or transposed and padded)
    return To, Ho, Wo, rows, Kc, Kp


class _Conv3dGemmWgrad(torch.autograd.Function):
    """Conv3d of the visual stems and, with a unit time axis, the per-frame ResNet's Conv2d (reference models/backbone.py:73-103,179-271,
    327-332, models/resnet.py:40-45).  Round 5: every pass is a tap-walk implicit GEMM on the fp16x3 kernels over channels-last rows --
    forward (any stride; x channels-last KEPT for backward), weight gradient (the walk turned round, reduction over dy's rows), data gradient
    (dy channels-last, flipped taps; round 6: the strided layers' as one walk per output parity class over the taps w[..., r::s]); results
    leave the walks as channel planes.  No patch matrix (MIOpen's fp32
    forward ran at 26 TFLOP/s, 12.4 of the 30.4 ms of a C5 step in round 4; the patch-matrix GEMMs of the first half of round 5 wrote 4-6 GB
    per step).  C_in % 32 != 0 layers other than the 3-channel first layers, M3T_CONV3D_IMPLICIT=0: patch matrix (m3t_im2col3d) x W^T and
    dy^T P on the GEMM; any number of rows (ragged last tile, round 6); M3T_CONV3D_MIOPEN=1 / shapes no walk covers: torch / MIOpen, announced
    once per site by stock_fallback()."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, as2d=False):
        # as2d (models.resnet.GemmConv2d): x [N, C, H, W] and the Conv2d PARAMETER w [Co, Ci, kh, kw] take a unit time axis HERE, so that the
        # gradient-sink and per-step magnitude lookups below are keyed on the Parameter itself (a .unsqueeze(2) view made by the
        # caller is a fresh tensor that matches neither table)
        ctx.as2d = bool(as2d)
        w_par = w
        if as2d:
            x, w = x.unsqueeze(2), w.unsqueeze(2)
            stride, padding = (1,) + tuple(stride), (0,) + tuple(padding)
        y = _Conv3dGemmWgrad._forward5(ctx, x, w, w_par, b, stride, padding)
        return y.squeeze(2) if as2d else y

    @staticmethod
    def _forward5(ctx, x, w, w_par, b, stride, padding):
        ctx.stride, ctx.padding, ctx.has_bias = stride, padding, b is not None
        ctx.sink_refs = (w_par if id(w_par) in _GRAD_SINKS else None, b if (b is not None and id(b) in _GRAD_SINKS) else None)      # gradient sinks
        plan = _conv3d_plan(x, w, stride, padding) if (CONV3D_GEMM[0] and x.is_cuda and (_PREC[0] & _lib.M3T_GEMM_F16X3 or _PREC[0] == 0)) else None
        ctx.prec = _PREC[0]
        ctx.impl = 0                     # channel width of the channels-last input kept by a tap-walk forward (0: none)
        if plan is None:
            stock_fallback("conv%dd %s k%s s%s" % (2 if ctx.as2d else 3, tuple(x.shape[1:]), tuple(w.shape), tuple(stride)),
                           "M3T_CONV3D_MIOPEN=1" if not CONV3D_GEMM[0] else "rows % 128 or C_out % 64: no GEMM tile fits, or a precision mode without the walk")
            y = torch.conv3d(x, w, b, stride, padding)
            CONV3D_CALLS["torch"] += 1
            ctx.save_for_backward(x, w)
            ctx.pat = None
            return y
        To, Ho, Wo, rows, Kc, Kp = plan
        N_, Ci, T_, H_, W_ = x.shape
        Co, _, kt, kh, kw = w.shape
        xc = _req(x.contiguous(), "x")
        slots = amax_slots(2, x.device)
        cw = Ci if Ci % 32 == 0 else (4 if (Ci <= 4 and kw <= 8) else 0)        # channel width of the channels-last input the walk reads
        if CONV3D_IMPLICIT[0] and (_PREC[0] & _lib.M3T_GEMM_F16X3) and cw:
            # second half of round 5: no patch matrix.  x channels-last (one tiled transpose that raises x's magnitude slot; KEPT for the
            # weight gradient: 1 x the activations instead of 9 - 27 x), both operands split once, the tap walk with the layer's stride,
            # bias in its epilogue, one transpose back.  The stems' first layers (3 input channels): channels padded to four, the kernel's
            # width to eight taps -- one 32-deep k tile per (kt, kh) pair (m3t_conv3d_fwd_taps4)
            ctx.w_keep = []
            a_w = weight_amax(w_par, ctx.w_keep)
            taps = kt * kh * kw
            w_native = None
            if cw == Ci and w.is_contiguous() and w.data_ptr() % 16 == 0:
                w_native = w.detach()                 # round 6: the [co][(tap, ci)] image straight from w (m3t_f16x3_split_perm: no permute copy)
                w_t = w_native.view(Co, Ci * taps)    # (only measured below: max |w| is the same for every permutation)
            elif cw == Ci:
                w_t = _req(w.detach().permute(0, 2, 3, 4, 1).contiguous(), "weight").view(Co, taps * Ci)       # [co][(tap, ci)]
            else:
                w8 = torch.zeros(Co, kt, kh, 8, 4, dtype=torch.float32, device=x.device)
                w8[:, :, :, :kw, :Ci].copy_(w.detach().permute(0, 2, 3, 4, 1))
                w_t = w8.view(Co, kt * kh * 32)
            if a_w is None:
                a_w = slots.data_ptr() + 8
                if not measure_amax([(w_t, a_w)]):
                    a_w = None
            if a_w is not None:
                srows, wk = N_ * T_ * H_ * W_, w_t.shape[1]
                x_slot, _X_SLOT[0] = _X_SLOT[0], None
                # round 6: x comes with the magnitude slot its producer raised (BatchNorm + ReLU, the residual add + ReLU) and nothing reads the
                # fp32 rows any more (the weight gradient's walk takes the image): planes -> image in ONE pass (was: transpose + measure, split)
                direct_img = bool(TRANSPOSE_IMAGES[0] and WGRAD_IMAGES[0] and x_slot is not None and cw == Ci and xc.data_ptr() == x.data_ptr())
                a_x = x_slot.data_ptr() if direct_img else slots.data_ptr()
                x_img = torch.empty(srows, cw, dtype=torch.float32, device=x.device)
                w_img = torch.empty_like(w_t)
                y_cl = torch.empty(rows, Co, dtype=torch.float32, device=x.device)
                y = torch.empty(N_, Co, To, Ho, Wo, dtype=torch.float32, device=x.device)
                wsd = workspace(x.device)
                if direct_img:
                    x_cl = None
                    ctx.w_keep.append(x_slot)
                    _lib.check(lib().m3t_bct_to_btc_img(_p(xc), _p(x_img), N_, Ci, T_ * H_ * W_, a_x, None, _stream()), "m3t_bct_to_btc_img")
                else:
                    x_cl = torch.empty(srows, cw, dtype=torch.float32, device=x.device)
                    amax_out(slots.data_ptr())
                    if cw == Ci:
                        _lib.check(lib().m3t_bct_to_btc(_p(xc), _p(x_cl), N_, Ci, T_ * H_ * W_, _stream()), "m3t_bct_to_btc")
                    else:
                        _lib.check(lib().m3t_planes_to_cl4(_p(xc), _p(x_cl), N_, Ci, T_ * H_ * W_, _stream()), "m3t_planes_to_cl4")
                    _lib.check(lib().m3t_f16x3_split(_p(x_cl), srows, cw, cw, _p(x_img), cw, slots.data_ptr(), _stream()), "m3t_f16x3_split")
                if w_native is not None:
                    _lib.check(lib().m3t_f16x3_split_perm(_p(w_native), Co, taps, Ci, Ci * taps, 1, taps, _p(w_img), a_w, _stream()), "m3t_f16x3_split_perm")
                else:
                    _lib.check(lib().m3t_f16x3_split(_p(w_t), Co, wk, wk, _p(w_img), wk, a_w, _stream()), "m3t_f16x3_split")
                geo = (kt, kh, kw, stride[0], stride[1], stride[2], padding[0], padding[1], padding[2], a_x, a_w, _p(wsd),
                       wsd.numel() * 4, _p(y), _stream())          # (y: planes, written by the walk's epilogue in one K pass)
                bp = _p(b) if b is not None else None
                if cw == Ci:
                    _lib.check(lib().m3t_conv3d_fwd_taps(_p(x_img), _p(w_img), bp, _p(y_cl), N_, Ci, Co, T_, H_, W_, *geo), "m3t_conv3d_fwd_taps")
                else:
                    _lib.check(lib().m3t_conv3d_fwd_taps4(_p(x_img), _p(w_img), bp, _p(y_cl), N_, Co, T_, H_, W_, *geo), "m3t_conv3d_fwd_taps4")
                # round 6: the weight gradient's walk reads the IMAGE of x (same bytes as x_cl, made above for the forward walk) and the image of
                # dy its data gradient makes anyway: no conversions in its loop (WGRAD_IMAGES; a first layer's four-channel rows stay fp32)
                ctx.x_is_img = bool(WGRAD_IMAGES[0] and cw == Ci)
                ctx.a_x = a_x
                ctx.save_for_backward(x, w, x_img if ctx.x_is_img else x_cl, slots)
                ctx.pat = (rows, Kc, Kp)
                ctx.a_w = a_w
                ctx.impl = cw
                CONV3D_CALLS["walk"] += 1
                return y
        CONV3D_CALLS["patch"] += 1
        pat = torch.empty(rows, Kp, dtype=torch.float32, device=x.device)
        _lib.check(lib().m3t_im2col3d(_p(xc), N_, Ci, T_, H_, W_, kt, kh, kw, stride[0], stride[1], stride[2], padding[0], padding[1], padding[2],
                                      _p(pat), rows, Kp, C.c_void_p(slots.data_ptr()), _stream()), "m3t_im2col3d")
        w2 = _req(w.contiguous(), "weight").view(Co, Kc)
        if Kp != Kc:                                        # (the first layer: C_in k^3 = 81 -> 128 zero-padded columns on both operands)
            wp = torch.zeros(Co, Kp, dtype=torch.float32, device=x.device)
            wp[:, :Kc].copy_(w2)
            w2 = wp
        ctx.w_keep = []
        a_w = weight_amax(w_par, ctx.w_keep)
        if a_w is None:
            a_w = slots.data_ptr() + 8
            if not measure_amax([(w2, a_w)]):
                a_w = None
        y_cl = torch.empty(rows, Co, dtype=torch.float32, device=x.device)
        sgemm(0, 1, rows, Co, Kp, pat, 0, Kp, w2, 0, Kp, y_cl, 0, Co, bias=b, amax=(slots.data_ptr(), a_w))
        y = torch.empty(N_, Co, To, Ho, Wo, dtype=torch.float32, device=x.device)
        _lib.check(lib().m3t_btc_to_bct(_p(y_cl), _p(y), N_, To * Ho * Wo, Co, _stream()), "m3t_btc_to_bct")
        ctx.save_for_backward(x, w, pat, slots)
        ctx.pat = (rows, Kc, Kp)
        ctx.a_w = a_w                 # max |W|: the same for every permutation of the weights (the data gradient's [(tap, co)][ci] matrix)
        return y

    @staticmethod
    def backward(ctx, dy):
        # the slot the producer of dy raised (BatchNorm's dx kernel: _note_grad_slot) -- looked up on the object autograd handed over, before a view of it is made
        dy_slot = _grad_slot(dy)[0] if (TRANSPOSE_IMAGES[0] and dy.is_contiguous()) else None
        if ctx.as2d:
            dx, dw, db = _Conv3dGemmWgrad._backward5(ctx, dy.unsqueeze(2), dy_slot)
            return (dx.squeeze(2) if dx is not None else None, dw.squeeze(2) if dw is not None else None, db, None, None, None)
        dx, dw, db = _Conv3dGemmWgrad._backward5(ctx, dy, dy_slot)
        return dx, dw, db, None, None, None

    @staticmethod
    def _backward5(ctx, dy, dy_slot=None):
        saved = ctx.saved_tensors
        x, w = saved[0], saved[1]
        impl = ctx.impl if ctx.pat is not None else 0                       # forward was a tap walk: saved[2] is x channels-last, no patch matrix
        kept = (saved[2], saved[3]) if (ctx.pat is not None and not impl) else None      # the forward's patch matrix and its magnitude slot
        st, pd = ctx.stride, ctx.padding
        dx = dw = db = None
        Co, Ci, kt, kh, kw = w.shape
        N_, _, T_, H_, W_ = x.shape
        # weight / bias gradients straight into the flat gradient buffer where FlatGradDDP registered sinks (no AccumulateGrad add kernel)
        w_sink = _take_sink(ctx.sink_refs[0]) if (ctx.needs_input_grad[1] and ctx.sink_refs[0] is not None) else None
        b_sink = _take_sink(ctx.sink_refs[1]) if (ctx.has_bias and ctx.needs_input_grad[2] and ctx.sink_refs[1] is not None) else None
        # round 5: the data gradient of a stride-1 layer as a tap-walk contraction over dy channels-last (m3t_conv3d_taps: an implicit GEMM, no
        # patch matrix, no col2im) -- MIOpen's data gradient (Col2Im3dU + Tensile GEMMs) was ~5 ms of a C5 step
        taps_dx = (ctx.needs_input_grad[0] and CONV3D_TAPS[0] and ctx.pat is not None and tuple(st) == (1, 1, 1) and Co % 32 == 0 and Ci % 64 == 0
                   and ((N_ * T_ * H_ * W_) % 128 == 0 or bool(ctx.prec & _lib.M3T_GEMM_F16X3)) and (ctx.prec & _lib.M3T_GEMM_F16X3 or ctx.prec == 0))
        # round 6: the data gradient of a STRIDED layer (the six stride-2 layers of the per-frame ResNet-18: reference models/resnet.py:24,67-69,
        # 95-105) as tap walks too -- one per parity class of the input grid.  Input position h = s h' + c receives only the taps k = s j + r with
        # r = (c + p) mod s, from source row h' + (c + p - r) / s - j: per class a stride-1 walk over the class's sub-grid with the sub-kernel
        # w[..., r::s] (flipped taps, the same kernel as the stride-1 layers), exactly the forward pass's multiply-adds.  MIOpen's igemm_bwd was
        # the last library kernel on the training path.
        strided_dx = (ctx.needs_input_grad[0] and not taps_dx and CONV3D_TAPS[0] and CONV3D_PRESPLIT[0] and ctx.pat is not None
                      and tuple(st) != (1, 1, 1) and Co % 32 == 0 and Ci % 64 == 0 and bool(ctx.prec & _lib.M3T_GEMM_F16X3) and ctx.a_w is not None)
        if ctx.needs_input_grad[0] and not taps_dx and not strided_dx:
            stock_fallback("conv%dd data gradient k%s s%s" % (2 if ctx.as2d else 3, tuple(w.shape), tuple(st)),
                           "no tap walk for this layer (C_out % 32, C_in % 64, a precision mode without it, or M3T_CONV3D_MIOPEN)")
            if x.shape[2] == 1 and w.shape[2] == 1 and st[0] == 1 and pd[0] == 0:
                # a 2-D convolution with a unit time axis (models.resnet.GemmConv2d): MIOpen's 2-D data gradient, not its 3-D one
                dx = torch.ops.aten.convolution_backward(dy.squeeze(2), x.squeeze(2), w.squeeze(2), None, list(st[1:]), list(pd[1:]), [1, 1],
                                                         False, [0, 0], 1, [True, False, False])[0].unsqueeze(2)
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, w, None, list(st), list(pd), [1, 1, 1], False, [0, 0, 0], 1,
                                                         [True, False, False])[0]
        slot_dy = pre_img = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]) or taps_dx or strided_dx:
            # dy channels-last [N,T',H',W',Co] = [rows, Co] for the GEMM: the library's tiled transpose (both sides coalesced), which
            # raises dy's magnitude slot on the way -- was torch's strided permute copy plus a measuring launch (1.1 ms of the C5 step)
            dyc = _req(dy.contiguous(), "dy")
            rows = dyc.numel() // Co
            Nn, Sp = dyc.shape[0], rows // dyc.shape[0]
            with_sums = ctx.has_bias and ctx.needs_input_grad[2]
            if with_sums:
                dpart = torch.empty(Nn * ((Sp + 31) // 32), Co, dtype=torch.float32, device=dy.device)
            # round 6: dy comes with the slot its producer raised and every consumer below reads the IMAGE (the pre-split data-gradient walks, the
            # weight gradient's walk on images): planes -> image in one pass, no fp32 rows (was: transpose + measure, then split)
            imgs_only = bool(dy_slot is not None and Co % 4 == 0 and dyc.data_ptr() == dy.data_ptr() and bool(ctx.prec & _lib.M3T_GEMM_F16X3)
                             and CONV3D_PRESPLIT[0] and getattr(ctx, "a_w", None) is not None and (strided_dx or taps_dx or not ctx.needs_input_grad[0])
                             and (not ctx.needs_input_grad[1] or (impl and getattr(ctx, "x_is_img", False))))
            if imgs_only:
                dy_cl, slot_dy = None, dy_slot
                pre_img = torch.empty(rows, Co, dtype=torch.float32, device=dy.device)
                _lib.check(lib().m3t_bct_to_btc_img(_p(dyc), _p(pre_img), Nn, Co, Sp, slot_dy.data_ptr(), _p(dpart) if with_sums else None, _stream()),
                           "m3t_bct_to_btc_img")
            else:
                dy_cl = torch.empty(rows, Co, dtype=torch.float32, device=dy.device)
                slot_dy = amax_slots(1, dy.device)
                amax_out(slot_dy.data_ptr())                      # (armed after every allocation, consumed by the very next call)
                if with_sums:      # the bias gradient's per-tile channel sums ride along
                    _lib.check(lib().m3t_bct_to_btc_sums(_p(dyc), _p(dy_cl), Nn, Co, Sp, _p(dpart), _stream()), "m3t_bct_to_btc_sums")
                else:
                    _lib.check(lib().m3t_bct_to_btc(_p(dyc), _p(dy_cl), Nn, Co, Sp, _stream()), "m3t_bct_to_btc")
        dy_img_ = [pre_img]

        def dy_image():      # dy channels-last split ONCE under the slot its transpose raised: read by the data gradient's walk(s) and the weight gradient's
            if dy_img_[0] is None:
                dy_img_[0] = torch.empty_like(dy_cl)
                _lib.check(lib().m3t_f16x3_split(_p(dy_cl), rows, Co, Co, _p(dy_img_[0]), Co, slot_dy.data_ptr(), _stream()), "m3t_f16x3_split")      # (never with imgs_only: the image is there)
            return dy_img_[0]
        if taps_dx:
            To, Ho, Wo = dy.shape[2], dy.shape[3], dy.shape[4]
            dx_cl = torch.empty(N_ * T_ * H_ * W_, Ci, dtype=torch.float32, device=dy.device)      # (scratch of the split-K layers; dx itself: planes)
            dx = torch.empty(N_, Ci, T_, H_, W_, dtype=torch.float32, device=dy.device)
            wsd = workspace(dy.device)
            if (ctx.prec & _lib.M3T_GEMM_F16X3) and ctx.a_w is not None and CONV3D_PRESPLIT[0]:
                # both operands split ONCE (m3t_f16x3_split: dy channels-last under the slot its transpose raised, the weights as the
                # K-contiguous [ci][(tap, co)] matrix): the tap walk re-reads every dy row 27 times -- its loop is then copies and MFMAs only
                dy_img = dy_image()
                w_img = _dgrad_images(ctx, w, st, pd, (T_, H_, W_), ctx.a_w)[1][0]
                _lib.check(lib().m3t_conv3d_taps_pre(_p(dy_img), _p(w_img), _p(dx_cl), N_, Co, Ci, T_, H_, W_, To, Ho, Wo, kt, kh, kw,
                                                     pd[0], pd[1], pd[2], -1, slot_dy.data_ptr(), ctx.a_w, _p(wsd), wsd.numel() * 4, _p(dx),
                                                     _stream()), "m3t_conv3d_taps_pre")
            else:
                w_taps = w.detach().permute(2, 3, 4, 0, 1).contiguous().view(kt * kh * kw * Co, Ci)      # [(tap, co)][ci] (a few MB per layer)
                _lib.check(lib().m3t_conv3d_taps(_p(dy_cl), _p(w_taps), _p(dx_cl), N_, Co, Ci, T_, H_, W_, To, Ho, Wo, kt, kh, kw,
                                                 pd[0], pd[1], pd[2], -1, ctx.prec, slot_dy.data_ptr(), ctx.a_w, _p(wsd), wsd.numel() * 4, _p(dx),
                                                 _stream()), "m3t_conv3d_taps")
        if strided_dx:
            To, Ho, Wo = dy.shape[2], dy.shape[3], dy.shape[4]
            dx = torch.zeros(N_, Ci, T_, H_, W_, dtype=torch.float32, device=dy.device)        # (classes without a tap stay zero: 1 x 1 stride-2 shortcuts)
            wsd = workspace(dy.device)
            dy_img = dy_image()
            plan, imgs = _dgrad_images(ctx, w, st, pd, (T_, H_, W_), ctx.a_w)
            for (cls, r, sub, size, base), w_img in zip(plan, imgs):
                crow = N_ * size[0] * size[1] * size[2]
                dxc_cl = torch.empty(crow, Ci, dtype=torch.float32, device=dy.device)
                dxc = torch.empty(N_, Ci, size[0], size[1], size[2], dtype=torch.float32, device=dy.device)
                _lib.check(lib().m3t_conv3d_taps_pre(_p(dy_img), _p(w_img), _p(dxc_cl), N_, Co, Ci, size[0], size[1], size[2], To, Ho, Wo,
                                                     sub[0], sub[1], sub[2], base[0], base[1], base[2], -1, slot_dy.data_ptr(), ctx.a_w,
                                                     _p(wsd), wsd.numel() * 4, _p(dxc), _stream()), "m3t_conv3d_taps_pre")
                dx[:, :, cls[0]::st[0], cls[1]::st[1], cls[2]::st[2]].copy_(dxc)
        if ctx.needs_input_grad[1] and impl:
            # the walk turned round: dW^T[(tap, ci)][co] summed over dy's rows, x channels-last from the forward pass (m3t_conv3d_wgrad_taps)
            taps, Kc = kt * kh * kw, impl * kt * kh * kw                      # (impl = 4 for a first layer: its channels padded)
            Mp = (Kc + 127) // 128 * 128
            tiles = (Mp // 128) * ((Co + 127) // 128)
            want = max(1, min((1536 + tiles - 1) // tiles, rows // 256))
            imgs = getattr(ctx, "x_is_img", False)
            dy_op = dy_image() if imgs else dy_cl
            # round 6: a gradient that goes into a sink feeds nothing on the chain -- its walk leaves for a weight-gradient stream (as the GRU levels'
            # and nn.Linear's do) and runs beside the BatchNorm / pooling / gate backward of the layers in front, which are HBM-bound
            wg = _conv_wgrad_stream(dy.device, w_sink)
            with (on_stream(wg) if wg is not None else _NULL):
                wsw = workspace(dy.device, min(want * Mp * Co * 4, 512 << 20))
                dwt = torch.empty(Mp, Co, dtype=torch.float32, device=dy.device)
                _lib.check(lib().m3t_conv3d_wgrad_taps(_p(saved[2]), _p(dy_op), _p(dwt), N_, impl, Co, T_, H_, W_, kt, kh, kw,
                                                       st[0], st[1], st[2], pd[0], pd[1], pd[2], ctx.prec | (_lib.M3T_CONV_IMAGES if imgs else 0),
                                                       getattr(ctx, "a_x", None) or saved[3].data_ptr(), slot_dy.data_ptr(), _p(wsw), wsw.numel() * 4, _stream()),
                           "m3t_conv3d_wgrad_taps")
                dw_v = dwt[:Kc].view(taps, impl, Co)[:, :Ci].permute(2, 1, 0)              # [co][ci][tap], strided
                if w_sink is not None:
                    w_sink.view(Co, Ci, taps).copy_(dw_v)
                else:
                    dw = dw_v.contiguous().view_as(w)
            if wg is not None:
                for t_ in (saved[2], dy_op, saved[3], slot_dy):
                    t_.record_stream(wg)
                _WGRAD_PENDING[(dy.device.type, dy.device.index)] = True
        elif ctx.needs_input_grad[1]:
            Kc = Ci * kt * kh * kw
            xc = _req(x.contiguous(), "x")
            # round 4: ONE launch writes the patch matrix (rows (n, t', h', w'), columns (ci, kt, kh, kw), zero padded to the GEMM's tiles)
            # and raises its magnitude slot -- was ~50 torch copy kernels per convolution plus a measuring pass;
            # round 5: the forward pass already wrote it (kept) unless it ran on MIOpen
            slot = kept[1] if kept is not None else amax_slots(1, x.device)

            def im2col(rows_p, Kp):
                if kept is not None and tuple(kept[0].shape) == (rows_p, Kp):
                    return kept[0]
                pat_ = torch.empty(rows_p, Kp, dtype=torch.float32, device=x.device)
                sl_ = slot if kept is None else amax_slots(1, x.device)
                _lib.check(lib().m3t_im2col3d(_p(xc), N_, Ci, T_, H_, W_, kt, kh, kw, st[0], st[1], st[2], pd[0], pd[1], pd[2], _p(pat_),
                                              rows_p, Kp, C.c_void_p(sl_.data_ptr()), _stream()), "m3t_im2col3d")
                return pat_

            if Co % 128 == 0 and Kc % 64 == 0:
                pat = im2col(rows, Kc)
                dw = w_sink if w_sink is not None else torch.empty_like(w)
                sgemm(1, 0, Co, Kc, rows, dy_cl, 0, Co, pat, 0, Kc, dw, 0, Kc, prec=ctx.prec, amax=(slot_dy.data_ptr(), slot.data_ptr()))
                if w_sink is not None:
                    dw = None
            elif Co % 64 == 0:
                # the stems' FIRST layers: C_out = 64 and C_in k^3 = 81 (VGG-M) / 735 (3-D ResNet) fit no interior tile of the
                # 16-bit-term GEMM as dW = dy^T P.  Transposed and padded they do: dW^T [Kp, Co] = P_pad^T dy with Kp = ceil128(C_in k^3)
                # (zero columns) and rows padded to a multiple of 32 -> the 128 x 64 tile, split-K over the ~1.5 M rows
                Kp, rows_p = (Kc + 127) // 128 * 128, (rows + 31) // 32 * 32
                pat_pad = im2col(rows_p, Kp)
                dyp = dy_cl
                if rows_p != rows:
                    dyp = torch.zeros(rows_p, Co, dtype=torch.float32, device=x.device)
                    dyp[:rows].copy_(dy_cl.view(rows, Co))
                dwt = torch.empty(Kp, Co, dtype=torch.float32, device=x.device)
                sgemm(1, 0, Kp, Co, rows_p, pat_pad, 0, Kp, dyp, 0, Co, dwt, 0, Co, prec=ctx.prec, amax=(slot.data_ptr(), slot_dy.data_ptr()))
                if w_sink is not None:
                    w_sink.view(Co, Kc).copy_(dwt[:Kc].t())
                else:
                    dw = dwt[:Kc].t().contiguous().view_as(w)
            else:
                Kp = (Kc + 3) // 4 * 4
                pat = im2col(rows, Kp)
                dwp = torch.empty(Co, Kp, dtype=torch.float32, device=x.device)
                sgemm(1, 0, Co, Kp, rows, dy_cl, 0, Co, pat, 0, Kp, dwp, 0, Kp, prec=ctx.prec, amax=(slot_dy.data_ptr(), slot.data_ptr()))
                if w_sink is not None:
                    w_sink.view(Co, Kc).copy_(dwp[:, :Kc])
                else:
                    dw = dwp[:, :Kc].contiguous().view_as(w)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = b_sink if b_sink is not None else torch.empty(Co, dtype=torch.float32, device=dy.device)
            colsum(dpart, 0, dpart.shape[0], Co, Co, db)
            if b_sink is not None:
                db = None
        return dx, dw, db


def conv3d(x, w, b, stride, padding):
    _X_SLOT[0] = _tagged_amax(x) if x.is_contiguous() else None
    try:
        return _Conv3dGemmWgrad.apply(x, w, b, tuple(stride), tuple(padding), False)
    finally:
        _X_SLOT[0] = None


# ----------------------------------------------------------------------------- channels-last 3-D stems (round 6)
# The VGG-M stems (reference models/backbone.py:73-103,179-271: five groups of Conv3d -> BatchNorm3d -> ReLU (-> MaxPool3d((1, 2, 2)))) as a chain
# of channels-last operators: the tap walks read and write [rows = N T H W][C] natively, BatchNorm and pooling get channels-last kernels
# (csrc/stem_cl.hip), and nothing between the video and the GRU input is transposed (until round 5: a planes -> channels-last transpose in front
# of every convolution and of every convolution's backward, ~25 launches and 0.85 ms per C5 step).  M3T_STEM_CL=0: the planes operators.
STEM_CL = [os.environ.get("M3T_STEM_CL", "1") != "0"]


class CLTensor:
    """a channels-last activation of a 3-D stem: `data` [N T H W, C] (a plain autograd tensor) + its grid; `slot`: the 1-element tensor
    holding data's magnitude slot (raised by the kernel that wrote data) or None.  A BatchNorm3d + ReLU may be PENDING on it (bn_cl(...,
    lazy=True)): a pooling that follows applies it inside its own window loop (pool_cl: one fused operator, relu(bn(x)) at full resolution is
    never written); anything else that touches `.data` applies it first."""
    __slots__ = ("_data", "N", "T", "H", "W", "_slot", "_pending")

    def __init__(self, data, N, T, H, W, slot=None, pending=None):
        self._data, self.N, self.T, self.H, self.W, self._slot, self._pending = data, N, T, H, W, slot, pending

    def _materialize(self):
        if self._pending is not None:
            args, self._pending = self._pending, None
            self._slot = amax_slots(1, self._data.device)
            self._data = _BNCL.apply(self._data, *args, self._slot)

    @property
    def data(self):
        self._materialize()
        return self._data

    @property
    def slot(self):
        self._materialize()
        return self._slot

    @property
    def C(self):
        return self._data.shape[1]

    def planes(self):
        """[N, C, T, H, W] (one tiled transpose: leaving the chain)"""
        return btc_to_bct(self.data.view(self.N, self.T * self.H * self.W, self.C)).view(self.N, self.C, self.T, self.H, self.W)


_GRAD_SLOT = {}          # id of a gradient tensor OBJECT -> (weakref, slot tensor, version): slots of gradients handed from backward to backward


def _note_grad_slot(t, slot, colsum=None):
    if len(_GRAD_SLOT) > 64:
        _GRAD_SLOT.clear()
    _GRAD_SLOT[id(t)] = (weakref.ref(t), slot, t._version, t.data_ptr(), colsum)


def _grad_slot(t):
    """the magnitude slot the producing backward kernel raised for the gradient tensor `t` -- only if autograd handed over that very object,
    unmodified (a sum of two branches' gradients, a copy or a view is another object: the consumer measures)"""
    e = _GRAD_SLOT.pop(id(t), None)
    if e is None or e[0]() is not t or e[2] != t._version or e[3] != t.data_ptr():
        return None, None
    return e[1], e[4]


def conv3d_cl_ok(x, w, stride, padding, groups, dilation, padding_mode):
    """the channels-last chain covers this convolution (an fp32 device input; C_out % 64; C_in % 32, or a first layer with <= 4 channels and
    <= 8 taps per row; the fp16x3 mode)"""
    if not (STEM_CL[0] and CONV3D_GEMM[0] and CONV3D_IMPLICIT[0] and CONV3D_TAPS[0] and CONV3D_PRESPLIT[0] and _PREC[0] == _lib.M3T_GEMM_F16X3):
        return False
    if groups != 1 or tuple(dilation) != (1, 1, 1) or padding_mode != "zeros" or not isinstance(padding, tuple):
        return False
    Co, Ci, kt, kh, kw = w.shape
    if Co % 64 != 0:
        return False
    if isinstance(x, CLTensor):
        return Ci % 32 == 0 and x.C == Ci
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and Ci <= 4 and kw <= 8 and not x.requires_grad


class _Conv3dCL(torch.autograd.Function):
    """Conv3d on channels-last rows: x [N T H W, Ci] (or the video planes [N, <= 4, T, H, W] of a first layer) -> y [N T' H' W', Co]; every pass
    one of the tap walks of _Conv3dGemmWgrad, without any transpose"""

    @staticmethod
    def forward(ctx, x, w, b, geo):
        N_, T_, H_, W_, stride, padding, first, x_slot = geo
        Co, Ci, kt, kh, kw = w.shape
        To = (T_ + 2 * padding[0] - kt) // stride[0] + 1
        Ho = (H_ + 2 * padding[1] - kh) // stride[1] + 1
        Wo = (W_ + 2 * padding[2] - kw) // stride[2] + 1
        rows, srows, taps = N_ * To * Ho * Wo, N_ * T_ * H_ * W_, kt * kh * kw
        ctx.geo, ctx.out_grid, ctx.has_bias, ctx.prec = geo, (To, Ho, Wo), b is not None, _PREC[0]
        ctx.sink_refs = (w if id(w) in _GRAD_SINKS else None, b if (b is not None and id(b) in _GRAD_SINKS) else None)
        ctx.w_keep = []
        slots = amax_slots(2, x.device)
        a_w = weight_amax(w, ctx.w_keep)
        if first:
            cw = 4
            w8 = torch.zeros(Co, kt, kh, 8, 4, dtype=torch.float32, device=x.device)
            w8[:, :, :, :kw, :Ci].copy_(w.detach().permute(0, 2, 3, 4, 1))
            w_t = w8.view(Co, kt * kh * 32)
            x_cl = torch.empty(srows, 4, dtype=torch.float32, device=x.device)
            a_x = slots.data_ptr()
            amax_out(a_x)
            _lib.check(lib().m3t_planes_to_cl4(_p(_req(x.contiguous(), "x")), _p(x_cl), N_, Ci, T_ * H_ * W_, _stream()), "m3t_planes_to_cl4")
        else:
            cw = Ci
            w_t = None                               # (the [co][(tap, ci)] image is written straight from w: m3t_f16x3_split_perm)
            x_cl = _req(x, "x")
            if x_slot is not None:
                a_x = x_slot.data_ptr()
                ctx.w_keep.append(x_slot)
            else:
                a_x = slots.data_ptr()
                measure_amax([(x_cl, a_x)])
        wc = _req(w.detach().contiguous(), "weight")
        if a_w is None:
            a_w = slots.data_ptr() + 8
            if not measure_amax([(w_t if w_t is not None else wc.view(Co, -1), a_w)]):      # (max |w|: the same for every permutation)
                raise M3THipError("conv3d_cl: the weights cannot be measured (16-B alignment, C_in k^3 % 4)")
        wk = w_t.shape[1] if w_t is not None else taps * Ci
        x_img = torch.empty_like(x_cl)
        w_img = torch.empty(Co, wk, dtype=torch.float32, device=x.device)
        y_cl = torch.empty(rows, Co, dtype=torch.float32, device=x.device)
        wsd = workspace(x.device)
        _lib.check(lib().m3t_f16x3_split(_p(x_cl), srows, cw, cw, _p(x_img), cw, a_x, _stream()), "m3t_f16x3_split")
        if w_t is not None:
            _lib.check(lib().m3t_f16x3_split(_p(w_t), Co, wk, wk, _p(w_img), wk, a_w, _stream()), "m3t_f16x3_split")
        else:
            _lib.check(lib().m3t_f16x3_split_perm(_p(wc), Co, taps, Ci, Ci * taps, 1, taps, _p(w_img), a_w, _stream()), "m3t_f16x3_split_perm")
        tail = (kt, kh, kw, stride[0], stride[1], stride[2], padding[0], padding[1], padding[2], a_x, a_w, _p(wsd), wsd.numel() * 4, None, _stream())
        bp = _p(b) if b is not None else None
        if first:
            _lib.check(lib().m3t_conv3d_fwd_taps4(_p(x_img), _p(w_img), bp, _p(y_cl), N_, Co, T_, H_, W_, *tail), "m3t_conv3d_fwd_taps4")
        else:
            _lib.check(lib().m3t_conv3d_fwd_taps(_p(x_img), _p(w_img), bp, _p(y_cl), N_, Ci, Co, T_, H_, W_, *tail), "m3t_conv3d_fwd_taps")
        ctx.x_is_img = bool(WGRAD_IMAGES[0] and not first and bool(_PREC[0] & _lib.M3T_GEMM_F16X3))      # (see _Conv3dGemmWgrad: the weight gradient reads images)
        ctx.save_for_backward(w, x_img if ctx.x_is_img else x_cl, slots)
        ctx.a_x, ctx.a_w, ctx.cw = a_x, a_w, cw
        CONV3D_CALLS["walk"] += 1
        return y_cl

    @staticmethod
    def backward(ctx, dy):
        w, x_cl, slots = ctx.saved_tensors
        N_, T_, H_, W_, st, pd, first, _ = ctx.geo
        To, Ho, Wo = ctx.out_grid
        Co, Ci, kt, kh, kw = w.shape
        # dy's magnitude slot and its column sums (= this layer's bias gradient), both produced by the kernel that wrote dy (BatchNorm's dx pass)
        slot_dy, dy_colsum = _grad_slot(dy) if dy.is_contiguous() else (None, None)
        dy_cl = _req(dy.contiguous(), "dy")
        rows = dy_cl.shape[0]
        dx = dw = db = None
        w_sink = _take_sink(ctx.sink_refs[0]) if (ctx.needs_input_grad[1] and ctx.sink_refs[0] is not None) else None
        b_sink = _take_sink(ctx.sink_refs[1]) if (ctx.has_bias and ctx.needs_input_grad[2] and ctx.sink_refs[1] is not None) else None
        if ctx.needs_input_grad[1] and ctx.sink_refs[0] is not None and w_sink is None:
            join_wgrad(dy.device)      # (the sink was taken already this step: autograd adds the tensor returned below onto a slice a weight-gradient stream may still write -- see _Linear)
        if slot_dy is None:
            slot_dy = amax_slots(1, dy.device)
            measure_amax([(dy_cl, slot_dy.data_ptr())])
        wsd = workspace(dy.device)
        dy_img = None
        if ctx.needs_input_grad[0] or (ctx.needs_input_grad[1] and ctx.x_is_img):
            dy_img = torch.empty_like(dy_cl)
            _lib.check(lib().m3t_f16x3_split(_p(dy_cl), rows, Co, Co, _p(dy_img), Co, slot_dy.data_ptr(), _stream()), "m3t_f16x3_split")
        if ctx.needs_input_grad[0]:
            if first or Ci % 64 != 0 or Co % 32 != 0:
                raise M3THipError("the channels-last chain has no data gradient for this layer (a first layer's input is the video)")
            dx = torch.empty(N_ * T_ * H_ * W_, Ci, dtype=torch.float32, device=dy.device)
            one = tuple(st) == (1, 1, 1)
            if not one:
                dx.zero_()
            plan, imgs = _dgrad_images(ctx, w, st, pd, (T_, H_, W_), ctx.a_w)
            for (cls, r, sub, size, base), w_img in zip(plan, imgs):
                dst = dx if one else torch.empty(N_ * size[0] * size[1] * size[2], Ci, dtype=torch.float32, device=dy.device)
                _lib.check(lib().m3t_conv3d_taps_pre(_p(dy_img), _p(w_img), _p(dst), N_, Co, Ci, size[0], size[1], size[2], To, Ho, Wo,
                                                     sub[0], sub[1], sub[2], base[0], base[1], base[2], -1, slot_dy.data_ptr(), ctx.a_w,
                                                     _p(wsd), wsd.numel() * 4, None, _stream()), "m3t_conv3d_taps_pre")
                if not one:
                    dx.view(N_, T_, H_, W_, Ci)[:, cls[0]::st[0], cls[1]::st[1], cls[2]::st[2], :].copy_(dst.view(N_, size[0], size[1], size[2], Ci))
        if ctx.needs_input_grad[1]:
            cw = ctx.cw
            taps, Kc = kt * kh * kw, cw * kt * kh * kw
            Mp = (Kc + 127) // 128 * 128
            tiles = (Mp // 128) * ((Co + 127) // 128)
            want = max(1, min((1536 + tiles - 1) // tiles, rows // 256))
            imgs = ctx.x_is_img
            dy_op = dy_img if imgs else dy_cl
            wg = _conv_wgrad_stream(dy.device, w_sink)           # (off the chain when the gradient goes into a sink: see _Conv3dGemmWgrad._backward5)
            with (on_stream(wg) if wg is not None else _NULL):
                wsw = workspace(dy.device, min(want * Mp * Co * 4, 512 << 20))
                dwt = torch.empty(Mp, Co, dtype=torch.float32, device=dy.device)
                _lib.check(lib().m3t_conv3d_wgrad_taps(_p(x_cl), _p(dy_op), _p(dwt), N_, cw, Co, T_, H_, W_, kt, kh, kw,
                                                       st[0], st[1], st[2], pd[0], pd[1], pd[2], ctx.prec | (_lib.M3T_CONV_IMAGES if imgs else 0),
                                                       ctx.a_x, slot_dy.data_ptr(), _p(wsw), wsw.numel() * 4, _stream()),
                           "m3t_conv3d_wgrad_taps")
                dw_v = dwt[:Kc].view(taps, cw, Co)[:, :Ci].permute(2, 1, 0)
                if w_sink is not None:
                    w_sink.view(Co, Ci, taps).copy_(dw_v)
                else:
                    dw = dw_v.contiguous().view_as(w)
            if wg is not None:
                for t_ in (x_cl, dy_op, slots, slot_dy) + tuple(ctx.w_keep):
                    if torch.is_tensor(t_):
                        t_.record_stream(wg)
                _WGRAD_PENDING[(dy.device.type, dy.device.index)] = True
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = b_sink if b_sink is not None else torch.empty(Co, dtype=torch.float32, device=dy.device)
            if dy_colsum is not None:
                db.copy_(dy_colsum)
            else:
                colsum(dy_cl, 0, rows, Co, Co, db)
            if b_sink is not None:
                db = None
        return dx, dw, db, None


def conv3d_cl(x, w, b, stride, padding):
    """x: CLTensor, or the video planes [N, <= 4, T, H, W] of a stem's first layer -> CLTensor"""
    stride, padding = tuple(stride), tuple(padding)
    if isinstance(x, CLTensor):
        geo = (x.N, x.T, x.H, x.W, stride, padding, False, x.slot)
        data = x.data
    else:
        geo = (x.shape[0], x.shape[2], x.shape[3], x.shape[4], stride, padding, True, None)
        data = x
    y = _Conv3dCL.apply(data, w, b, geo)
    k = w.shape[2:]
    To, Ho, Wo = ((d + 2 * p_ - k_) // s_ + 1 for d, p_, k_, s_ in zip(geo[1:4], padding, k, stride))
    return CLTensor(y, geo[0], To, Ho, Wo, None)


class _BNCL(torch.autograd.Function):
    """BatchNorm3d (+ReLU) over channels-last rows (csrc/stem_cl.hip m3t_bn_cl_*); y's and dx's magnitude slots are raised by the kernels"""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, training, momentum, eps, relu, y_slot):
        x = _req(x, "x")
        M, Cc = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
        ws = workspace(x.device, int(lib().m3t_bn_cl_ws_bytes(M, Cc)))
        amax_out(y_slot.data_ptr() if y_slot is not None else None)
        try:
            rc = lib().m3t_bn_cl_fwd(_p(x), M, Cc, _p(gamma), _p(beta), _p(run_mean), _p(run_var), float(momentum), float(eps), int(training),
                                     int(relu), _p(y), _p(stats[0]), _p(stats[1]), _p(ws), ws.numel() * 4, _stream())
            _lib.check(rc, "m3t_bn_cl_fwd")
        except BaseException:
            _amax_clear()
            raise
        ctx.save_for_backward(x, y if relu else None, gamma, stats)
        ctx.training, ctx.relu = bool(training), bool(relu)
        ctx.sink_refs = (gamma if (gamma is not None and id(gamma) in _GRAD_SINKS) else None,
                         beta if (beta is not None and id(beta) in _GRAD_SINKS) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, stats = ctx.saved_tensors
        dy = _req(dy.contiguous(), "dy")
        M, Cc = x.shape
        dx = torch.empty_like(x)
        g = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
        ws = workspace(x.device, int(lib().m3t_bn_cl_ws_bytes(M, Cc)))
        gs = _take_sink(ctx.sink_refs[0]) if (ctx.needs_input_grad[1] and ctx.sink_refs[0] is not None) else None
        bs = _take_sink(ctx.sink_refs[1]) if (ctx.needs_input_grad[2] and ctx.sink_refs[1] is not None) else None
        slot = amax_slots(1, x.device)
        csum = torch.empty(Cc, dtype=torch.float32, device=x.device)
        amax_out(slot.data_ptr())
        try:
            rc = lib().m3t_bn_cl_bwd(_p(dy), _p(x), _p(y), _p(gamma), _p(stats[0]), _p(stats[1]), M, Cc, int(ctx.training), int(ctx.relu), _p(dx),
                                     _p(gs if gs is not None else g[0]), _p(bs if bs is not None else g[1]), _p(csum), _p(ws), ws.numel() * 4,
                                     _stream())
            _lib.check(rc, "m3t_bn_cl_bwd")
        except BaseException:
            _amax_clear()
            raise
        # the convolution in front of this BatchNorm takes dx as its dy: its magnitude slot and its column sums (that layer's bias gradient) ride along
        _note_grad_slot(dx, slot, csum)
        return (dx, (g[0] if (gamma is not None and gs is None) else None), (g[1] if (gamma is not None and bs is None) else None),
                None, None, None, None, None, None, None)


def bn_cl(x, gamma, beta, run_mean, run_var, training, momentum, eps, relu=True, lazy=False):
    """lazy (with relu): the operator is left PENDING on the result -- a pooling with tiling windows that follows fuses it (pool_cl)"""
    if lazy and relu and BN_POOL_FUSED[0]:
        return CLTensor(x.data, x.N, x.T, x.H, x.W, None, (gamma, beta, run_mean, run_var, training, momentum, eps, True))
    slot = amax_slots(1, x.data.device)
    y = _BNCL.apply(x.data, gamma, beta, run_mean, run_var, training, momentum, eps, relu, slot)
    return CLTensor(y, x.N, x.T, x.H, x.W, slot)


BN_POOL_FUSED = [os.environ.get("M3T_BN_POOL_FUSED", "1") != "0"]      # 0: BatchNorm + ReLU and the pooling as two operators (A/B)


class _BNPoolCL(torch.autograd.Function):
    """BatchNorm3d + ReLU + MaxPool3d((1, k, k), stride (1, k, k)) on channels-last frames as one operator (csrc/stem_cl.hip m3t_bn_pool_cl_*:
    reference models/backbone.py:77-80,83-86,89-92): relu(bn(x)) at full resolution is neither written nor read, forward or backward"""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, training, momentum, eps, geo, y_slot):
        P, H, W, k = geo
        x = _req(x, "x")
        Cc = x.shape[1]
        Ho, Wo = H // k, W // k
        yp = torch.empty(P * Ho * Wo, Cc, dtype=torch.float32, device=x.device)
        win = torch.empty(P * Ho * Wo, Cc, dtype=torch.uint8, device=x.device)
        stats = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
        ws = workspace(x.device, int(lib().m3t_bn_cl_ws_bytes(P * H * W, Cc)))
        amax_out(y_slot.data_ptr() if y_slot is not None else None)
        try:
            rc = lib().m3t_bn_pool_cl_fwd(_p(x), P, H, W, Cc, k, _p(gamma), _p(beta), _p(run_mean), _p(run_var), float(momentum), float(eps),
                                          int(training), _p(yp), C.c_void_p(win.data_ptr()), _p(stats[0]), _p(stats[1]), _p(ws), ws.numel() * 4,
                                          _stream())
            _lib.check(rc, "m3t_bn_pool_cl_fwd")
        except BaseException:
            _amax_clear()
            raise
        ctx.save_for_backward(x, yp, win, gamma, stats)
        ctx.geo, ctx.training = geo, bool(training)
        ctx.sink_refs = (gamma if (gamma is not None and id(gamma) in _GRAD_SINKS) else None,
                         beta if (beta is not None and id(beta) in _GRAD_SINKS) else None)
        return yp

    @staticmethod
    def backward(ctx, dy):
        x, yp, win, gamma, stats = ctx.saved_tensors
        P, H, W, k = ctx.geo
        dy = _req(dy.contiguous(), "dy")
        Cc = x.shape[1]
        dx = torch.empty_like(x)
        g = torch.empty(2, Cc, dtype=torch.float32, device=x.device)
        ws = workspace(x.device, int(lib().m3t_bn_cl_ws_bytes(P * H * W, Cc)))
        gs = _take_sink(ctx.sink_refs[0]) if (ctx.needs_input_grad[1] and ctx.sink_refs[0] is not None) else None
        bs = _take_sink(ctx.sink_refs[1]) if (ctx.needs_input_grad[2] and ctx.sink_refs[1] is not None) else None
        slot = amax_slots(1, x.device)
        csum = torch.empty(Cc, dtype=torch.float32, device=x.device)
        amax_out(slot.data_ptr())
        try:
            rc = lib().m3t_bn_pool_cl_bwd(_p(dy), _p(x), _p(yp), C.c_void_p(win.data_ptr()), _p(gamma), _p(stats[0]), _p(stats[1]), P, H, W, Cc, k,
                                          int(ctx.training), _p(dx), _p(gs if gs is not None else g[0]), _p(bs if bs is not None else g[1]),
                                          _p(csum), _p(ws), ws.numel() * 4, _stream())
            _lib.check(rc, "m3t_bn_pool_cl_bwd")
        except BaseException:
            _amax_clear()
            raise
        _note_grad_slot(dx, slot, csum)
        return (dx, (g[0] if (gamma is not None and gs is None) else None), (g[1] if (gamma is not None and bs is None) else None),
                None, None, None, None, None, None, None)


class _PoolCL(torch.autograd.Function):
    """nn.MaxPool3d((1, k, k)) on channels-last frames (csrc/stem_cl.hip m3t_pool_cl_*)"""

    @staticmethod
    def forward(ctx, x, geo, y_slot):
        P, H, W, k, s, p = geo
        x = _req(x, "x")
        Cc = x.shape[1]
        Ho, Wo = (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1
        y = torch.empty(P * Ho * Wo, Cc, dtype=torch.float32, device=x.device)
        win = torch.empty(P * Ho * Wo, Cc, dtype=torch.uint8, device=x.device)
        amax_out(y_slot.data_ptr() if y_slot is not None else None)
        try:
            _lib.check(lib().m3t_pool_cl_fwd(_p(x), P, H, W, Cc, k[0], k[1], s[0], s[1], p[0], p[1], _p(y), C.c_void_p(win.data_ptr()), _stream()),
                       "m3t_pool_cl_fwd")
        except BaseException:
            _amax_clear()
            raise
        ctx.save_for_backward(win)
        ctx.geo = geo
        return y

    @staticmethod
    def backward(ctx, dy):
        (win,) = ctx.saved_tensors
        P, H, W, k, s, p = ctx.geo
        dy = _req(dy.contiguous(), "dy")
        Cc = dy.shape[1]
        dx = torch.empty(P * H * W, Cc, dtype=torch.float32, device=dy.device)
        _lib.check(lib().m3t_pool_cl_bwd(_p(dy), C.c_void_p(win.data_ptr()), P, H, W, Cc, k[0], k[1], s[0], s[1], p[0], p[1], _p(dx), _stream()),
                   "m3t_pool_cl_bwd")
        return dx, None, None


def pool_cl(x, k, s, p):
    k, s, p = tuple(k), tuple(s), tuple(p)
    pend = x._pending
    if pend is not None and k == s and p == (0, 0) and k[0] == k[1] and k[0] in (2, 3) and x.H >= k[0] and x.W >= k[0]:
        gamma, beta, run_mean, run_var, training, momentum, eps, _ = pend
        slot = amax_slots(1, x._data.device)
        y = _BNPoolCL.apply(x._data, gamma, beta, run_mean, run_var, training, momentum, eps, (x.N * x.T, x.H, x.W, k[0]), slot)
        return CLTensor(y, x.N, x.T, x.H // k[0], x.W // k[0], slot)
    slot = amax_slots(1, x.data.device)
    y = _PoolCL.apply(x.data, (x.N * x.T, x.H, x.W, k, s, p), slot)
    Ho, Wo = (x.H + 2 * p[0] - k[0]) // s[0] + 1, (x.W + 2 * p[1] - k[1]) // s[1] + 1
    return CLTensor(y, x.N, x.T, Ho, Wo, slot)


def conv2d(x, w, b, stride, padding):
    """nn.Conv2d of the per-frame ResNet (reference models/resnet.py:18-24,95-105) on the 3-D walks with a unit time axis; w is the Conv2d
    Parameter itself (gradient sinks and the per-step magnitude table are keyed on it)"""
    _X_SLOT[0] = _tagged_amax(x) if x.is_contiguous() else None
    try:
        return _Conv3dGemmWgrad.apply(x, w, b, tuple(stride), tuple(padding), True)
    finally:
        _X_SLOT[0] = None
