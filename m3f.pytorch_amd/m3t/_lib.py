"""ctypes binding of libm3t_hip.so -- the exact C ABI declared in include/m3t_hip.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# M3T_LIB_PATH: another build of the same library (A/B runs of kernel variants in tools/); default: the in-tree build
LIB_PATH = os.environ.get("M3T_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libm3t_hip.so")
CSRC_DIR = os.path.join(os.path.dirname(_HERE), "csrc")

M3T_EINVAL = 10001
M3T_ESPIN = 10002
M3T_SCAN_NO_PERSIST = 1
M3T_BF16 = 2
M3T_GEMM_BESIDE_SCAN = 512      # scheduling hint, see include/m3t_hip.h
M3T_GEMM_F16X3 = 1024   # two fp16 terms per scaled operand, three products: fp32-accurate (include/m3t_hip.h)
M3T_GEMM_HIGH = 256      # two bf16 terms per operand, four products: torch.set_float32_matmul_precision('high')
M3T_GEMM_EXCLUSIVE = 8
M3T_CONV_IMAGES = 4096      # m3t_conv3d_wgrad_taps: the operands are their m3t_f16x3_split images (include/m3t_hip.h)
M3T_SCAN_FP32 = 4
M3T_SCAN_WHH = 8
M3T_SCAN_FAULT = 16
M3T_SCAN_WIDE = 32        # room for a second persistent scan beside this one (include/m3t_hip.h)
M3T_MAX_SCANS = 8

_f = C.c_void_p      # device pointer
_i = C.c_int
_z = C.c_size_t
_s = C.c_void_p      # hipStream_t


class GruFwdDesc(C.Structure):
    _fields_ = [("xproj", _f), ("w_hh", _f), ("b_hh", _f), ("out", _f), ("gates", _f), ("h_n", _f),
                ("H", _i), ("reverse", _i), ("ldx", _i), ("xoff", _i), ("ldo", _i), ("ooff", _i)]


class GruBwdDesc(C.Structure):
    _fields_ = [("dout", _f), ("out", _f), ("gates", _f), ("w_hh_t", _f), ("dh_n", _f),
                ("dgx", _f), ("dgh", _f), ("dh", _f), ("db_part", _f), ("db_ih", _f), ("db_hh", _f),
                ("H", _i), ("reverse", _i), ("ldo", _i), ("ooff", _i), ("ldg", _i), ("goff", _i), ("amax", _f), ("wfrag", _f)]


class WindowProblem(C.Structure):
    _fields_ = [("A", _f), ("B", _f), ("C", _f), ("bias", _f), ("amax_a", _f), ("amax_b", _f), ("accumulate", _i)]


M3T_WINDOW_BATCH = 8

# name -> argtypes; the test-suite checks that every symbol of include/m3t_hip.h is here and exported.
SIGNATURES = {
    "m3t_version": [],
    "m3t_device_arch": [C.c_char_p, _i],
    "m3t_sgemm": [_i, _i, _i, _i, _i, _f, _i, _f, _i, _f, _i, _f, _i, _i, _i, _i, _i, _i, _f, _z, _i, _s],
    "m3t_sgemm_scaled": [_i, _i, _i, _i, _i, _f, _i, _f, _i, _f, _i, _f, _i, _i, _i, _i, _i, _i, _f, _z, _i, _f, _f, _s],
    "m3t_sgemm_window": [_i, _i, _i, _i, _i, _i, _i, _f, _i, _f, _i, _f, _i, _f, _i, _i, _i, _f, _f, _s],
    "m3t_sgemm_window_batch": [_i, C.POINTER(WindowProblem), _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _s],
    "m3t_absmax": [_i, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), _s],
    "m3t_sgemm_plan": [_i, _i, _i, _i, _i, _z, _i, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "m3t_im2col3d": [_f] + [_i] * 14 + [_f, C.c_longlong, _i, C.c_void_p, _s],
    "m3t_conv3d_taps": [_f, _f, _f] + [_i] * 17 + [_f, _f, _f, _z, _f, _s],
    "m3t_sgemm_pre": [_i, _i, _i, _f, _i, _f, _i, _f, _i, _f, _i, _i, _f, _f, _s],
    "m3t_f16x3_split": [_f, _z, _i, _z, _f, _z, _f, _s],
    "m3t_f16x3_split_perm": [_f, _z, _i, _i, _z, _z, _z, _f, _f, _s],
    "m3t_f16x3_image_b": [_f, _i, _i, _z, _f, _f, _s],
    "m3t_sgemm_bimg": [_i, _i, _i, _f, _i, _f, _f, _i, _f, _i, _i, _f, _z, _f, _f, _s],
    "m3t_sgemm_ring": [_i, _i, _i, _i, _i, _f, _i, _f, _i, _f, _i, _f, _i, _i, _i, _i, _i, _i, _f, _z, _i, _f, _f, _i, _s],
    "m3t_conv3d_taps_pre": [_f, _f, _f] + [_i] * 16 + [_f, _f, _f, _z, _f, _s],
    "m3t_conv3d_fwd_taps": [_f, _f, _f, _f] + [_i] * 15 + [_f, _f, _f, _z, _f, _s],
    "m3t_conv3d_fwd_taps4": [_f, _f, _f, _f] + [_i] * 14 + [_f, _f, _f, _z, _f, _s],
    "m3t_planes_to_cl4": [_f, _f, _i, _i, C.c_longlong, _s],
    "m3t_conv3d_wgrad_taps": [_f, _f, _f] + [_i] * 16 + [_f, _f, _f, _z, _s],
    "m3t_amax_out": [C.c_void_p],
    "m3t_colsum": [_f, _i, _i, _i, _f, _i, _f, _z, _s],
    "m3t_transpose": [_f, _i, _i, _i, _f, _i, _s],
    "m3t_relu_bwd": [_f, _f, _z, _s],
    "m3t_gru_scan_fwd": [C.POINTER(GruFwdDesc), _i, _i, _i, _f, _z, _i, _s],
    "m3t_gru_scan_bwd": [C.POINTER(GruBwdDesc), _i, _i, _i, _f, _z, _i, _s],
    "m3t_gru_bwd_prepare_floats": [_i],
    "m3t_gru_bwd_prepare": [C.POINTER(C.c_void_p), _i, _i, _i, C.POINTER(C.c_void_p), _s],
    "m3t_gru_persist_count": [],
    "m3t_gru_scan_workgroups": [_i, _i, _i, _i, _i, _i],
    "m3t_gru_poll_error": [],
    "m3t_gru_error_reset": [],
    "m3t_gru_error_defer": [_i],
    "m3t_gru_inject_error": [_s],
    "m3t_gru_persist_owner": [],
    "m3t_gru_scan_arena": [C.c_void_p, _z],
    "m3t_gru_scan_arena_reset": [C.c_void_p],
    "m3t_gru_scan_after": [C.c_void_p],
    "m3t_gru_scan_progress": [C.c_void_p, _i, C.POINTER(C.c_int), C.POINTER(C.c_uint)],
    "m3t_gru_scan_progress_reset": [C.c_void_p],
    "m3t_gru_scan_progress_ok": [_i, _i, _i, _i, _i, _i],
    "m3t_stream_wait_progress": [C.c_void_p, C.c_uint, _s],
    "m3t_gru_scan_events": [C.c_void_p, C.c_void_p],
    "m3t_gru_persist_profile": [C.c_void_p],
    "m3t_att_fuse_fwd": [_f, _f, _f, _f, _f, _i, _i, _s],
    "m3t_att_fuse_bwd": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _s],
    "m3t_bn_cl_ws_bytes": [_z, _i],
    "m3t_bn_cl_fwd": [_f, _z, _i, _f, _f, _f, _f, C.c_float, C.c_float, _i, _i, _f, _f, _f, _f, _z, _s],
    "m3t_bn_cl_bwd": [_f, _f, _f, _f, _f, _f, _z, _i, _i, _i, _f, _f, _f, _f, _f, _z, _s],
    "m3t_bn_pool_cl_fwd": [_f, _z, _i, _i, _i, _i, _f, _f, _f, _f, C.c_float, C.c_float, _i, _f, C.c_void_p, _f, _f, _f, _z, _s],
    "m3t_bn_pool_cl_bwd": [_f, _f, _f, C.c_void_p, _f, _f, _f, _z, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _z, _s],
    "m3t_pool_cl_fwd": [_f, _z, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, C.c_void_p, _s],
    "m3t_pool_cl_bwd": [_f, C.c_void_p, _z, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _s],
    "m3t_va_loss": [_f, _i, _i, _i, _i, _f, _f, _f, _f, _i, C.c_float, C.c_float, C.c_float, _i, _f, _f, _f, _z, _s],
    "m3t_va_loss_ws_bytes": [_i],
    "m3t_weight_norm_fwd": [_f, _f, _f, _f, _i, _i, _i, _s],
    "m3t_weight_norm_bwd": [_f, _f, _f, _f, _f, _f, _i, _i, _i, _s],
    "m3t_causal_conv_fwd": [_f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _s],
    "m3t_causal_conv_wgrad": [_f, _f, _f, _i, _i, _i, _i, _i, _i, _f, _z, _s],
    "m3t_conv1d_fwd": [_f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.c_float, C.c_ulonglong, _i, _s],
    "m3t_conv1d_fwd_scaled": [_f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.c_float, C.c_ulonglong, _i, _f, _f, _s],
    "m3t_conv1d_wgrad": [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _z, _i, _s],
    "m3t_conv1d_wgrad_scaled": [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _z, _i, _f, _f, _s],
    "m3t_pool_planes_fwd": [_f, C.c_longlong, _i, _i, _i, _i, _i, _i, _i, _i, _f, C.c_void_p, _s],
    "m3t_pool_planes_bwd": [_f, C.c_void_p, C.c_longlong, _i, _i, _i, _i, _i, _i, _i, _i, _f, _s],
    "m3t_bn_planes_ws_bytes": [_i, _i, _i],
    "m3t_bn_planes_fwd": [_f, _i, _i, _i, _f, _f, _f, _f, C.c_float, C.c_float, _i, _i, _f, _f, _f, _f, _z, _s],
    "m3t_bn_planes_bwd": [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _f, _z, _s],
    "m3t_bn_rows_ws_bytes": [_i, _i],
    "m3t_bn_rows_fwd": [_f, _i, _i, _f, _f, _f, _f, C.c_float, C.c_float, _i, _i, _f, _f, _f, _f, _z, _s],
    "m3t_bn_rows_bwd": [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _f, _f, _f, _f, _z, _s],
    "m3t_bct_to_btc": [_f, _f, _i, _i, _i, _s],
    "m3t_bct_to_btc_sums": [_f, _f, _i, _i, _i, _f, _s],
    "m3t_bct_to_btc_img": [_f, _f, _i, _i, _i, C.c_void_p, _f, _s],
    "m3t_btc_to_bct": [_f, _f, _i, _i, _i, _s],
    "m3t_mask_pos": [_f, _f, _f, _f, _z, _s],
    "m3t_add_relu": [_f, _f, _f, _z, _s],
    "m3t_mask_pos_drop": [_f, _f, _f, _i, _i, C.c_float, C.c_ulonglong, _s],
    "m3t_cbam_channel_fwd": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _s],
    "m3t_cbam_channel_bwd": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _f, _z, _s],
    "m3t_cbam_spatial_fwd": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, C.c_float, C.c_float, _f, _z, _s],
    "m3t_cbam_spatial_bwd": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _z, _s],
    "m3t_cbam_fused_ok": [_i, _i, _i, _i],
    "m3t_cbam_fused_ws_bytes": [_i, _i, _i, _i, _i],
    "m3t_cbam_fwd": [_f] * 20 + [_i] * 6 + [C.c_float, C.c_float, _f, _z, _s],
    "m3t_cbam_bwd": [_f] * 23 + [_i] * 6 + [_f, _z, _s],
    "m3t_smooth_tracks": [_f, _f, _i, _i, _i, _f, _s],
    "m3t_ccc_masked": [_f, _f, _f, C.c_longlong, _i, _f, _s],
    "m3t_frame_window": [_f, C.c_longlong, _i, _i, _i, _f, _f, C.c_longlong, _s],
    "m3t_power_spectrum": [_f, C.c_longlong, _i, _f, _s],
    "m3t_power_to_db": [_f, C.c_longlong, C.c_float, C.c_float, _f, _f, _z, _s],
    "m3t_stack_context": [_f, C.c_longlong, _i, C.c_longlong, _i, _i, _i, _f, _s],
    "m3t_grad_norm_scale": [_f, _z, C.c_float, C.c_float, _f, _f, _z, _s],
    "m3t_grad_poison": [_f, _f, _s],
    "m3t_grad_dead_check": [_f, _s],
    "m3t_adam_step": [_f, _f, _f, _f, _z, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _i, _f, _s],
    "m3t_sgd_step": [_f, _f, _f, _z, C.c_float, C.c_float, C.c_float, _i, _f, _s],
}

RESTYPES = {"m3t_gru_bwd_prepare_floats": C.c_size_t, "m3t_bn_rows_ws_bytes": C.c_size_t, "m3t_bn_planes_ws_bytes": C.c_size_t, "m3t_va_loss_ws_bytes": C.c_size_t, "m3t_bn_cl_ws_bytes": C.c_size_t, "m3t_cbam_fused_ws_bytes": C.c_size_t}

_lib = None


class M3THipError(RuntimeError):
    pass


def load():
    """Load libm3t_hip.so.  Fails loudly: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must load ITS bundled HIP runtime (libamdhip64.so.7, same SONAME as /opt/rocm's) before this
    # library is mapped: whichever copy is mapped first serves both, and torch cannot find the GPU on the other one.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise M3THipError(
            "libm3t_hip.so not found at %s -- build it with `make -C %s` (or __graft_entry__.build()). "
            "The M3T hot path has no CPU/eager fallback." % (LIB_PATH, CSRC_DIR))
    lib = C.CDLL(LIB_PATH)
    for name, argt in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch: loud by design
        fn.argtypes = argt
        fn.restype = RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def _recover_scan_error():
    """The error state is sticky on purpose (include/m3t_hip.h, error model): everything queued behind a dead scan must still
    see it on the device.  So: wait for the device -- every step queued so far has then been skipped by its own guard --
    and only then clear the state, just before the failure is raised to the caller (who may redo the step)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        try:
            torch.cuda.synchronize()
        except Exception:  # noqa: BLE001  (a device in an error state: the raise below still tells the caller)
            pass
    _lib.m3t_gru_error_reset()


def check(rc, what):
    if rc != 0:
        note = {M3T_EINVAL: " (M3T_EINVAL: bad arguments)",
                M3T_ESPIN: " (M3T_ESPIN: an earlier persistent GRU scan gave up waiting for a peer workgroup; its results are invalid)"}
        if rc == M3T_ESPIN and _lib is not None:
            _recover_scan_error()
        raise M3THipError("%s failed with code %d%s" % (what, rc, note.get(rc, " (hipError_t)")))


def poll_scan_error(what="persistent GRU scan"):
    """Raise M3THipError if a persistent scan has hit its spin limit (include/m3t_hip.h, error model).  No synchronisation
    happens on the healthy path: synchronise first when the answer must cover work still in flight.  On a failure the
    device is synchronised and the (sticky) error state cleared before the exception leaves."""
    if _lib is None:
        return
    step = _lib.m3t_gru_poll_error()
    if step:
        _recover_scan_error()
        raise M3THipError("%s: a workgroup gave up waiting for its peers at step %d (M3T_ESPIN); outputs and gradients of "
                          "that scan are invalid, every optimizer step queued behind it was skipped on the device -- is another "
                          "process running persistent scans on this GPU?" % (what, step - 1))
