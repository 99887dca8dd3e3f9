"""m3t -- host-side plumbing of the MI355X-native M3T hot path.

`m3t._lib` binds the C ABI of libm3t_hip.so (include/m3t_hip.h) with ctypes;
`m3t.ops` wraps the entry points as torch.autograd Functions.  There is NO CPU or
eager fallback: every op raises if the HIP library is missing or a tensor is not a
contiguous fp32 device tensor.
"""
from . import _lib  # noqa: F401
