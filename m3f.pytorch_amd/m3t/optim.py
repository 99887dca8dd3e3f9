"""Optimizer steps as single HIP kernels over the flat parameter / gradient buffers of m3t.ddp.FlatGradDDP
(SURVEY.md 8(f) row f-2).  Semantics = torch.optim.Adam / torch.optim.SGD as configured by the reference's
AffWild2VA.configure_optimizers (reference models/model.py:375-407): Adam(lr 5e-5, weight_decay 1e-4) or
SGD(momentum 0.9, weight_decay 5e-4)."""
import ctypes as C

import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr())


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Guarded:
    """the norm FlatGradDDP.finish() returned guards the step: m3t_adam_step / m3t_sgd_step change nothing when it is not
    finite (gradients of a failed persistent scan -- m3t_grad_norm_scale returns NaN for those -- or an overflow)"""

    def _guard(self):
        n = self.ddp.last_norm
        return _p(n) if n is not None else None


class FlatAdam(_Guarded):
    def __init__(self, ddp, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4):
        if ddp.flat_params is None:
            raise ValueError("FlatGradDDP(..., flatten_params=True) is required")
        self.ddp, self.lr, self.betas, self.eps, self.weight_decay = ddp, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(ddp.flat_params)
        self.v = torch.zeros_like(ddp.flat_params)
        self.t = 0

    def step(self):
        self.t += 1
        d = self.ddp
        rc = _lib.load().m3t_adam_step(_p(d.flat_params), _p(d.flat), _p(self.m), _p(self.v), d.flat.numel(), self.lr,
                                       self.betas[0], self.betas[1], self.eps, self.weight_decay, self.t, self._guard(), _s())
        _lib.check(rc, "m3t_adam_step")


class FlatSGD(_Guarded):
    def __init__(self, ddp, lr, momentum=0.9, weight_decay=5e-4):
        if ddp.flat_params is None:
            raise ValueError("FlatGradDDP(..., flatten_params=True) is required")
        self.ddp, self.lr, self.momentum, self.weight_decay = ddp, lr, momentum, weight_decay
        self.buf = torch.zeros_like(ddp.flat_params)
        self.t = 0

    def step(self):
        self.t += 1
        d = self.ddp
        rc = _lib.load().m3t_sgd_step(_p(d.flat_params), _p(d.flat), _p(self.buf), d.flat.numel(), self.lr, self.momentum,
                                      self.weight_decay, self.t, self._guard(), _s())
        _lib.check(rc, "m3t_sgd_step")
