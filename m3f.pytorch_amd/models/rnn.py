"""models.rnn -- BiGRU stack + per-timestep FC head, MI355X-native.

Drop-in for the reference's models/rnn.py:11-81 (`GRU`): same constructor, same
forward contract ([B,T,I] -> [B,T,2H] | [B,T,nC] | (out, h[2L,B,H])), same parameter
names/shapes (`gru.weight_ih_l{k}[_reverse]`, `gru.weight_hh_...`, `gru.bias_..`,
`fc.*`), same initialisation recipe (rnn.py:57-69) drawn in the same RNG order.
The arithmetic runs in libm3t_hip.so: input projections and FC layers on the fp32-MFMA
GEMM, the recurrence in the grouped per-step scan kernels (m3t.ops.multi_bigru).
`Attention/Decoder/AttEncDec` (rnn.py:84-165, --fusion_type att_dec) are out of scope
(non-deterministic teacher forcing).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from m3t import ops


class BiGRUParameters(nn.Module):
    """Parameter holder with nn.GRU's names, shapes, registration order and default init
    (bidirectional, batch_first).  No forward: the scan lives in m3t.ops."""

    def __init__(self, input_size, hidden_size, num_layers):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        H = hidden_size
        for layer in range(num_layers):
            in_l = input_size if layer == 0 else 2 * H
            for suffix in ("", "_reverse"):
                tag = "l%d%s" % (layer, suffix)
                self.register_parameter("weight_ih_" + tag, nn.Parameter(torch.empty(3 * H, in_l)))
                self.register_parameter("weight_hh_" + tag, nn.Parameter(torch.empty(3 * H, H)))
                self.register_parameter("bias_ih_" + tag, nn.Parameter(torch.empty(3 * H)))
                self.register_parameter("bias_hh_" + tag, nn.Parameter(torch.empty(3 * H)))
        bound = 1.0 / math.sqrt(H) if H > 0 else 0.0
        for p in self.parameters():
            nn.init.uniform_(p, -bound, bound)

    def flatten_parameters(self):   # reference calls it every forward (rnn.py:72); nothing to do here
        return None

    def flat(self):
        """[w_ih, w_hh, b_ih, b_hh] for (layer 0 fwd, layer 0 rev, layer 1 fwd, ...)."""
        out = []
        for layer in range(self.num_layers):
            for suffix in ("", "_reverse"):
                tag = "l%d%s" % (layer, suffix)
                out += [getattr(self, "weight_ih_" + tag), getattr(self, "weight_hh_" + tag),
                        getattr(self, "bias_ih_" + tag), getattr(self, "bias_hh_" + tag)]
        return out


def _build_head(hidden_size, num_classes, num_fcs, dropout):
    """FC head with the reference's child indices (rnn.py:20-55) so state_dict keys match."""
    H = hidden_size
    if num_fcs == 1:
        return nn.Linear(2 * H, num_classes)
    widths = [2 * H] + [H] * (num_fcs - 1) + [num_classes]
    mods = []
    for i in range(num_fcs):
        mods.append(nn.Linear(widths[i], widths[i + 1]))
        if i != num_fcs - 1:
            mods.append(nn.ReLU(True))
            if dropout:
                mods.append(nn.Dropout(0.5))
    return nn.Sequential(*mods)


class GRU(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, num_classes, num_fcs=1, dropout=False, return_h=False):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.num_classes = num_classes
        self.return_h = return_h
        self.gru = BiGRUParameters(input_size, hidden_size, num_layers)
        if num_classes > 0:
            if num_fcs not in (1, 2, 3):
                raise ValueError("num_fcs must be 1, 2 or 3")
            self.fc = _build_head(hidden_size, num_classes, num_fcs, dropout)
        self._init_gru(input_size, hidden_size)

    def _init_gru(self, input_size, hidden_size):
        # rnn.py:57-69: per-gate U(+-sqrt(3)*sqrt(2/(I+H))) for every weight_ih (layer-1 also uses the
        # layer-0 I), per-gate orthogonal weight_hh, zero biases.
        H = hidden_size
        lim = math.sqrt(3.0) * math.sqrt(2.0 / (input_size + hidden_size))
        with torch.no_grad():
            for name, p in self.gru.named_parameters():
                for g0 in range(0, 3 * H, H):
                    if "weight_ih" in name:
                        nn.init.uniform_(p[g0:g0 + H], -lim, lim)
                    elif "weight_hh" in name:
                        nn.init.orthogonal_(p[g0:g0 + H])
                    elif "bias" in name:
                        nn.init.constant_(p[g0:g0 + H], 0.0)

    # -- pieces exposed so that callers can advance several independent GRUs in one grouped scan
    def stack(self, x):
        return (x, self.gru.flat(), self.num_layers)

    def head(self, out):
        if self.num_classes <= 0:
            return out
        if isinstance(self.fc, nn.Linear):
            return ops.linear(out, self.fc.weight, self.fc.bias, 0)
        layers = [m for m in self.fc if isinstance(m, nn.Linear)]
        drops = [m for m in self.fc if isinstance(m, nn.Dropout)]
        for i, lin in enumerate(layers):
            last = i == len(layers) - 1
            out = ops.linear(out, lin.weight, lin.bias, 0 if last else 1)     # ReLU fused in the GEMM epilogue
            if not last and drops and self.training:
                # nn.Dropout(0.5) of the `dropout=True` heads (reference rnn.py:24-28,40-49): Philox mask generated inside
                # the kernel, forward and backward (no mask tensor, no torch RNG kernel); self.drop_seeds overrides the seeds (tests)
                seeds = getattr(self, "drop_seeds", None)
                out = ops.relu_dropout(out, drops[0].p, None if seeds is None else seeds[i])
        return out

    def forward(self, x):
        self.gru.flatten_parameters()
        ((out, h_n),) = ops.multi_bigru([self.stack(x)])
        out = self.head(out)
        if self.return_h:
            return out, h_n
        return out


def run_grus(modules, inputs):
    """Advance several independent `GRU` modules (same batch, length and depth) together:
    one grouped scan per layer level instead of one per module.  Returns the per-module
    outputs after their FC heads (and ignores return_h)."""
    res = ops.multi_bigru([m.stack(x) for m, x in zip(modules, inputs)])
    return [m.head(o) for m, (o, _) in zip(modules, res)]


def run_grus_cat(modules, inputs, lo, hi):
    """`run_grus` followed by `torch.cat(outputs[lo:hi], dim=-1)` (the visual towers' gru_v | gru_a features, reference
    models/model.py:111-112 / models/backbone.py:281-283): the list of outputs with the concatenation in place of the group.
    When the grouped modules end without an FC head (num_classes <= 0) their scans write straight into the concatenated buffer
    (m3t.ops.multi_bigru(cat=...)): no cat copy forward, no slice copies backward."""
    if hi - lo >= 2 and all(m.num_classes <= 0 for m in modules[lo:hi]):
        res = ops.multi_bigru([m.stack(x) for m, x in zip(modules, inputs)], cat=(lo, hi))
        outs = [m.head(o) if not (lo < i < hi) else None for i, (m, (o, _)) in enumerate(zip(modules, res))]
        return outs[:lo + 1] + outs[hi:]
    outs = run_grus(modules, inputs)
    return outs[:lo] + [torch.cat(outs[lo:hi], dim=-1)] + outs[hi:]
