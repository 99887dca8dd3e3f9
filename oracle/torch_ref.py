"""CPU baseline ("port") for bench.py's cpu_baseline leg -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The reference is pure Python on stock torch ops, so its CPU path IS `nn.GRU`, `nn.Linear`,
`weight_norm(nn.Conv1d)`, `F.softmax` ... called in the order of reference models/model.py:108-118,
models/rnn.py:71-81, models/att_fusion.py:18-27, models/tcn.py:43-46.  This file composes exactly
those stock ops (it contains none of the reference's source) so that timing it on the GPU box's host
cores is equivalent to timing the reference's CPU path there; the reference itself never travels.
It was validated against the imported reference in the build container through the shared golden
vectors (tests/test_torch_ref_golden.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class RefGRU(nn.Module):
    """== reference models/rnn.py GRU (stock nn.GRU + FC head), parameter names identical."""

    def __init__(self, input_size, hidden_size, num_layers, num_classes, num_fcs=1):
        super().__init__()
        self.num_classes = num_classes
        self.gru = nn.GRU(input_size, hidden_size, num_layers, batch_first=True, bidirectional=True)
        if num_classes > 0:
            if num_fcs == 1:
                self.fc = nn.Linear(2 * hidden_size, num_classes)
            else:
                dims = [2 * hidden_size] + [hidden_size] * (num_fcs - 1) + [num_classes]
                mods = []
                for i in range(num_fcs):
                    mods.append(nn.Linear(dims[i], dims[i + 1]))
                    if i != num_fcs - 1:
                        mods.append(nn.ReLU(True))
                self.fc = nn.Sequential(*mods)

    def forward(self, x):
        out, _ = self.gru(x)
        return self.fc(out) if self.num_classes > 0 else out


class RefAttFusion(nn.Module):
    def __init__(self, input_dim=(512, 512), hidden_dim=128):
        super().__init__()
        self.use_proj = input_dim[0] != input_dim[1]
        if self.use_proj:
            self.proj_v = nn.Linear(input_dim[1], input_dim[0])
        self.scorer_a = RefGRU(input_dim[0], hidden_dim, 1, 1, 1)
        self.scorer_v = RefGRU(input_dim[0], hidden_dim, 1, 1, 1)

    def forward(self, x_a, x_v):
        if self.use_proj:
            x_v = self.proj_v(x_v)
        h = torch.cat((torch.sigmoid(self.scorer_v(x_v)), torch.sigmoid(self.scorer_a(x_a))), dim=-1)
        h = F.softmax(h, dim=-1)
        return h[..., 0:1] * x_v + h[..., 1:2] * x_a


class RefAVFeatureGraph(nn.Module):
    """Config C3/C4 from stock ops; attribute names as AffWild2VA."""

    def __init__(self, d_a=128, d_v=256, num_hidden=512):
        super().__init__()
        self.audio = RefGRU(d_a, 256, 2, -1)
        self.visual = nn.Module()
        self.visual.gru_v = RefGRU(d_v, num_hidden, 2, -1)
        self.visual.gru_a = RefGRU(d_v, num_hidden, 2, -1)
        self.proj_v = nn.Linear(num_hidden * 4, 512)
        self.att_fuse = RefAttFusion((512, 512), 128)
        self.fusion = RefGRU(512, num_hidden, 2, 9, 2)

    def forward(self, x_a, x_v):
        a = self.audio(x_a)
        v = self.proj_v(torch.cat((self.visual.gru_v(x_v), self.visual.gru_a(x_v)), dim=-1))
        return self.fusion(self.att_fuse(a, v))


def ccc_loss(y_hat, y):
    """1 - concordance_cc2 over the flattened batch (unbiased var, biased cov)."""
    x, t = y_hat.reshape(-1), y.reshape(-1)
    mx, mt = x.mean(), t.mean()
    cov = ((x - mx) * (t - mt)).mean()
    return 1 - 2 * cov / (x.var() + t.var() + (mx - mt) ** 2)


def mtl_loss(y_hat, valence, arousal, class_expr, expr_valid, lam=0.5):
    loss = lam * ccc_loss(y_hat[..., 7], valence) + (1 - lam) * ccc_loss(y_hat[..., -1], arousal)
    ce = F.cross_entropy(y_hat[..., :7].reshape(-1, 7), class_expr.reshape(-1), reduction="none")
    return loss + 0.8 * (ce * expr_valid.reshape(-1).float()).mean()
