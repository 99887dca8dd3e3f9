"""Benchmark / parity workloads assembled from the drop-in modules.

These are the reference's own graphs with the conv towers replaced by pre-computed per-frame
features -- exactly what BASELINE.json's configs describe ("precomputed 256-d SENet feats",
"256-d video + 128-d audio").  Attribute names follow AffWild2VA so state_dict keys line up.
"""
import os

import torch
import torch.nn as nn

from . import ops
from models.rnn import GRU, run_grus, run_grus_cat
from models.att_fusion import AttFusion
from models.tcn import TemporalConvNet


class AVFeatureGraph(nn.Module):
    """Config C3/C4: AffWild2VA.forward, audiovisual/attention (reference models/model.py:108-118)
    on feature inputs: audio GRU(d_a,256,2) | gru_v, gru_a GRU(d_v,H,2) on the SAME visual features
    (mirrors `se_features` passed twice, model.py:111) -> cat -> proj_v Linear(4H,512) ->
    AttFusion([512,512],128) -> fusion GRU(512,H,2,9,2)."""

    def __init__(self, d_a=128, d_v=256, num_hidden=512, fc_outputs=9, num_fc_layers=2):
        super().__init__()
        self.audio = GRU(d_a, 256, 2, -1, num_fc_layers)
        self.visual = nn.Module()
        self.visual.gru_v = GRU(d_v, num_hidden, 2, -1, num_fc_layers)
        self.visual.gru_a = GRU(d_v, num_hidden, 2, -1, num_fc_layers)
        self.proj_v = nn.Linear(num_hidden * 4, 512)
        self.att_fuse = AttFusion([512, 512], 128)
        self.fusion = GRU(512, num_hidden, 2, fc_outputs, num_fc_layers)

    def forward(self, x_a, x_v):
        a, v12 = run_grus_cat([self.audio, self.visual.gru_v, self.visual.gru_a], [x_a, x_v, x_v], 1, 3)      # v12 = cat(v1, v2)
        v = ops.linear(v12, self.proj_v.weight, self.proj_v.bias, 0)
        return self.fusion(self.att_fuse(a, v))


class TcnHead(nn.Module):
    """Config C1: TemporalConvNet(d_in,[H]*L,3) + Linear(H,2) composed as VA_3DVGGM's tcn back-end
    (reference models/backbone.py:107-111,139-141).  Input channel-first [B,d_in,T]."""

    def __init__(self, d_in=128, hidden=512, levels=2):
        super().__init__()
        self.tcn = nn.ModuleList([TemporalConvNet(d_in, [hidden] * levels, 3), nn.Linear(hidden, 2)])

    def forward(self, x):
        h = self.tcn[0].forward_btc(ops.bct_to_btc(x))
        return ops.linear(h, self.tcn[1].weight, self.tcn[1].bias, 0)


class TcnGru(nn.Module):
    """Config C2: TemporalConvNet(d_in,[H,H],3) -> GRU(H,H,2,2,2).  Input channel-first [B,d_in,T]."""

    def __init__(self, d_in=256, hidden=512):
        super().__init__()
        self.tcn = TemporalConvNet(d_in, [hidden, hidden], 3)
        self.gru = GRU(hidden, hidden, 2, 2, 2)

    def forward(self, x):
        return self.gru(self.tcn.forward_btc(ops.bct_to_btc(x)))


# ---------------------------------------------------------------------------------------------- the timed step
# ONE definition of "a step" for bench.py, tools/aux_bench.py and the full-size parity tests (tests/test_gpu_bench_path.py):
# what the test pins against the reference-generated goldens is, line for line, what the bench times.
def c3_bucket_order(model):
    """gradient buckets of the C3/C4 graph in the order backward finishes them (fusion -> att_fuse + proj_v -> encoders)"""
    return [list(model.fusion.parameters()),
            list(model.att_fuse.parameters()) + list(model.proj_v.parameters()),
            list(model.visual.parameters()) + list(model.audio.parameters())]


def make_c3_step(model, batch, max_norm=1.0, **ddp_kw):
    """(ddp, step) for the C3/C4 workload: zero_grad -> forward -> ccc_mtl loss (valence = output 7, arousal = output 8,
    reference models/model.py:153,164,176-177) -> backward -> FlatGradDDP.finish() (gradient all-reduce for N > 1, 1/N,
    clip at max_norm: reference train.py:35).  batch: dict x_a [B,T,d_a], x_v [B,T,d_v], valence, arousal, class_expr,
    expr_valid (device tensors).  step() returns (loss, stats, y)."""
    from .ddp import FlatGradDDP
    ddp = FlatGradDDP(model, bucket_order=c3_bucket_order(model), max_norm=max_norm, **ddp_kw)
    if os.environ.get("M3T_DDP_EARLY_BUCKET") == "1" and ddp.world > 1:
        # opt-in A/B for the first multi-GPU run: the fusion GRU's gradients are all-reduced beside the rest of backward.  What may
        # be resident at once in that window: the encoder level's wide launch + the audio launch (scorers: no residency requirement)
        from . import _lib
        B, T = batch["x_a"].shape[0], batch["x_a"].shape[1]
        fl = _lib.M3T_GEMM_F16X3
        wgs = (_lib.load().m3t_gru_scan_workgroups(4, 512, B, T, fl | _lib.M3T_SCAN_WIDE, 1) +
               _lib.load().m3t_gru_scan_workgroups(2, 256, B, T, fl, 1))
        n_cus = torch.cuda.get_device_properties(batch["x_a"].device).multi_processor_count
        ddp.early_bucket_after(model.fusion, wgs, n_cus)

    def step():
        ddp.zero_grad()
        y = model(batch["x_a"], batch["x_v"])
        loss, stats = ops.va_loss(y, batch["valence"], batch["arousal"], batch["class_expr"], batch["expr_valid"],
                                  iv=7, ia=8, n_expr=7, w_v=0.5, w_a=0.5, expr_w=0.8)
        loss.backward()
        ddp.finish()
        return loss, stats, y

    return ddp, step


def make_seq_step(model, x, valence, arousal, max_norm=1.0, **ddp_kw):
    """(ddp, step) for the single-input sequence workloads (C1 TcnHead, C2 TcnGru): `ccc` loss on the last two outputs
    (reference models/model.py:153-164 with loss='ccc'), FlatGradDDP over one bucket."""
    from .ddp import FlatGradDDP
    ddp = FlatGradDDP(model, bucket_order=[list(model.parameters())], max_norm=max_norm, **ddp_kw)

    def step():
        ddp.zero_grad()
        y = model(x)
        loss, stats = ops.va_loss(y, valence, arousal)
        loss.backward()
        ddp.finish()
        return loss, stats, y

    return ddp, step
