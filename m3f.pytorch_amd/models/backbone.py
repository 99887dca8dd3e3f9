"""models.backbone -- visual front-ends + temporal back-ends, MI355X-native temporal part.

API/state-dict compatible with the reference's models/backbone.py for the classes on the
north_star path: `VA_3DVGGM` (:62-161), `VA_3DVGGM_Split` (:164-311, the default
`--backbone v2p_split`) and `VA_3DResNet` (:314-372).  The 3-D stems' convolutions (forward, weight
and data gradient: tap-walk implicit GEMMs, rounds 5-6; the VGG-M stems as a channels-last chain),
BatchNorm3d + ReLU and spatial pooling, the temporal back-ends (BiGRU stacks, TCN) and the CBAM
gates / BatchNorm2d inside the ResNet run in the HIP library; a configuration the library does not
cover takes the stock op and says so once on stderr (m3t.ops.stock_fallback).  `VA_3DDenseNet` / `VA_VGGFace` are
out of scope (not reachable from AffWild2VA.forward).
"""
import math

import torch
import torch.nn as nn
from torch.nn.modules.utils import _triple

from m3t import ops
from .resnet import ResNet, ResNetV2, BasicBlock, BasicBlockV2
from .rnn import GRU, run_grus
from .tcn import TemporalConvNet, WeightNormConv1d


class Conv3d(nn.Conv3d):
    """nn.Conv3d (same parameters / state_dict keys) on m3t.ops.conv3d (round 5): forward, weight gradient and data gradient (round 6: the
    strided layers' too, as parity-class walks) as tap-walk implicit GEMMs over channels-last activations on the fp16x3 kernels -- no patch matrix (MIOpen's fp32 forward ran
    at 26 TFLOP/s); M3T_CONV3D_IMPLICIT=0: the patch-matrix GEMMs.  Falls back to the stock op for configurations it does not cover (and
    under M3T_CONV3D_MIOPEN=1)."""

    cl_chain = False          # set on the convolutions of a VGG-M stem (_vgg_group): run them as a channels-last chain (m3t.ops.CLTensor)

    def forward(self, x):
        # ONE path whatever the grad mode (round 6): validation / test steps under no_grad (reference models/model.py:226-246,320-337) and
        # --freeze_enc training (model.py:376-386) run the same walks as a training step; autograd skips what needs no gradient
        if isinstance(x, ops.CLTensor) or self.cl_chain:
            if ops.conv3d_cl_ok(x, self.weight, self.stride, self.padding, self.groups, self.dilation, self.padding_mode):
                return ops.conv3d_cl(x, self.weight, self.bias, self.stride, self.padding)
            if isinstance(x, ops.CLTensor):
                x = x.planes()                     # (a layer the chain does not cover: leave it)
        if (x.is_cuda and x.dtype == torch.float32 and self.groups == 1 and tuple(self.dilation) == (1, 1, 1)
                and self.padding_mode == "zeros" and isinstance(self.padding, tuple)):
            return ops.conv3d(x, self.weight, self.bias, self.stride, self.padding)
        ops.stock_fallback("models.backbone.Conv3d", "CPU / non-fp32 input, groups, dilation or a padding mode the walks do not cover")
        return super().forward(x)


class BatchNorm3dReLU(nn.BatchNorm3d):
    """nn.BatchNorm3d followed by ReLU as ONE operator on the HIP kernels (m3t.ops.bn_planes: the ReLU costs no pass over the
    activation of its own, forward or backward); same parameters, buffers and state_dict keys as nn.BatchNorm3d.  In the stems'
    nn.Sequential it takes the BatchNorm3d slot and an nn.Identity the ReLU's, so every index -- and every checkpoint key -- stays."""

    def forward(self, x):
        if isinstance(x, ops.CLTensor):
            Cc = x.C
            if ops.BN_PLANES[0] and self.affine and self.track_running_stats and self.momentum is not None and Cc % 4 == 0 and 256 % (Cc // 4) == 0:
                if self.training:
                    ops.count_batch(self.num_batches_tracked)
                return ops.bn_cl(x, self.weight, self.bias, self.running_mean, self.running_var, self.training, self.momentum, self.eps, True, lazy=True)
            x = x.planes()
        if (ops.BN_PLANES[0] and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and self.affine and self.track_running_stats
                and self.momentum is not None):
            if self.training:
                ops.count_batch(self.num_batches_tracked)
            return ops.bn_planes(x, self.weight, self.bias, self.running_mean, self.running_var, self.training, self.momentum,
                                 self.eps, True)
        ops.stock_fallback("models.backbone.BatchNorm3dReLU", "M3T_BN_PLANES=0, CPU / non-fp32 input or a configuration without running statistics")
        return torch.relu(super().forward(x))


class SpatialMaxPool3d(nn.MaxPool3d):
    """nn.MaxPool3d whose window does not span frames (kernel (1, k, k), stride (1, s, s): every pooling layer of the stems, reference
    backbone.py:80,86,92,182) as a pooling of the N C T planes on the HIP kernels (m3t.ops.pool_planes: one byte per output remembers
    the winner, the backward pass is a gather -- torch's 3-D kernels took 1.3 ms of the 32 ms C5 step, its 2-D backward 0.8 ms of the
    ResNet3D step).  No parameters.  Other configurations and inputs take the stock op."""

    def forward(self, x):
        k, s, p, d = (_triple(v) for v in (self.kernel_size, self.stride, self.padding, self.dilation))
        if isinstance(x, ops.CLTensor):
            if (k[0] == 1 and s[0] == 1 and p[0] == 0 and d == (1, 1, 1) and not self.return_indices and not self.ceil_mode and k[1] * k[2] <= 255
                    and p[1] < k[1] and p[2] < k[2] and x.C % 4 == 0):
                return ops.pool_cl(x, k[1:], s[1:], p[1:])
            x = x.planes()
        if (x.dim() == 5 and x.is_cuda and x.dtype == torch.float32 and k[0] == 1 and s[0] == 1 and p[0] == 0 and d == (1, 1, 1)
                and not self.return_indices and not self.ceil_mode and k[1] * k[2] <= 255):
            return ops.pool_planes(x, k[1:], s[1:], p[1:])
        ops.stock_fallback("models.backbone.SpatialMaxPool3d", "a window over frames, ceil_mode / return_indices, or a CPU / non-fp32 input")
        return super().forward(x)


def _norm_relu3d(kind, channels):
    """[normalisation, ReLU] of a stem layer (reference backbone.py:77-79 etc.: BatchNorm3d / GroupNorm(32) then ReLU(True))"""
    if kind == 'bn':
        return [BatchNorm3dReLU(channels), nn.Identity()]
    return [nn.GroupNorm(32, channels), nn.ReLU(True)]


def _vgg_group(idx, norm):
    """Layer group `conv{idx}` of the VGG-M style stem (reference backbone.py:73-103,179-184,243-271)."""
    chain = norm == 'bn'                  # BatchNorm3d + ReLU have a channels-last kernel; GroupNorm stems stay on planes
    if idx == 1:
        c1 = Conv3d(3, 64, 3, stride=(1, 2, 2), padding=(1, 0, 0))
        c1.cl_chain = chain
        return [c1] + _norm_relu3d(norm, 64) + [SpatialMaxPool3d(kernel_size=(1, 2, 2), stride=(1, 2, 2))]
    cin, cout, pool = {2: (64, 128, True), 3: (128, 256, True), 4: (256, 512, False), 5: (512, 512, False)}[idx]
    mods = [Conv3d(cin, cout, 3, 1, padding=(1, 0, 0))] + _norm_relu3d(norm, cout)
    if pool:
        mods.append(SpatialMaxPool3d(kernel_size=(1, 2, 2), stride=(1, 2, 2)))
    return mods


def _init_like_reference(net):
    """reference `_initialize_weights` (backbone.py:147-161): Conv3d ~ N(0, sqrt(2/(k^3*C_out))), zero bias;
    Conv1d kaiming-normal (a no-op draw for weight-normed convs, whose `.weight` is recomputed);
    BatchNorm3d/1d -> (1, 0)."""
    for m in net.modules():
        if isinstance(m, nn.Conv3d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2.0 / n))
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, WeightNormConv1d):
            nn.init.kaiming_normal_(torch.empty_like(m.weight_v))
        elif isinstance(m, nn.Conv1d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, (nn.BatchNorm3d, nn.BatchNorm1d)):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


def _simple_tcn(in_dim, hidden, k, pad):
    return nn.Sequential(nn.Conv1d(in_dim, hidden, k, 1, pad), nn.BatchNorm1d(512), nn.ReLU(True),
                         nn.Conv1d(hidden, hidden, k, 1, pad), nn.BatchNorm1d(512), nn.ReLU(True))


# M3T_TOWERS_CONCURRENT=0: the two private towers of VA_3DVGGM_Split one after the other on the caller's stream (A/B; default: concurrent)
_TOWERS_CONCURRENT = [__import__("os").environ.get("M3T_TOWERS_CONCURRENT", "1") != "0"]


def _squeeze_hw(x):
    """[B,C,T,1,1] -> [B,C,T] (the reference's bare .squeeze() also drops B or T when they are 1)."""
    if isinstance(x, ops.CLTensor):          # the end of a channels-last stem: rows (n, t) x C
        if x.H == 1 and x.W == 1:
            return ops.btc_to_bct(x.data.view(x.N, x.T, x.C))
        x = x.planes()
    return x.flatten(3).squeeze(-1)


class VA_3DVGGM(nn.Module):
    def __init__(self, inputDim=512, hiddenDim=512, nLayers=2, nClasses=2, frameLen=16, backend='gru', norm_layer='bn',
                 nFCs=1):
        super().__init__()
        self.inputDim, self.hiddenDim, self.nClasses = inputDim, hiddenDim, nClasses
        self.frameLen, self.nLayers, self.backend, self.nFCs = frameLen, nLayers, backend, nFCs
        stem = []
        for g in range(1, 6):
            stem += _vgg_group(g, norm_layer)
        self.v2p = nn.Sequential(*stem)
        if backend == 'gru':
            self.gru = GRU(inputDim, hiddenDim, nLayers, nClasses, nFCs)
        elif backend == 'tcn':
            self.tcn = nn.ModuleList([TemporalConvNet(inputDim, [hiddenDim] * nLayers, 3), nn.Linear(hiddenDim, 2)])
        elif backend == 'tcn_simple':
            self.tcn = nn.ModuleList([_simple_tcn(inputDim, hiddenDim, 3, 1), nn.Linear(hiddenDim, 2)])
        elif backend == 'fc':
            self.fc = nn.Sequential(nn.Linear(512, hiddenDim), nn.ReLU(True), nn.Linear(hiddenDim, nClasses))
        _init_like_reference(self)

    def temporal(self, feats):
        """feats channel-first [B,512,T] -> predictions (the part after the conv stem, backbone.py:136-145)."""
        if self.backend == 'gru':
            return self.gru(ops.bct_to_btc(feats))
        if self.backend == 'tcn':
            h = self.tcn[0].forward_btc(ops.bct_to_btc(feats))         # stays channel-last: no transpose back
            return ops.linear(h, self.tcn[1].weight, self.tcn[1].bias, 0)
        if self.backend == 'tcn_simple':
            h = ops.simple_tcn(ops.bct_to_btc(feats), self.tcn[0])      # Conv1d/BN1d/ReLU x2 on the HIP kernels
            return ops.linear(h, self.tcn[1].weight, self.tcn[1].bias, 0)
        if self.backend == 'fc':
            h = ops.bct_to_btc(feats)
            h = ops.linear(h, self.fc[0].weight, self.fc[0].bias, 1)
            return ops.linear(h, self.fc[2].weight, self.fc[2].bias, 0).mean(dim=1)
        return feats

    def forward(self, x):
        with ops.batch_counters():          # (every BatchNorm's num_batches_tracked in one launch)
            return self.temporal(_squeeze_hw(self.v2p(x)))


class VA_3DVGGM_Split(nn.Module):
    def __init__(self, inputDim=512, hiddenDim=512, nLayers=2, frameLen=16, nClasses=2, backend='gru', norm_layer='bn',
                 split_layer=5, nFCs=1, use_mtl=False):
        super().__init__()
        self.inputDim, self.hiddenDim, self.frameLen, self.nLayers = inputDim, hiddenDim, frameLen, nLayers
        self.nClasses, self.backend, self.split_layer = nClasses, backend, split_layer
        self.norm_layer, self.nFCs, self.use_mtl = norm_layer, nFCs, use_mtl
        assert split_layer >= 2, 'degenerate multi-tower structure'
        shared, v_priv, a_priv = _vgg_group(1, norm_layer), [], []
        for g in range(2, 6):
            if split_layer >= g:
                shared += _vgg_group(g, norm_layer)
            else:                      # construction order V then A per group, as the reference (RNG order)
                v_priv += _vgg_group(g, norm_layer)
                a_priv += _vgg_group(g, norm_layer)
        self.shared = nn.Sequential(*shared)
        if split_layer != 5:
            self.v_private = nn.Sequential(*v_priv)
            self.a_private = nn.Sequential(*a_priv)
        if backend == 'gru':
            if split_layer == 5:
                self.gru = GRU(inputDim + 512 + 512, hiddenDim, nLayers, nClasses, nFCs)
            else:
                # mtl: V tower -> 7 expr logits + valence (nClasses-1), A tower -> arousal (1); -1/-2 => no FC
                self.gru_v = GRU(inputDim + 512, hiddenDim, nLayers, nClasses - 1, nFCs)
                self.gru_a = GRU(inputDim + 512, hiddenDim, nLayers, min(nClasses, 1), nFCs)
        elif backend == 'tcn_simple' and split_layer != 5:
            self.tcn_v = nn.ModuleList([_simple_tcn(inputDim + 512, hiddenDim, 5, 2)])
            self.tcn_a = nn.ModuleList([_simple_tcn(inputDim + 512, hiddenDim, 5, 2)])
            if use_mtl:
                if nClasses > 0:
                    self.tcn_v.append(nn.Linear(hiddenDim, nClasses - 1))
                    self.tcn_a.append(nn.Linear(hiddenDim, 1))
            else:
                self.tcn_v.append(nn.Linear(hiddenDim, 1))
                self.tcn_a.append(nn.Linear(hiddenDim, 1))
        _init_like_reference(self)

    def features(self, x, se, au):
        """Conv towers + feature concat -> the two channel-last [B,T,1024] GRU inputs (split_layer != 5)."""
        x = self.shared(x)
        if _TOWERS_CONCURRENT[0] and isinstance(x, ops.CLTensor):
            # round 6: the two private towers are independent and their deep layers are small grids (conv4: 144 tiles, conv5: 16) -- the
            # arousal tower runs on the side stream beside the valence tower; autograd runs each tower's backward on its forward's stream
            # (C5 13.21 -> 12.97 ms, inference 3.53 -> 3.45 ms; NOTEBOOK R6.9)
            main, side = ops.cur_stream(x.data.device), ops.side_stream(x.data.device)
            xd = x.data
            side.wait_stream(main)
            xd.record_stream(side)
            with ops.on_stream(side):
                f_a = _squeeze_hw(self.a_private(x))
            f_v = _squeeze_hw(self.v_private(x))
            main.wait_stream(side)
            f_a.record_stream(main)
        else:
            f_v, f_a = _squeeze_hw(self.v_private(x)), _squeeze_hw(self.a_private(x))
        x_v = torch.cat((f_v, se), dim=1)    # valence tower | SENet feats
        x_a = torch.cat((f_a, au), dim=1)    # arousal tower | (TCAE-AU or SENet) feats
        return x_v, x_a

    def forward(self, x, se, au):
        with ops.batch_counters():          # (every BatchNorm's num_batches_tracked in one launch)
            return self._forward(x, se, au)

    def _forward(self, x, se, au):
        if self.split_layer != 5:
            x_v, x_a = self.features(x, se, au)
            if self.backend == 'gru':
                y_v, y_a = run_grus([self.gru_v, self.gru_a], [ops.bct_to_btc(x_v), ops.bct_to_btc(x_a)])
                return torch.cat((y_v, y_a), dim=-1)
            if self.backend.startswith('tcn'):
                h_v = ops.simple_tcn(ops.bct_to_btc(x_v), self.tcn_v[0])
                h_a = ops.simple_tcn(ops.bct_to_btc(x_a), self.tcn_a[0])
                y_v = ops.linear(h_v, self.tcn_v[1].weight, self.tcn_v[1].bias, 0)
                y_a = ops.linear(h_a, self.tcn_a[1].weight, self.tcn_a[1].bias, 0)
                return torch.cat((y_v, y_a), dim=-1)
            return None
        x = torch.cat((_squeeze_hw(self.shared(x)), se, au), dim=1)
        if self.backend == 'gru':
            x = self.gru(ops.bct_to_btc(x))
        return x


class VA_3DResNet(nn.Module):
    def __init__(self, inputDim=512, hiddenDim=512, nLayers=2, nClasses=2, frameLen=16, backend='gru', use_cbam=False,
                 resnet_ver='v2', resnet_depth=18, frontend_agg_mode='ap', nFCs=1):
        super().__init__()
        self.inputDim, self.hiddenDim, self.nClasses = inputDim, hiddenDim, nClasses
        self.frameLen, self.nLayers, self.backend, self.nFCs = frameLen, nLayers, backend, nFCs
        self.c3d = nn.Sequential(
            Conv3d(3, 64, kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            BatchNorm3dReLU(64), nn.Identity(),
            SpatialMaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)))
        assert resnet_depth in [18, 34] and resnet_ver in ['v1', 'v2'], \
            'unsupported ResNet configuration: {}, {}'.format(resnet_depth, resnet_ver)
        cfg = [2, 2, 2, 2] if resnet_depth == 18 else [3, 4, 6, 3]
        if resnet_ver == 'v2':
            self.resnet = ResNetV2(BasicBlockV2, cfg, inputDim, zero_init_residual=False, agg_mode=frontend_agg_mode,
                                   fmap_out_size=3, use_cbam=use_cbam)
        else:
            self.resnet = ResNet(BasicBlock, cfg, inputDim, zero_init_residual=True, agg_mode=frontend_agg_mode,
                                 fmap_out_size=3, use_cbam=use_cbam)
        if backend == 'gru':
            self.gru = GRU(inputDim, hiddenDim, nLayers, nClasses, nFCs)
        _init_like_reference(self)

    def forward(self, x):
        with ops.batch_counters():          # (every BatchNorm's num_batches_tracked -- 20 in the ResNet, 8 in its CBAM gates, the stem's -- in one launch)
            return self._forward(x)

    def _forward(self, x):
        x = self.c3d(x)                                            # [B,64,T,h,w]
        x = x.transpose(1, 2).reshape(-1, 64, x.size(3), x.size(4))  # fold T into the batch: per-frame ResNet
        x = self.resnet(x).view(-1, self.frameLen, self.inputDim)
        if self.backend == 'gru':
            x = self.gru(x)
        return x
