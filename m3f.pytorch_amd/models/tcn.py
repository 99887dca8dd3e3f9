"""models.tcn -- weight-normed dilated causal TCN, MI355X-native.

Drop-in for the reference's models/tcn.py:7-64 (`Chomp1d`, `TemporalBlock`,
`TemporalConvNet`): same constructors, channel-first [B,C,T] in/out, same parameter
names (`conv{1,2}.{bias,weight_g,weight_v}`, `downsample.{weight,bias}`) INCLUDING the
`net.0.* / net.4.*` aliases the reference's state_dict carries (tcn.py:31-32).

Inside, activations are channel-last [B,T,C]; each block is one autograd Function
(m3t.ops.temporal_block): weight-norm kernel -> implicit-GEMM dilated causal conv on fp32
MFMA with the temporal halo staged in LDS and bias+ReLU(+residual+ReLU) fused in the
epilogue.  The conv left-pads only, so the reference's Chomp1d copy never happens.
"""
import math

import torch
import torch.nn as nn

from m3t import ops


class Chomp1d(nn.Module):
    """Kept for API/state-dict parity (reference tcn.py:7-13).  The fused conv never calls it."""

    def __init__(self, chomp_size):
        super().__init__()
        self.chomp_size = chomp_size

    def forward(self, x):
        return x[:, :, :x.size(2) - self.chomp_size].contiguous()


class WeightNormConv1d(nn.Module):
    """Parameter holder equivalent to weight_norm(nn.Conv1d(...)) (reference tcn.py:19-20):
    parameters `bias`, `weight_g` [Co,1,1], `weight_v` [Co,Ci,k] in that registration order,
    initialised like nn.Conv1d followed by weight_norm (g = ||v||)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, stride, padding, dilation
        v = torch.empty(out_channels, in_channels, kernel_size)
        nn.init.kaiming_uniform_(v, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_channels * kernel_size)
        self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        self.weight_g = nn.Parameter(v.norm(2, dim=(1, 2), keepdim=True))
        self.weight_v = nn.Parameter(v)

    @property
    def weight(self):
        """Effective weight g * v / ||v|| (host-side view for inspection only)."""
        return self.weight_g * self.weight_v / self.weight_v.norm(2, dim=(1, 2), keepdim=True)


class TemporalBlock(nn.Module):
    def __init__(self, n_inputs, n_outputs, kernel_size, stride, dilation, padding, dropout=0.2):
        super().__init__()
        if stride != 1 or padding != (kernel_size - 1) * dilation:
            raise ValueError("TemporalBlock: the causal kernel needs stride 1 and padding (k-1)*dilation")
        self.dilation = dilation
        self.p_drop = dropout
        self.conv1 = WeightNormConv1d(n_inputs, n_outputs, kernel_size, stride, padding, dilation)
        self.chomp1 = Chomp1d(padding)
        self.relu1 = nn.ReLU()
        self.dropout1 = nn.Dropout(dropout)
        self.conv2 = WeightNormConv1d(n_outputs, n_outputs, kernel_size, stride, padding, dilation)
        self.chomp2 = Chomp1d(padding)
        self.relu2 = nn.ReLU()
        self.dropout2 = nn.Dropout(dropout)
        # structural alias: gives the state_dict its net.0.* / net.4.* duplicate keys
        self.net = nn.Sequential(self.conv1, self.chomp1, self.relu1, self.dropout1,
                                 self.conv2, self.chomp2, self.relu2, self.dropout2)
        self.downsample = nn.Conv1d(n_inputs, n_outputs, 1) if n_inputs != n_outputs else None
        self.relu = nn.ReLU()
        self.init_weights()

    def init_weights(self):
        # reference tcn.py:37-41 writes conv{1,2}.weight.data, which the weight-norm pre-hook
        # overwrites: an effective no-op that still consumes RNG.  Reproduce the draws, keep the no-op.
        torch.empty_like(self.conv1.weight_v).normal_(0, 0.01)
        torch.empty_like(self.conv2.weight_v).normal_(0, 0.01)
        if self.downsample is not None:
            self.downsample.weight.data.normal_(0, 0.01)

    def forward_btc(self, x, seeds=None):
        """x channel-last [B,T,C_in] -> [B,T,C_out].  Train mode: the two nn.Dropout(p) of the block (reference tcn.py:23,29)
        are masks generated INSIDE the conv epilogues from two 64-bit seeds (Philox4x32-10 keyed by seed, counter = element;
        the backward pass regenerates them: no mask tensor in HBM).  The seeds come from torch's CPU generator, so
        torch.manual_seed makes a run reproducible; `seeds` overrides them (tests)."""
        drop_p = self.p_drop if (self.training and self.p_drop > 0) else 0.0
        if drop_p > 0 and seeds is None:
            seeds = torch.empty(2, dtype=torch.int64).random_().tolist()
        wd = bd = None
        if self.downsample is not None:
            wd, bd = self.downsample.weight, self.downsample.bias
        return ops.temporal_block(x, self.conv1.weight_v, self.conv1.weight_g, self.conv1.bias,
                                  self.conv2.weight_v, self.conv2.weight_g, self.conv2.bias,
                                  wd, bd, self.dilation, None, None, drop_p, seeds or (0, 0))

    def forward(self, x):
        return ops.btc_to_bct(self.forward_btc(ops.bct_to_btc(x)))


class TemporalConvNet(nn.Module):
    def __init__(self, num_inputs, num_channels, kernel_size=2, dropout=0.2):
        super().__init__()
        blocks = []
        for level, width in enumerate(num_channels):
            dil = 2 ** level
            c_in = num_inputs if level == 0 else num_channels[level - 1]
            blocks.append(TemporalBlock(c_in, width, kernel_size, stride=1, dilation=dil,
                                        padding=(kernel_size - 1) * dil, dropout=dropout))
        self.network = nn.Sequential(*blocks)

    def forward_btc(self, x):
        for blk in self.network:
            x = blk.forward_btc(x)
        return x

    def forward(self, x):
        # one layout change on the way in, one on the way out; blocks stay channel-last
        return ops.btc_to_bct(self.forward_btc(ops.bct_to_btc(x)))
