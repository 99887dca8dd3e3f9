"""models.cbam -- per-frame SE/CBAM gating, MI355X-native.

Drop-in for the reference's models/cbam.py:10-111 (`Flatten`, `BasicConv`, `ChannelGate`,
`ChannelPool`, `SpatialGate`, `CBAM`): same constructors and parameter names
(`ChannelGate.mlp.{1,3}.*`, `SpatialGate.spatial.conv.weight`, `SpatialGate.spatial.bn.*`).
The gates run as HIP kernels (m3t_cbam_channel_*, m3t_cbam_spatial_*): wavefront-shuffle
squeeze per (frame, channel) plane, shared MLP out of LDS, and the spatial gate's
5x5 conv + BatchNorm2d(1) + sigmoid + rescale chain.
"""
import torch
import torch.nn as nn

from m3t import ops


class Flatten(nn.Module):
    def forward(self, x):
        return x.reshape(x.size(0), -1)


class BasicConv(nn.Module):
    """conv -> [bn] -> [relu] holder (reference cbam.py:15-30).  Only the SpatialGate instance
    (2->1, 5x5, bn, no relu) is on the hot path; it is executed by the HIP spatial-gate kernels."""

    def __init__(self, in_planes, out_planes, kernel_size, stride=1, padding=0, dilation=1, groups=1, relu=True,
                 bn=True, bias=False):
        super().__init__()
        self.out_channels = out_planes
        self.conv = nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_planes, eps=1e-5, momentum=0.01, affine=True) if bn else None
        self.relu = nn.ReLU() if relu else None

    def forward(self, x):      # generic composition (not used by SpatialGate.forward)
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        if self.relu is not None:
            x = self.relu(x)
        return x


class ChannelGate(nn.Module):
    def __init__(self, gate_channels, reduction_ratio=16):
        super().__init__()
        self.gate_channels = gate_channels
        self.mlp = nn.Sequential(
            Flatten(),
            nn.Linear(gate_channels, gate_channels // reduction_ratio),
            nn.ReLU(),
            nn.Linear(gate_channels // reduction_ratio, gate_channels),
        )

    def forward(self, x):
        return ops.channel_gate(x, self.mlp[1].weight, self.mlp[1].bias, self.mlp[3].weight, self.mlp[3].bias)


class ChannelPool(nn.Module):
    """(max over C, mean over C) stacked on dim 1 -- order (max, mean) as reference cbam.py:67-71.
    Host-side utility; SpatialGate.forward fuses it into the HIP kernel."""

    def forward(self, x):
        return torch.cat((x.max(1, keepdim=True)[0], x.mean(1, keepdim=True)), dim=1)


class SpatialGate(nn.Module):
    def __init__(self):
        super().__init__()
        k = 5
        self.compress = ChannelPool()
        self.spatial = BasicConv(2, 1, k, stride=1, padding=(k - 1) // 2, relu=False)

    def forward(self, x):
        bn = self.spatial.bn
        y = ops.spatial_gate(x, self.spatial.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                             self.training, bn.momentum, bn.eps)
        if self.training:
            ops.count_batch(bn.num_batches_tracked)
        return y


class CBAM(nn.Module):
    def __init__(self, gate_channels, reduction_ratio=16):
        super().__init__()
        self.ChannelGate = ChannelGate(gate_channels, reduction_ratio)
        self.SpatialGate = SpatialGate()

    def forward(self, x):
        """SpatialGate(ChannelGate(x)) (reference cbam.py:108-111) -- as ONE fused operator where the shape allows (every
        map of the ResNet stages: the intermediate x * channel_scale is never written, 8 instead of 10+ HBM passes for
        forward + backward), else the two gates one after the other.  Same parameters, same state_dict, same results."""
        cg, sg = self.ChannelGate, self.SpatialGate
        if ops.cbam_fused_ok(x, cg.mlp[1].weight.shape[0]):
            bn = sg.spatial.bn
            y = ops.cbam(x, cg.mlp[1].weight, cg.mlp[1].bias, cg.mlp[3].weight, cg.mlp[3].bias, sg.spatial.conv.weight,
                         bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training, bn.momentum, bn.eps)
            if self.training:
                ops.count_batch(bn.num_batches_tracked)
            return y
        return sg(cg(x))
