"""models.resnet -- per-frame ResNet-18/34 feature extractor with optional CBAM gating.

API/state-dict compatible with the reference's models/resnet.py:18-124 (`BasicBlock`,
`ResNet`, v1) and :127-251 (`BasicBlockV2`, `ResNetV2`).  Every CBAM gate inside the blocks (models.cbam), since round 4
BatchNorm2d with the ReLU behind it (PlaneBatchNorm2d) and since round 5 the dense 2-D convolutions (GemmConv2d: forward with any
stride, weight gradient and the data gradient as tap-walk implicit GEMMs over channels-last rows on the fp16x3 kernels, no patch
matrix) run in the HIP library, in every grad mode (round 6: validation / test steps and frozen encoders take the same walks).
"""
import torch
import torch.nn as nn

from m3t import ops
from .cbam import CBAM


class PlaneBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers, state_dict keys) on the HIP channel-plane kernels (m3t.ops.bn_planes, csrc/bn.hip: the
    per-frame ResNet's maps are 32 768 ... 262 144 planes of 784 ... 16 floats), with the ReLU that follows it fused in when
    `fuse_relu` (no pass over the activation of its own, forward or backward).  Other inputs (CPU, other dtypes) take the stock op."""

    def __init__(self, num_features, fuse_relu=False, **kw):
        super().__init__(num_features, **kw)
        self.fuse_relu = fuse_relu

    def forward(self, x):
        if (ops.BN_PLANES[0] and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and self.affine and self.track_running_stats
                and self.momentum is not None):
            if self.training:
                ops.count_batch(self.num_batches_tracked)
            return ops.bn_planes(x, self.weight, self.bias, self.running_mean, self.running_var, self.training, self.momentum,
                                 self.eps, self.fuse_relu)
        ops.stock_fallback("models.resnet.PlaneBatchNorm2d", "M3T_BN_PLANES=0, CPU / non-fp32 input or a configuration without running statistics")
        y = super().forward(x)
        return torch.relu(y) if self.fuse_relu else y


class GemmConv2d(nn.Conv2d):
    """nn.Conv2d (same parameters / state_dict keys) of the per-frame ResNet on m3t.ops.conv3d with a unit time axis (round 5): forward (any
    stride), weight gradient and data gradient (round 6: the strided layers' as parity-class walks, stride on the source side) as tap-walk
    implicit GEMMs over channels-last activations on the fp16x3 kernels (the per-frame maps are 512 frames x 28^2 ... 4^2 positions, K =
    9 C_in; ragged last row tile; no patch matrix, no MIOpen kernel).  Other inputs (CPU, other dtypes, groups / dilation,
    M3T_CONV3D_MIOPEN=1) take the stock op and say so once on stderr."""

    def forward(self, x):
        # one path whatever the grad mode (validation / test steps, frozen encoders: see models.backbone.Conv3d)
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and self.groups == 1 and tuple(self.dilation) == (1, 1)
                and self.padding_mode == "zeros" and isinstance(self.padding, tuple)):
            return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding)
        ops.stock_fallback("models.resnet.GemmConv2d", "CPU / non-fp32 input, groups, dilation or a padding mode the walks do not cover")
        return super().forward(x)


def conv3x3(in_planes, out_planes, stride=1):
    return GemmConv2d(in_planes, out_planes, 3, stride, 1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return GemmConv2d(in_planes, out_planes, 1, stride, bias=False)


class BasicBlock(nn.Module):
    """conv-bn-relu-conv-bn -> [CBAM] -> + identity -> relu (CBAM sits BEFORE the residual add,
    reference resnet.py:50-53)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, use_cbam=False):
        super().__init__()
        self.conv1, self.bn1 = conv3x3(inplanes, planes, stride), PlaneBatchNorm2d(planes, fuse_relu=True)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = conv3x3(planes, planes), PlaneBatchNorm2d(planes)
        self.downsample, self.stride = downsample, stride
        self.cbam = CBAM(planes) if use_cbam else None

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.bn2(self.conv2(self.bn1(self.conv1(x))))            # (bn1 applies the ReLU)
        if self.cbam is not None:
            y = self.cbam(y)
        return ops.add_relu(y, shortcut)                             # (one pass: m3t_add_relu)


class BasicBlockV2(nn.Module):
    """Pre-activation block (reference resnet.py:127-173)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, is_first_block_of_first_layer=False,
                 use_cbam=False):
        super().__init__()
        self.is_first_block_of_first_layer = is_first_block_of_first_layer
        if not is_first_block_of_first_layer:
            self.bn1 = PlaneBatchNorm2d(inplanes, fuse_relu=True)
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn2 = PlaneBatchNorm2d(planes, fuse_relu=True)
        self.conv2 = conv3x3(planes, planes)
        self.relu = nn.ReLU(True)
        self.downsample, self.stride = downsample, stride
        self.cbam = CBAM(planes) if use_cbam else None

    def forward(self, x):
        pre = x if self.is_first_block_of_first_layer else self.bn1(x)      # (bn1 / bn2 apply their ReLU)
        shortcut = x if self.downsample is None else self.downsample(pre)
        y = self.conv2(self.bn2(self.conv1(pre)))
        if self.cbam is not None:
            y = self.cbam(y)
        return y + shortcut


def _init_trunk(net):
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    # CBAM: the spatial gate's BN gamma starts at 0 => scale = sigmoid(0) = 0.5 (reference resnet.py:81-85)
    for name, p in net.named_parameters():
        if name.endswith("weight") and "bn" in name and "SpatialGate" in name:
            nn.init.constant_(p, 0)


class _Trunk(nn.Module):
    widths = (64, 128, 256, 512)

    def _stages(self, block, layers, use_cbam):
        self.inplanes = 64
        for i, (w, n) in enumerate(zip(self.widths, layers)):
            setattr(self, "layer%d" % (i + 1), self._make_layer(block, w, n, stride=1 if i == 0 else 2, use_cbam=use_cbam))

    def _run_stages(self, x):
        for i in range(4):
            x = getattr(self, "layer%d" % (i + 1))(x)
        return x


class ResNet(_Trunk):
    def __init__(self, block, layers, num_classes=256, zero_init_residual=True, agg_mode="ap", fmap_out_size=3,
                 use_cbam=False):
        super().__init__()
        self.agg_mode = agg_mode
        self._stages(block, layers, use_cbam)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * fmap_out_size * fmap_out_size, num_classes)   # unused in 'ap' mode (as reference)
        _init_trunk(self)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, use_cbam=False):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                 PlaneBatchNorm2d(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, down, use_cbam=use_cbam)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes, use_cbam=use_cbam) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self._run_stages(x)
        if self.agg_mode == "ap":
            x = self.avgpool(x).flatten(1)
        if self.agg_mode == "fc":
            x = self.fc(x.flatten(1))
        return x


class ResNetV2(_Trunk):
    def __init__(self, block, layers, num_classes=256, zero_init_residual=False, agg_mode="ap", fmap_out_size=3,
                 use_cbam=False):
        super().__init__()
        self.agg_mode = agg_mode
        self._stages(block, layers, use_cbam)
        self.bn5 = PlaneBatchNorm2d(self.inplanes, fuse_relu=True)
        self.relu5 = nn.Identity()                                     # (bn5 applies the ReLU; the attribute stays for the module tree)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * fmap_out_size * fmap_out_size, num_classes)
        _init_trunk(self)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlockV2):
                    nn.init.constant_(m.bn1.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, use_cbam=False):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = GemmConv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False)
        seq = [block(self.inplanes, planes, stride, down, stride == 1, use_cbam=use_cbam)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes, use_cbam=use_cbam) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.relu5(self.bn5(self._run_stages(x)))
        x = self.avgpool(x).flatten(1)       # the reference pools (twice in 'ap' mode: idempotent) then flattens
        if self.agg_mode == "fc":
            x = self.fc(x)
        return x
