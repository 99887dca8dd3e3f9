"""models.att_fusion -- attention fusion of audio and video features, MI355X-native.

Drop-in for the reference's models/att_fusion.py:8-27: `AttFusion(input_dim, hidden_dim)`,
`forward(x_a, x_v)` with ARGUMENT ORDER (audio, video) and softmax index 0 = video;
parameters `[proj_v.*,] scorer_{a,v}.gru.*`, `scorer_{a,v}.fc.{weight,bias}`.
Both scorer BiGRUs advance in one grouped scan; sigmoid -> 2-way softmax -> weighted sum
is one fused HIP kernel (m3t_att_fuse_fwd/bwd).
"""
import torch.nn as nn

from m3t import ops
from .rnn import GRU, run_grus


class AttFusion(nn.Module):
    def __init__(self, input_dim=[512, 512], hidden_dim=128):
        super().__init__()
        d_audio, d_video = input_dim[0], input_dim[1]
        self.use_proj = d_video != d_audio
        if self.use_proj:
            self.proj_v = nn.Linear(d_video, d_audio)          # parameter holder; GEMM runs in HIP
        self.scorer_a = GRU(d_audio, hidden_dim, 1, 1, 1)
        self.scorer_v = GRU(d_audio, hidden_dim, 1, 1, 1)

    def forward(self, x_a, x_v):
        if self.use_proj:
            x_v = ops.linear(x_v, self.proj_v.weight, self.proj_v.bias, 0)
        s_v, s_a = run_grus([self.scorer_v, self.scorer_a], [x_v, x_a])     # raw scores [B,T,1]
        return ops.att_fuse(s_v, s_a, x_v, x_a)
