"""Timesteps / TimestepEmbedding restated (parity unpinned against diffusers 0.16).

In-tree textual spec of the sinusoid: /root/reference/base/models/utils.py:74-94
(freqs = exp(-ln(1e4) * k / half), args = t * freqs); diffusers orders [sin, cos] and
`flip_sin_to_cos=True` (unet.py:108,153) swaps to [cos, sin]."""
import math

import torch
from torch import nn


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip = flip_sin_to_cos
        self.shift = downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        expo = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
        expo = expo / (half - self.shift)
        ang = timesteps[:, None].float() * torch.exp(expo)[None, :]
        s, c = torch.sin(ang), torch.cos(ang)
        return torch.cat([c, s], dim=-1) if self.flip else torch.cat([s, c], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


class ImagePositionalEmbeddings(nn.Module):   # symbol only: vsr/models/diffusers_attention.py imports it, the VSR path never builds it
    def __init__(self, *a, **k):
        raise NotImplementedError
