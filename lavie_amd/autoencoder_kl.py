"""`AutoencoderKL` in stock PyTorch-ROCm ops — outside the latents/s metric, no HIP kernels of ours.

The reference takes the class from diffusers 0.16.0 (base/pipelines/sample.py, interpolation/sample.py:230) and vendors only
its outer shell (`vsr/models/autoencoder_kl.py:46-334`: encoder -> quant_conv -> DiagonalGaussianDistribution;
post_quant_conv -> decoder; slicing).  The shell below follows that vendored text; `Encoder` / `Decoder` / the mid-block
attention live in diffusers internals that are not in the tree, so they are restated from the published Stable Diffusion
VAE architecture with diffusers 0.16's state-dict names (resnets: norm1/conv1/norm2/conv2/conv_shortcut; attention:
group_norm/query/key/value/proj_attn; downsamplers.0.conv / upsamplers.0.conv).  PARITY UNPINNED: no source, fixture or
test for these internals exists inside the reference."""
from types import SimpleNamespace
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import nn


class ResnetBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int, groups: int):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class AttentionBlock(nn.Module):
    """single-head spatial self-attention of the VAE mid block (diffusers 0.16 `AttentionBlock`)."""

    def __init__(self, c: int, groups: int):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.query, self.key, self.value, self.proj_attn = (nn.Linear(c, c) for _ in range(4))

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).reshape(b, c, h * w).transpose(1, 2)
        o = F.scaled_dot_product_attention(self.query(t)[:, None], self.key(t)[:, None], self.value(t)[:, None])[:, 0]
        return x + self.proj_attn(o).transpose(1, 2).reshape(b, c, h, w)


class _Sampler(nn.Module):
    def __init__(self, c: int, down: bool):
        super().__init__()
        self.down = down
        self.conv = nn.Conv2d(c, c, 3, stride=2 if down else 1, padding=0 if down else 1)

    def forward(self, x):
        if self.down:
            return self.conv(F.pad(x, (0, 1, 0, 1)))                   # asymmetric padding of the SD encoder
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Block(nn.Module):
    def __init__(self, cin: int, cout: int, layers: int, groups: int, sampler: str):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        if sampler == "down":
            self.downsamplers = nn.ModuleList([_Sampler(cout, True)])
        elif sampler == "up":
            self.upsamplers = nn.ModuleList([_Sampler(cout, False)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        for s in list(getattr(self, "downsamplers", [])) + list(getattr(self, "upsamplers", [])):
            x = s(x)
        return x


class _Mid(nn.Module):
    def __init__(self, c: int, groups: int):
        super().__init__()
        self.attentions = nn.ModuleList([AttentionBlock(c, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, groups), ResnetBlock2D(c, c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, cin, latent, widths, layers, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(cin, widths[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([
            _Block(widths[max(i - 1, 0)], w, layers, groups, "down" if i + 1 < len(widths) else "") for i, w in enumerate(widths)])
        self.mid_block = _Mid(widths[-1], groups)
        self.conv_norm_out = nn.GroupNorm(groups, widths[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(widths[-1], 2 * latent, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class Decoder(nn.Module):
    def __init__(self, latent, cout, widths, layers, groups):
        super().__init__()
        rev = list(reversed(widths))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0], groups)
        self.up_blocks = nn.ModuleList([
            _Block(rev[max(i - 1, 0)], w, layers + 1, groups, "up" if i + 1 < len(rev) else "") for i, w in enumerate(rev)])
        self.conv_norm_out = nn.GroupNorm(groups, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], cout, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DiagonalGaussianDistribution:
    def __init__(self, moments: torch.Tensor):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKL(nn.Module):
    """vsr/models/autoencoder_kl.py:77-203 (constructor keywords, encode / decode surface, slicing)."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, block_out_channels: Tuple[int, ...] = (128, 256, 512, 512),
                 layers_per_block: int = 2, latent_channels: int = 4, norm_num_groups: int = 32, sample_size: int = 512,
                 scaling_factor: float = 0.18215, **_unused):
        super().__init__()
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.use_slicing = False
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                                      block_out_channels=tuple(block_out_channels), scaling_factor=scaling_factor,
                                      sample_size=sample_size)

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    @torch.no_grad()
    def encode(self, x: torch.Tensor):
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(self.quant_conv(self.encoder(x))))

    @torch.no_grad()
    def decode(self, z: torch.Tensor):
        if self.use_slicing and z.shape[0] > 1:
            return SimpleNamespace(sample=torch.cat([self.decoder(self.post_quant_conv(s)) for s in z.split(1)]))
        return SimpleNamespace(sample=self.decoder(self.post_quant_conv(z)))
