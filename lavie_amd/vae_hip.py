"""VAE decoder on the engine's operators: the 3x3 convolutions, GroupNorm + SiLU, nearest-x2 upsampling and
the linear projections of `AutoencoderKL.decode` run through the C-ABI kernels of liblavie_hip.so (`lavie_conv3x3_f16` with the
fused shortcut / folded upsample, `lavie_group_norm_f16`, `lavie_linear_f16`) on channels-last fp16 rows; only the 4-channel
input convolutions, the 3-channel output convolution and the mid block's single-head attention (head dim 512 > the kernel's
160) stay on stock PyTorch ops.

Why: in the full cascade (tools/bench_cascade.py) the stock fp32 decode of 61 frames at 1280x2048 took 321 s of 525 s.
`HipAutoencoderKL(vae)` wraps a `lavie_amd.autoencoder_kl.AutoencoderKL` (same weights, same `decode(z).sample` /
`encode` / `config` surface) and can be passed wherever the pipelines take a `vae`.  Parity: against the wrapped stock module
(tests/test_gpu_cascade.py); the stock module itself is parity-unpinned (see its header)."""
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import ops


def _rows(x: torch.Tensor) -> torch.Tensor:
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous()


class _Res:
    def __init__(self, r, dev):
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        h16 = lambda t: t.detach().to(dev, torch.float16).contiguous()
        self.cin, self.cout = r.conv1.in_channels, r.conv1.out_channels
        self.g1, self.b1, self.g2, self.b2 = f32(r.norm1.weight), f32(r.norm1.bias), f32(r.norm2.weight), f32(r.norm2.bias)
        self.w1, self.c1 = ops.pack_conv3x3(h16(r.conv1.weight)), f32(r.conv1.bias)
        if r.conv_shortcut is None:
            self.w2, self.c2, self.short = ops.pack_conv3x3(h16(r.conv2.weight)), f32(r.conv2.bias), False
        else:       # 1x1 shortcut fused as extra K columns of conv2 (as the UNet's resnets do)
            self.w2 = ops.pack_conv3x3(h16(r.conv2.weight), h16(r.conv_shortcut.weight))
            self.c2, self.short = f32(r.conv2.bias + r.conv_shortcut.bias), True

    def __call__(self, x, n, h, w):
        a = ops.group_norm(x, self.g1, self.b1, n, 32, 1e-6, True)
        a = ops.conv3x3(a, self.w1, self.c1, n, h, w)
        a = ops.group_norm(a, self.g2, self.b2, n, 32, 1e-6, True)
        if self.short:
            return ops.conv3x3(a, self.w2, self.c2, n, h, w, sc1=x)
        return ops.conv3x3(a, self.w2, self.c2, n, h, w, residual=x)


class HipAutoencoderKL(torch.nn.Module):
    def __init__(self, vae):
        super().__init__()
        self.vae = vae                                   # the stock module: encode(), the small edge convolutions, config
        self.config = vae.config
        self._packed = None
        self._packed_enc = None

    def _mid(self, p, x, n, h, w, c):
        """mid block: resnet, single-head attention (GroupNorm and the four projections on the engine, the (h w) x (h w)
        product with head dim C on stock SDPA), resnet."""
        x = p.mid[0](x, n, h, w)
        a = ops.group_norm(x, p.att_g, p.att_b, n, 32, 1e-6, False)
        qkv = ops.linear(a, p.att_wqkv, bias=p.att_bqkv).reshape(n, h * w, 3, c)
        o = F.scaled_dot_product_attention(qkv[:, None, :, 0], qkv[:, None, :, 1], qkv[:, None, :, 2])[:, 0]
        x = ops.linear(o.reshape(n * h * w, c).contiguous(), p.att_wo, bias=p.att_bo, residual=x)
        return p.mid[1](x, n, h, w)

    @staticmethod
    def _pack_mid(mid, dev):
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        h16 = lambda t: t.detach().to(dev, torch.float16).contiguous()
        att = mid.attentions[0]
        p = SimpleNamespace()
        p.mid = [_Res(mid.resnets[0], dev), _Res(mid.resnets[1], dev)]
        p.att_g, p.att_b = f32(att.group_norm.weight), f32(att.group_norm.bias)
        p.att_wqkv = h16(torch.cat([att.query.weight, att.key.weight, att.value.weight], 0))
        p.att_bqkv = f32(torch.cat([att.query.bias, att.key.bias, att.value.bias], 0))
        p.att_wo, p.att_bo = h16(att.proj_attn.weight), f32(att.proj_attn.bias)
        return p

    @torch.no_grad()
    def encode(self, x: torch.Tensor):
        """Encoder resnets / mid block / output norm on the engine; the 3-channel input convolution, the three stride-2
        downsampling convolutions (asymmetric (0,1,0,1) padding, which the gather kernel does not express) and the 8-channel
        output convolutions stay on stock ops."""
        from .autoencoder_kl import DiagonalGaussianDistribution
        dev = x.device
        e = self.vae.encoder
        if self._packed_enc is None or self._packed_enc[0] != dev:
            f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
            p = self._pack_mid(e.mid_block, dev)
            p.downs = [[_Res(r, dev) for r in blk.resnets] for blk in e.down_blocks]
            p.out_g, p.out_b = f32(e.conv_norm_out.weight), f32(e.conv_norm_out.bias)
            self._packed_enc = (dev, p)
        p = self._packed_enc[1]
        wdt = next(self.vae.parameters()).dtype
        t = e.conv_in(x.to(wdt)).to(torch.float16)
        n, c, h, w = t.shape
        r = _rows(t)
        for blk, res in zip(e.down_blocks, p.downs):
            for rb in res:
                r = rb(r, n, h, w)
            c = res[-1].cout
            if hasattr(blk, "downsamplers"):
                t = blk.downsamplers[0](r.reshape(n, h, w, c).permute(0, 3, 1, 2).to(wdt)).to(torch.float16)
                n, c, h, w = t.shape
                r = _rows(t)
        r = self._mid(p, r, n, h, w, c)
        r = ops.group_norm(r, p.out_g, p.out_b, n, 32, 1e-6, True)
        moments = self.vae.quant_conv(e.conv_out(r.reshape(n, h, w, c).permute(0, 3, 1, 2).to(wdt)))
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(moments.float()))

    def _pack(self, dev):
        d = self.vae.decoder
        groups = d.conv_norm_out.num_groups
        if groups != 32:
            raise NotImplementedError("HipAutoencoderKL: norm_num_groups must be 32")
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        h16 = lambda t: t.detach().to(dev, torch.float16).contiguous()
        p = self._pack_mid(d.mid_block, dev)
        p.ups = []
        for blk in d.up_blocks:
            res = [_Res(r, dev) for r in blk.resnets]
            up = None
            if hasattr(blk, "upsamplers"):
                c = blk.upsamplers[0].conv
                up = (ops.pack_conv3x3(h16(c.weight)), f32(c.bias))
            p.ups.append((res, up))
        p.out_g, p.out_b = f32(d.conv_norm_out.weight), f32(d.conv_norm_out.bias)
        self._packed = (dev, p)
        return p

    @torch.no_grad()
    def decode(self, z: torch.Tensor):
        dev = z.device
        p = self._packed[1] if self._packed is not None and self._packed[0] == dev else self._pack(dev)
        d = self.vae.decoder
        wdt = next(self.vae.parameters()).dtype
        x = d.conv_in(self.vae.post_quant_conv(z.to(wdt)))               # 4 -> 4 -> C channels: stock ops (K = 36)
        n, c, h, w = x.shape
        x = _rows(x.to(torch.float16))
        x = self._mid(p, x, n, h, w, c)
        for res, up in p.ups:
            for r in res:
                x = r(x, n, h, w)
            if up is not None:
                x = ops.conv3x3(x, up[0], up[1], n, h, w, ups=1)       # nearest x2 folded into the gather
                h, w = 2 * h, 2 * w
        x = ops.group_norm(x, p.out_g, p.out_b, n, 32, 1e-6, True)
        x = x.reshape(n, h, w, -1).permute(0, 3, 1, 2)
        return SimpleNamespace(sample=d.conv_out(x.to(wdt)))             # C -> 3 channels: stock op
