"""fp32 CPU restatement of the VSR stage's temporal building blocks (TEST INFRASTRUCTURE, see oracle/__init__.py) —
SURVEY.md §8 f2, first pieces: the (T,1,1) temporal convolution and `ResnetBlock3DCNN`.

PINNED: `vsr/models/resnet.py` imports only torch / einops and is imported directly in the build container
(tests/test_oracle_vs_reference.py, tier T1); outputs frozen in tests/golden/vsr_resnet3dcnn.pt."""
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def temporal_conv(x, w, b):
    """nn.Conv3d(kernel (T,1,1), stride 1, padding (T//2,0,0)) (vsr/models/resnet.py:258-259, 274) written as T shifted
    matrix products: x [b, c, f, h, w], w [cout, cin, T, 1, 1]."""
    taps = w.shape[2]
    f = x.shape[2]
    xp = F.pad(x, (0, 0, 0, 0, taps // 2, taps // 2))
    y = sum(torch.einsum("oc,bcfhw->bofhw", w[:, :, t, 0, 0], xp[:, :, t:t + f]) for t in range(taps))
    return y + b.reshape(1, -1, 1, 1, 1)


def resnet_block_3dcnn(sd: SD, p: str, x, temb, groups: int = 32, eps: float = 1e-6):
    """ResnetBlock3DCNN.forward (vsr/models/resnet.py:283-315), in == out channels (no shortcut), time_embedding_norm
    'default', output_scale_factor 1: GroupNorm over (C/G, F, H, W) + SiLU -> temporal conv (T) + temb -> GroupNorm +
    SiLU -> temporal conv (3) -> residual."""
    h = F.silu(F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps))
    h = temporal_conv(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"])
    t = F.linear(F.silu(temb), sd[p + "time_emb_proj.weight"], sd[p + "time_emb_proj.bias"])
    h = h + t[:, :, None, None, None]
    h = F.silu(F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps))
    h = temporal_conv(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"])
    return x + h


# --------------------------------------------------------------------------- VSR Transformer3DModel
def vsr_transformer_block(sd: SD, p: str, x, ctx, frames: int, heads: int, only_cross_attention: bool, ucfg):
    """vsr/models/attention.py:552-594 (eval): attn1 is text cross-attention on `only_cross_attention` levels and spatial
    self-attention otherwise; then text cross-attention, temporal attention (scale -> rotary -> + relative-position bias
    -> softmax, :728-775: the base model's TemporalAttention under the names attn_temporal / norm_temporal), feed-forward."""
    from . import unet_fp32 as O
    n1 = O.layer_norm(sd, p + "norm1.", x)
    x = x + O.cross_attention(sd, p + "attn1.", n1, ctx if only_cross_attention else None, heads)
    x = x + O.cross_attention(sd, p + "attn2.", O.layer_norm(sd, p + "norm2.", x), ctx, heads)
    bf, d, c = x.shape
    xt = x.reshape(bf // frames, frames, d, c).permute(0, 2, 1, 3).reshape(-1, frames, c)
    nt = O.layer_norm(sd, p + "norm_temporal.", xt)
    xt = xt + O.temporal_attention(sd, p + "attn_temporal.", nt, ucfg)
    x = xt.reshape(bf // frames, d, frames, c).permute(0, 2, 1, 3).reshape(bf, d, c)
    return x + O.geglu_ff(sd, p + "ff.", O.layer_norm(sd, p + "norm3.", x))


def vsr_transformer3d(sd: SD, p: str, x, ctx, heads: int = 8, only_cross_attention: bool = False, groups: int = 32):
    """vsr/models/attention.py:386-439 with use_linear_projection=True: ResnetBlock3DCNN (3,1,1) without time embedding
    (:350, 395-398), THEN the residual is taken (:400); per-frame GroupNorm eps 1e-6, Linear proj_in on tokens, one block,
    Linear proj_out, + residual."""
    from . import unet_fp32 as O
    b, c, f, h, w = x.shape
    x = resnet_block_3dcnn_no_temb(sd, p + "resblock_temporal.", x, groups)
    frames_ = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    ctx_rep = ctx.repeat_interleave(f, dim=0)
    t = F.group_norm(frames_, groups, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    t = t.permute(0, 2, 3, 1).reshape(b * f, h * w, c)
    t = F.linear(t, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    ucfg = O.UNetConfig(heads=heads)
    t = vsr_transformer_block(sd, p + "transformer_blocks.0.", t, ctx_rep, f, heads, only_cross_attention, ucfg)
    t = F.linear(t, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    t = t.reshape(b * f, h, w, c).permute(0, 3, 1, 2) + frames_
    return t.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)


def resnet_block_3dcnn_no_temb(sd: SD, p: str, x, groups: int = 32, eps: float = 1e-6):
    """ResnetBlock3DCNN with temb_channels=None (vsr/models/attention.py:350): no time-embedding term."""
    h = F.silu(F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps))
    h = temporal_conv(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"])
    h = F.silu(F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps))
    h = temporal_conv(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"])
    return x + h


# --------------------------------------------------------------------------- TemporalModule3D and the whole VSR UNet
def spatial_resnet_block(sd: SD, p: str, x, temb, groups: int, eps: float):
    """vsr ResnetBlock3D (vsr/models/resnet.py:118-217) = the base model's block; `eps` differs by call site."""
    from . import unet_fp32 as O
    cfg = O.UNetConfig(norm_groups=groups, norm_eps=eps)
    return O.resnet_block(sd, p, x, temb, cfg)


def temporal_module(sd: SD, p: str, x, temb, groups: int = 32):
    """TemporalModule3D.forward (vsr/models/temporal_module.py:153-178) with attention_block_types ("", ""), no video
    condition, use_scale_shift False: ResnetBlock3DCNN (5,1,1) -> ResnetBlock3D (both eps 1e-6, with temb) -> zero-initialised
    1x1 shift conv -> input + shift."""
    from . import unet_fp32 as O
    h = resnet_block_3dcnn(sd, p + "resblocks_3d_t.", x, temb, groups)
    h = spatial_resnet_block(sd, p + "resblocks_3d_s.", h, temb, groups, 1e-6)
    return x + O.conv_frames(h, sd[p + "shift_conv.weight"], sd[p + "shift_conv.bias"], padding=0)


def vsr_unet_forward(sd: SD, sample, low_res, timesteps, ctx, class_labels, block_out_channels=(256, 512, 512, 1024),
                     attn_levels=(False, True, True, True), only_cross_attention=(True, True, True, False),
                     layers_per_block: int = 2, heads: int = 8, groups: int = 32, eps: float = 1e-5):
    """UNet3DVSRModel.forward (vsr/models/unet.py:408-600): sample [b,4,f,h,w] and low_res [b,3,f,h,w] are concatenated on
    channels (:453); emb = time_embedding + class_embedding[noise level] (:489-505); a TemporalModule3D follows every down
    block (after its downsampler, skips taken before it), the mid block and every up block (after its upsampler)."""
    from . import unet_fp32 as O
    x = torch.cat([sample, low_res], dim=1).float()
    ucfg = O.UNetConfig(block_out_channels=tuple(block_out_channels), norm_groups=groups, norm_eps=eps, heads=heads)
    t = torch.as_tensor(timesteps).reshape(-1).expand(x.shape[0])
    emb = O.time_embedding(sd, t, ucfg) + sd["class_embedding.weight"][torch.as_tensor(class_labels).reshape(-1)]
    ctx = ctx.float()
    nlev = len(block_out_channels)
    x = O.conv_frames(x, sd["conv_in.weight"], sd["conv_in.bias"])
    skips = [x]
    for lvl in range(nlev):
        for j in range(layers_per_block):
            x = O.resnet_block(sd, f"down_blocks.{lvl}.resnets.{j}.", x, emb, ucfg)
            if attn_levels[lvl]:
                x = vsr_transformer3d(sd, f"down_blocks.{lvl}.attentions.{j}.", x, ctx, heads, only_cross_attention[lvl], groups)
            skips.append(x)
        if lvl != nlev - 1:
            x = O.downsample(sd, f"down_blocks.{lvl}.downsamplers.0.", x)
            skips.append(x)
        x = temporal_module(sd, f"down_temporal_blocks.{lvl}.", x, emb, groups)
    x = O.resnet_block(sd, "mid_block.resnets.0.", x, emb, ucfg)
    x = vsr_transformer3d(sd, "mid_block.attentions.0.", x, ctx, heads, False, groups)
    x = O.resnet_block(sd, "mid_block.resnets.1.", x, emb, ucfg)
    x = temporal_module(sd, "mid_temporal_block.", x, emb, groups)
    for i in range(nlev):
        lvl = nlev - 1 - i
        for j in range(layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = O.resnet_block(sd, f"up_blocks.{i}.resnets.{j}.", x, emb, ucfg)
            if attn_levels[lvl]:
                x = vsr_transformer3d(sd, f"up_blocks.{i}.attentions.{j}.", x, ctx, heads, only_cross_attention[lvl], groups)
        if i != nlev - 1:
            x = O.upsample(sd, f"up_blocks.{i}.upsamplers.0.", x)
        x = temporal_module(sd, f"up_temporal_blocks.{i}.", x, emb, groups)
    x = F.silu(F.group_norm(x, groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps))
    return O.conv_frames(x, sd["conv_out.weight"], sd["conv_out.bias"])
