"""fp32 CPU restatement of `UNet3DConditionModel.forward` (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional and state-dict driven: every function takes the reference's state dict (same key
names and shapes as `/root/reference/base/models/unet.py` builds) plus a key prefix, so a real
`lavie_base.pt` or the reference's own random-init state dict can be evaluated directly.

Tensors follow the reference's conventions: video activations are `[b, c, f, h, w]`,
transformer tokens are `[(b f), (h w), c]`.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


@dataclass(frozen=True)
class UNetConfig:
    """Constructor arguments of the reference UNet that matter on the inference path
    (/root/reference/base/models/unet.py:102-141, as instantiated by
    `from_pretrained_2d` with SD-1.4's config: cross_attention_dim 768, heads 8)."""
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    heads: int = 8                      # attention_head_dim=8 is used as the HEAD COUNT (unet_blocks.py:289-291)
    cross_attention_dim: int = 768
    norm_groups: int = 32
    norm_eps: float = 1e-5
    rotary_dim: int = 32                # RotaryEmbedding(32), unet.py:185
    rel_buckets: int = 32               # RelativePositionBias(num_buckets=32, max_distance=32), attention.py:577
    rel_max_distance: int = 32
    # which levels carry transformer blocks: CrossAttnDownBlock3D x3 + DownBlock3D (unet.py:110-122)
    attn_levels: Tuple[bool, ...] = field(default=(True, True, True, False))
    # --- the frame-interpolation model's block variant (interpolation/models/attention.py:456-606) ---
    sparse_causal_attn1: bool = False   # `use_first_frame`: attn1 keys/values = first frame || previous frame (:609-665)
    temporal_plain: bool = False        # attn_temp is a plain CrossAttention over frames: no rotary, no bias (:525-533, 211-289)
    ff_before_temporal: bool = False    # block order spatial -> text -> FF -> temporal (:566-606)


BASE = UNetConfig()
# UNet3DConditionModel.from_pretrained_2d(..., use_concat=True, copy_no_mask=True) (interpolation/models/unet.py:477-506):
# 8 input channels (noisy latent || copied low-frame-rate latent), use_first_frame=True, use_relative_position=False
INTERPOLATION = UNetConfig(in_channels=8, sparse_causal_attn1=True, temporal_plain=True, ff_before_temporal=True)


# --------------------------------------------------------------------------- embeddings
def timestep_sinusoid(t: torch.Tensor, dim: int) -> torch.Tensor:
    """Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) (unet.py:153, third-party diffusers;
    in-tree textual spec base/models/utils.py:74-94): [cos(t w_k) | sin(t w_k)], w_k = 1e4^(-k/half)."""
    half = dim // 2
    w = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t.reshape(-1, 1).to(torch.float32) * w.reshape(1, -1)
    return torch.cat([ang.cos(), ang.sin()], dim=1)


def time_embedding(sd: SD, t: torch.Tensor, cfg: UNetConfig) -> torch.Tensor:
    """unet.py:428-434: sinusoid -> Linear -> SiLU -> Linear (TimestepEmbedding, third-party)."""
    e = timestep_sinusoid(t, cfg.block_out_channels[0])
    e = F.linear(e, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    return F.linear(F.silu(e), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])


# --------------------------------------------------------------------------- conv / resnet
def conv_frames(x, w, b, stride=1, padding=1):
    """InflatedConv3d (resnet.py:13-21): a Conv2d applied to every frame independently."""
    bsz, c, f, h, wd = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(bsz * f, c, h, wd), w, b, stride=stride, padding=padding)
    return y.reshape(bsz, f, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)


def group_norm_video(x, w, b, groups, eps):
    """nn.GroupNorm on the 5-D tensor: statistics span (C/G, F, H, W) — ACROSS frames
    (resnet.py:180,191; unet.py:504)."""
    return F.group_norm(x, groups, w, b, eps)


def resnet_block(sd: SD, p: str, x, temb, cfg: UNetConfig):
    """ResnetBlock3D.forward (resnet.py:177-207), time_embedding_norm='default', scale factor 1."""
    h = F.silu(group_norm_video(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.norm_groups, cfg.norm_eps))
    h = conv_frames(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"])
    tproj = F.linear(F.silu(temb), sd[p + "time_emb_proj.weight"], sd[p + "time_emb_proj.bias"])
    h = h + tproj[:, :, None, None, None]
    h = F.silu(group_norm_video(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.norm_groups, cfg.norm_eps))
    h = conv_frames(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"])
    if (p + "conv_shortcut.weight") in sd:      # present iff in_channels != out_channels (resnet.py:171-175)
        x = conv_frames(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"], padding=0)
    return x + h


def downsample(sd: SD, p: str, x):
    """Downsample3D (resnet.py:102-110): 3x3 stride-2 pad-1 conv per frame."""
    return conv_frames(x, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=2, padding=1)


def upsample(sd: SD, p: str, x):
    """Upsample3D (resnet.py:44-76): nearest x2 on (h, w) only, then 3x3 conv."""
    x = x.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
    return conv_frames(x, sd[p + "conv.weight"], sd[p + "conv.bias"])


# --------------------------------------------------------------------------- attention
def split_heads(t, heads):
    b, n, c = t.shape
    return t.reshape(b, n, heads, c // heads).permute(0, 2, 1, 3)          # b h n d


def merge_heads(t):
    b, h, n, d = t.shape
    return t.permute(0, 2, 1, 3).reshape(b, n, h * d)


def cross_attention(sd: SD, p: str, x, ctx, heads):
    """CrossAttention.forward + _attention (attention.py:146-206, 209-239): no-bias q/k/v,
    softmax(scale q k^T) v, to_out with bias.  ctx=None -> self attention."""
    src = x if ctx is None else ctx
    q = split_heads(F.linear(x, sd[p + "to_q.weight"]), heads)
    k = split_heads(F.linear(src, sd[p + "to_k.weight"]), heads)
    v = split_heads(F.linear(src, sd[p + "to_v.weight"]), heads)
    scale = q.shape[-1] ** -0.5
    prob = torch.softmax(scale * (q @ k.transpose(-1, -2)), dim=-1)
    return F.linear(merge_heads(prob @ v), sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def sparse_causal_attention(sd: SD, p: str, x, frames: int, heads: int):
    """SparseCausalAttention.forward (interpolation/models/attention.py:609-665): x [(b f), d, c]; frame i attends to the
    2d keys/values [frame 0 of its video || frame max(i-1, 0)] (:630-639), plain softmax(scale q k^T) v (:268-289)."""
    bf, d, c = x.shape
    q = split_heads(F.linear(x, sd[p + "to_q.weight"]), heads)
    former = (torch.arange(frames) - 1).clamp(min=0)

    def gather(t):
        t = t.reshape(bf // frames, frames, d, c)
        return torch.cat([t[:, [0] * frames], t[:, former]], dim=2).reshape(bf, 2 * d, c)

    k = split_heads(gather(F.linear(x, sd[p + "to_k.weight"])), heads)
    v = split_heads(gather(F.linear(x, sd[p + "to_v.weight"])), heads)
    prob = torch.softmax((q.shape[-1] ** -0.5) * (q @ k.transpose(-1, -2)), dim=-1)
    return F.linear(merge_heads(prob @ v), sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def rel_pos_bucket_table(n: int, num_buckets: int = 32, max_distance: int = 32) -> torch.Tensor:
    """RelativePositionBias._relative_position_bucket on rel = k_pos - q_pos
    (attention.py:681-705).  Entry [i, j] is the bucket for query i, key j (int64)."""
    q = torch.arange(n).reshape(n, 1)
    k = torch.arange(n).reshape(1, n)
    dist = q - k                                   # n = -(k - q)
    half = num_buckets // 2
    bucket = (dist < 0).long() * half              # key in the future -> upper half of the table
    dist = dist.abs()
    exact = half // 2
    log_part = exact + (torch.log(dist.float() / exact) / math.log(max_distance / exact) * (half - exact)).long()
    log_part = torch.clamp(log_part, max=half - 1)
    return bucket + torch.where(dist < exact, dist, log_part)


def rel_pos_bias(sd: SD, p: str, n: int, cfg: UNetConfig) -> torch.Tensor:
    """[heads, n, n] additive bias (attention.py:701-707)."""
    table = rel_pos_bucket_table(n, cfg.rel_buckets, cfg.rel_max_distance)
    emb = sd[p + "time_rel_pos_bias.relative_attention_bias.weight"]      # [32, heads]
    return emb[table].permute(2, 0, 1)


def rotary(t: torch.Tensor, rot_dim: int = 32, theta: float = 10000.0) -> torch.Tensor:
    """RotaryEmbedding(32).rotate_queries_or_keys (third-party, unpinned; call site
    attention.py:644-646).  t: [..., n, d]; positions 0..n-1 on axis -2; channel pairs
    (2k, 2k+1) of the first `rot_dim` channels rotate by pos * theta^(-2k/rot_dim); the rest
    pass through.  Angles are always evaluated in fp32 (SURVEY §8c decision)."""
    n = t.shape[-2]
    inv = theta ** (-torch.arange(0, rot_dim, 2, dtype=torch.float32) / rot_dim)          # [rot_dim/2]
    ang = torch.arange(n, dtype=torch.float32).reshape(n, 1) * inv.reshape(1, -1)         # [n, rot_dim/2]
    c, s = ang.cos(), ang.sin()
    even, odd = t[..., 0:rot_dim:2], t[..., 1:rot_dim:2]
    out = t.clone()
    out[..., 0:rot_dim:2] = even * c - odd * s
    out[..., 1:rot_dim:2] = odd * c + even * s
    return out


def temporal_attention_core(q, k, v, bias, rot_dim: int = 32):
    """TemporalAttention._attention (attention.py:634-667) on split heads.
    q, k, v: [n, heads, f, dh]; bias: [heads, f, f] -> [n, heads, f, dh].  q is scaled BEFORE the
    rotary embedding (640, 644-646); scores get the relative-position bias (650) and the
    max-subtracted softmax (656-658)."""
    q = rotary(q * (q.shape[-1] ** -0.5), rot_dim)
    k = rotary(k, rot_dim)
    score = q @ k.transpose(-1, -2) + bias
    prob = torch.softmax(score - score.amax(dim=-1, keepdim=True), dim=-1)
    return prob @ v


def temporal_attention(sd: SD, p: str, x, cfg: UNetConfig):
    """TemporalAttention.forward (attention.py:580-632): x [b*d, f, c]; no-bias q/k/v, to_out with bias."""
    heads = cfg.heads
    q = split_heads(F.linear(x, sd[p + "to_q.weight"]), heads)
    k = split_heads(F.linear(x, sd[p + "to_k.weight"]), heads)
    v = split_heads(F.linear(x, sd[p + "to_v.weight"]), heads)
    o = temporal_attention_core(q, k, v, rel_pos_bias(sd, p, x.shape[1], cfg), cfg.rotary_dim)
    return F.linear(merge_heads(o), sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def geglu_ff(sd: SD, p: str, x):
    """diffusers FeedForward(activation_fn='geglu') (attention.py:479,558; in-tree textual spec
    vsr/models/diffusers_attention.py:734-822): proj -> (h, gate) -> h * gelu_erf(gate) -> Linear."""
    h, gate = F.linear(x, sd[p + "net.0.proj.weight"], sd[p + "net.0.proj.bias"]).chunk(2, dim=-1)
    return F.linear(h * F.gelu(gate), sd[p + "net.2.weight"], sd[p + "net.2.bias"])


def layer_norm(sd: SD, p: str, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


def transformer_block(sd: SD, p: str, x, ctx, frames: int, cfg: UNetConfig):
    """BasicTransformerBlock.forward, eval branch (attention.py:511-534, 548-560).
    x: [(b f), d, c]; ctx: [(b f), 77, cross_dim].  Order: spatial -> text -> temporal -> FF.
    Interpolation variant (interpolation/models/attention.py:566-606): sparse-causal attn1, order spatial -> text -> FF ->
    temporal, and attn_temp is the plain CrossAttention (no rotary, no relative-position bias)."""
    n1 = layer_norm(sd, p + "norm1.", x)
    if cfg.sparse_causal_attn1:
        x = x + sparse_causal_attention(sd, p + "attn1.", n1, frames, cfg.heads)
    else:
        x = x + cross_attention(sd, p + "attn1.", n1, None, cfg.heads)
    x = x + cross_attention(sd, p + "attn2.", layer_norm(sd, p + "norm2.", x), ctx, cfg.heads)
    if cfg.ff_before_temporal:
        x = x + geglu_ff(sd, p + "ff.", layer_norm(sd, p + "norm3.", x))
    bf, d, c = x.shape
    xt = x.reshape(bf // frames, frames, d, c).permute(0, 2, 1, 3).reshape(-1, frames, c)      # (b d) f c
    nt = layer_norm(sd, p + "norm_temp.", xt)
    if cfg.temporal_plain:
        xt = xt + cross_attention(sd, p + "attn_temp.", nt, None, cfg.heads)
    else:
        xt = xt + temporal_attention(sd, p + "attn_temp.", nt, cfg)
    x = xt.reshape(bf // frames, d, frames, c).permute(0, 2, 1, 3).reshape(bf, d, c)
    if cfg.ff_before_temporal:
        return x
    return x + geglu_ff(sd, p + "ff.", layer_norm(sd, p + "norm3.", x))


def transformer3d(sd: SD, p: str, x, ctx, cfg: UNetConfig):
    """Transformer3DModel.forward (attention.py:358-407): per-FRAME GroupNorm (eps 1e-6),
    1x1 proj_in, one transformer block, 1x1 proj_out, residual."""
    b, c, f, h, w = x.shape
    frames = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    ctx_rep = ctx.repeat_interleave(f, dim=0)                                    # 'b n c -> (b f) n c'
    t = F.group_norm(frames, cfg.norm_groups, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    t = F.conv2d(t, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    t = t.permute(0, 2, 3, 1).reshape(b * f, h * w, -1)
    t = transformer_block(sd, p + "transformer_blocks.0.", t, ctx_rep, f, cfg)
    t = t.reshape(b * f, h, w, -1).permute(0, 3, 1, 2)
    t = F.conv2d(t, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"]) + frames
    return t.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)


# --------------------------------------------------------------------------- whole UNet
def unet_forward(sd: SD, sample, timesteps, ctx, cfg: UNetConfig = BASE):
    """UNet3DConditionModel.forward (unet.py:366-512) with the block wiring of
    unet_blocks.py:226-232, 320-362, 417-441, 524-574, 625-648.
    sample [b, 4, f, h, w]; timesteps scalar or [b]; ctx [b, 77, cross_dim] -> [b, 4, f, h, w]."""
    sample = sample.float()
    t = torch.as_tensor(timesteps).reshape(-1).expand(sample.shape[0])
    emb = time_embedding(sd, t, cfg)
    ctx = ctx.float()
    nlev = len(cfg.block_out_channels)

    x = conv_frames(sample, sd["conv_in.weight"], sd["conv_in.bias"])
    skips = [x]
    for lvl in range(nlev):
        for j in range(cfg.layers_per_block):
            x = resnet_block(sd, f"down_blocks.{lvl}.resnets.{j}.", x, emb, cfg)
            if cfg.attn_levels[lvl]:
                x = transformer3d(sd, f"down_blocks.{lvl}.attentions.{j}.", x, ctx, cfg)
            skips.append(x)
        if lvl != nlev - 1:
            x = downsample(sd, f"down_blocks.{lvl}.downsamplers.0.", x)
            skips.append(x)

    x = resnet_block(sd, "mid_block.resnets.0.", x, emb, cfg)
    x = transformer3d(sd, "mid_block.attentions.0.", x, ctx, cfg)
    x = resnet_block(sd, "mid_block.resnets.1.", x, emb, cfg)

    for i in range(nlev):
        lvl = nlev - 1 - i
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(sd, f"up_blocks.{i}.resnets.{j}.", x, emb, cfg)
            if cfg.attn_levels[lvl]:
                x = transformer3d(sd, f"up_blocks.{i}.attentions.{j}.", x, ctx, cfg)
        if i != nlev - 1:
            x = upsample(sd, f"up_blocks.{i}.upsamplers.0.", x)

    x = F.silu(group_norm_video(x, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], cfg.norm_groups, cfg.norm_eps))
    return conv_frames(x, sd["conv_out.weight"], sd["conv_out.bias"])


# --------------------------------------------------------------------------- parameter inventory
def param_shapes(cfg: UNetConfig = BASE) -> Dict[str, Tuple[int, ...]]:
    """Every state-dict entry the reference constructor creates (unet.py:142-295), in the
    oracle's own derivation; tests compare it with the reference's real state dict and with the
    product's `lavie_amd.spec.param_shapes`."""
    out: Dict[str, Tuple[int, ...]] = {}
    ch = cfg.block_out_channels
    temb = ch[0] * 4
    ctxd = cfg.cross_attention_dim

    def conv(p, cin, cout, k):
        out[p + "weight"] = (cout, cin, k, k)
        out[p + "bias"] = (cout,)

    def lin(p, cin, cout, bias=True):
        out[p + "weight"] = (cout, cin)
        if bias:
            out[p + "bias"] = (cout,)

    def norm(p, c):
        out[p + "weight"] = (c,)
        out[p + "bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + "norm1.", cin)
        conv(p + "conv1.", cin, cout, 3)
        lin(p + "time_emb_proj.", temb, cout)
        norm(p + "norm2.", cout)
        conv(p + "conv2.", cout, cout, 3)
        if cin != cout:
            conv(p + "conv_shortcut.", cin, cout, 1)

    def attn(p, c, kv):
        lin(p + "to_q.", c, c, False)
        lin(p + "to_k.", kv, c, False)
        lin(p + "to_v.", kv, c, False)
        lin(p + "to_out.0.", c, c)

    def transformer(p, c):
        norm(p + "norm.", c)
        conv(p + "proj_in.", c, c, 1)
        b = p + "transformer_blocks.0."
        attn(b + "attn1.", c, c)
        norm(b + "norm1.", c)
        attn(b + "attn2.", c, ctxd)
        norm(b + "norm2.", c)
        attn(b + "attn_temp.", c, c)
        if not cfg.temporal_plain:
            out[b + "attn_temp.time_rel_pos_bias.relative_attention_bias.weight"] = (cfg.rel_buckets, cfg.heads)
            out[b + "attn_temp.rotary_emb.freqs"] = (cfg.rotary_dim // 2,)
        norm(b + "norm_temp.", c)
        lin(b + "ff.net.0.proj.", c, 8 * c)
        lin(b + "ff.net.2.", 4 * c, c)
        norm(b + "norm3.", c)
        conv(p + "proj_out.", c, c, 1)

    conv("conv_in.", cfg.in_channels, ch[0], 3)
    lin("time_embedding.linear_1.", ch[0], temb)
    lin("time_embedding.linear_2.", temb, temb)
    nlev = len(ch)
    skip_ch = [ch[0]]
    cur = ch[0]
    for lvl in range(nlev):
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{lvl}.resnets.{j}.", cur, ch[lvl])
            cur = ch[lvl]
            if cfg.attn_levels[lvl]:
                transformer(f"down_blocks.{lvl}.attentions.{j}.", cur)
            skip_ch.append(cur)
        if lvl != nlev - 1:
            conv(f"down_blocks.{lvl}.downsamplers.0.conv.", cur, cur, 3)
            skip_ch.append(cur)
    resnet("mid_block.resnets.0.", cur, cur)
    transformer("mid_block.attentions.0.", cur)
    resnet("mid_block.resnets.1.", cur, cur)
    for i in range(nlev):
        lvl = nlev - 1 - i
        for j in range(cfg.layers_per_block + 1):
            resnet(f"up_blocks.{i}.resnets.{j}.", cur + skip_ch.pop(), ch[lvl])
            cur = ch[lvl]
            if cfg.attn_levels[lvl]:
                transformer(f"up_blocks.{i}.attentions.{j}.", cur)
        if i != nlev - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv.", cur, cur, 3)
    norm("conv_norm_out.", ch[0])
    conv("conv_out.", ch[0], cfg.out_channels, 3)
    return out
