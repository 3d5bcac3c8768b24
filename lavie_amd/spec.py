"""Parameter inventory of the denoiser: the checkpoint-format contract.

Names and shapes are exactly those of the reference's state dict
(`UNet3DConditionModel.__init__`, /root/reference/base/models/unet.py:142-295, 830 tensors,
909,124,116 parameters at the base config) so that `lavie_base.pt` drops in unchanged
(`find_model`, base/download.py:10-18)."""
from typing import Dict, Iterator, List, Tuple

from .config import UNetConfig, BASE_CONFIG

Shape = Tuple[int, ...]


def _affine(prefix: str, c: int) -> Iterator[Tuple[str, Shape]]:
    yield prefix + ".weight", (c,)
    yield prefix + ".bias", (c,)


def _conv(prefix: str, cin: int, cout: int, k: int) -> Iterator[Tuple[str, Shape]]:
    yield prefix + ".weight", (cout, cin, k, k)
    yield prefix + ".bias", (cout,)


def _linear(prefix: str, cin: int, cout: int, bias: bool = True) -> Iterator[Tuple[str, Shape]]:
    yield prefix + ".weight", (cout, cin)
    if bias:
        yield prefix + ".bias", (cout,)


def _resnet(prefix: str, cin: int, cout: int, temb: int) -> Iterator[Tuple[str, Shape]]:
    yield from _affine(prefix + ".norm1", cin)
    yield from _conv(prefix + ".conv1", cin, cout, 3)
    yield from _linear(prefix + ".time_emb_proj", temb, cout)
    yield from _affine(prefix + ".norm2", cout)
    yield from _conv(prefix + ".conv2", cout, cout, 3)
    if cin != cout:
        yield from _conv(prefix + ".conv_shortcut", cin, cout, 1)


def _attention(prefix: str, c: int, kv_dim: int) -> Iterator[Tuple[str, Shape]]:
    yield from _linear(prefix + ".to_q", c, c, bias=False)
    yield from _linear(prefix + ".to_k", kv_dim, c, bias=False)
    yield from _linear(prefix + ".to_v", kv_dim, c, bias=False)
    yield from _linear(prefix + ".to_out.0", c, c)


def _temporal_conv(prefix: str, cin: int, cout: int, taps: int) -> Iterator[Tuple[str, Shape]]:
    yield prefix + ".weight", (cout, cin, taps, 1, 1)
    yield prefix + ".bias", (cout,)


def _transformer(prefix: str, c: int, cfg: UNetConfig, level: int = -1) -> Iterator[Tuple[str, Shape]]:
    """`level` = -1 for the mid block, which always self-attends."""
    vsr = cfg.vsr_blocks
    cross1 = vsr and level >= 0 and bool(cfg.only_cross_attention[level])
    tname = "temporal" if vsr else "temp"
    if vsr:                                   # resblock_temporal (vsr/models/attention.py:350): ResnetBlock3DCNN, no temb
        yield from _affine(prefix + ".resblock_temporal.norm1", c)
        yield from _temporal_conv(prefix + ".resblock_temporal.conv1", c, c, 3)
        yield from _affine(prefix + ".resblock_temporal.norm2", c)
        yield from _temporal_conv(prefix + ".resblock_temporal.conv2", c, c, 3)
    yield from _affine(prefix + ".norm", c)
    yield from (_linear(prefix + ".proj_in", c, c) if vsr else _conv(prefix + ".proj_in", c, c, 1))
    blk = prefix + ".transformer_blocks.0"
    yield from _attention(blk + ".attn1", c, cfg.cross_attention_dim if cross1 else c)
    yield from _affine(blk + ".norm1", c)
    yield from _attention(blk + ".attn2", c, cfg.cross_attention_dim)
    yield from _affine(blk + ".norm2", c)
    yield from _attention(blk + f".attn_{tname}", c, c)
    if not cfg.temporal_plain:
        yield blk + f".attn_{tname}.time_rel_pos_bias.relative_attention_bias.weight", (cfg.rel_buckets, cfg.heads)
        yield blk + f".attn_{tname}.rotary_emb.freqs", (cfg.rotary_dim // 2,)
    yield from _affine(blk + f".norm_{tname}", c)
    yield from _linear(blk + ".ff.net.0.proj", c, 8 * c)
    yield from _linear(blk + ".ff.net.2", 4 * c, c)
    yield from _affine(blk + ".norm3", c)
    yield from (_linear(prefix + ".proj_out", c, c) if vsr else _conv(prefix + ".proj_out", c, c, 1))


def _temporal_module(prefix: str, c: int, temb: int) -> Iterator[Tuple[str, Shape]]:
    """TemporalModule3D (vsr/models/temporal_module.py:65-150) with attention_block_types ("", "")."""
    t, s = prefix + ".resblocks_3d_t", prefix + ".resblocks_3d_s"
    yield from _affine(t + ".norm1", c)
    yield from _temporal_conv(t + ".conv1", c, c, 5)
    yield from _linear(t + ".time_emb_proj", temb, c)
    yield from _affine(t + ".norm2", c)
    yield from _temporal_conv(t + ".conv2", c, c, 3)
    yield from _resnet(s, c, c, temb)
    yield from _conv(prefix + ".shift_conv", c, c, 1)


def iter_params(cfg: UNetConfig = BASE_CONFIG) -> Iterator[Tuple[str, Shape]]:
    widths = cfg.block_out_channels
    temb = cfg.time_embed_dim
    levels = len(widths)
    tmod = cfg.vsr_temporal_modules
    if cfg.num_class_embeds:
        yield "class_embedding.weight", (cfg.num_class_embeds, temb)
    if cfg.vsr_blocks and not cfg.temporal_plain:
        yield "temporal_rotary_emb.freqs", (cfg.rotary_dim // 2,)        # the shared RotaryEmbedding (vsr/models/unet.py:206)
    yield from _conv("conv_in", cfg.in_channels, widths[0], 3)
    yield from _linear("time_embedding.linear_1", widths[0], temb)
    yield from _linear("time_embedding.linear_2", temb, temb)

    skips: List[int] = [widths[0]]
    cur = widths[0]
    for lvl, width in enumerate(widths):
        for j in range(cfg.layers_per_block):
            yield from _resnet(f"down_blocks.{lvl}.resnets.{j}", cur, width, temb)
            cur = width
            if cfg.attn_levels[lvl]:
                yield from _transformer(f"down_blocks.{lvl}.attentions.{j}", cur, cfg, lvl)
            skips.append(cur)
        if lvl + 1 < levels:
            yield from _conv(f"down_blocks.{lvl}.downsamplers.0.conv", cur, cur, 3)
            skips.append(cur)
        if tmod:
            yield from _temporal_module(f"down_temporal_blocks.{lvl}", cur, temb)

    yield from _resnet("mid_block.resnets.0", cur, cur, temb)
    yield from _transformer("mid_block.attentions.0", cur, cfg)
    yield from _resnet("mid_block.resnets.1", cur, cur, temb)
    if tmod:
        yield from _temporal_module("mid_temporal_block", cur, temb)

    for i in range(levels):
        lvl = levels - 1 - i
        for j in range(cfg.layers_per_block + 1):
            yield from _resnet(f"up_blocks.{i}.resnets.{j}", cur + skips.pop(), widths[lvl], temb)
            cur = widths[lvl]
            if cfg.attn_levels[lvl]:
                yield from _transformer(f"up_blocks.{i}.attentions.{j}", cur, cfg, lvl)
        if i + 1 < levels:
            yield from _conv(f"up_blocks.{i}.upsamplers.0.conv", cur, cur, 3)
        if tmod:
            yield from _temporal_module(f"up_temporal_blocks.{i}", cur, temb)

    yield from _affine("conv_norm_out", widths[0])
    yield from _conv("conv_out", widths[0], cfg.out_channels, 3)


def param_shapes(cfg: UNetConfig = BASE_CONFIG) -> Dict[str, Shape]:
    return dict(iter_params(cfg))


def param_count(cfg: UNetConfig = BASE_CONFIG) -> int:
    total = 0
    for _, shape in iter_params(cfg):
        n = 1
        for s in shape:
            n *= s
        total += n
    return total
