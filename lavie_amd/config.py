"""Static description of the denoiser, mirroring the constructor arguments the reference
instantiates (`/root/reference/base/models/unet.py:102-141` through `from_pretrained_2d`,
`unet.py:540-588`, with SD-1.4's `unet/config.json`: cross_attention_dim 768, 8 heads)."""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass(frozen=True)
class UNetConfig:
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    heads: int = 8                       # `attention_head_dim=8` is the head COUNT (unet_blocks.py:289-291)
    cross_attention_dim: int = 768
    norm_groups: int = 32
    norm_eps: float = 1e-5               # resnet / conv_norm_out GroupNorm; Transformer3DModel.norm uses 1e-6
    rotary_dim: int = 32                 # RotaryEmbedding(32), unet.py:185
    rel_buckets: int = 32                # RelativePositionBias(num_buckets=32, max_distance=32), attention.py:577
    rel_max_distance: int = 32
    attn_levels: Tuple[bool, ...] = field(default=(True, True, True, False))
    # Block variant of the frame-interpolation model (interpolation/models/attention.py:456-606, unet.py:477-506)
    sparse_causal_attn1: bool = False    # use_first_frame: attn1 keys/values = first frame || previous frame
    temporal_plain: bool = False         # attn_temp without rotary embedding / relative-position bias
    ff_before_temporal: bool = False     # block order spatial -> text -> FF -> temporal
    # Block variant of the VSR stage's UNet3DVSRModel (vsr/models/attention.py:314-594)
    vsr_blocks: bool = False             # resblock_temporal in front, attn_temporal / norm_temporal names, Linear proj_in/out
    only_cross_attention: Tuple[bool, ...] = ()      # per level: attn1 attends to the text context
    vsr_temporal_modules: bool = False   # TemporalModule3D after every down / mid / up block (vsr/models/unet.py:238-330)
    num_class_embeds: int = 0            # > 0: nn.Embedding noise-level table added to the time embedding (:176-177)

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    def validate(self) -> None:
        n = len(self.block_out_channels)
        if len(self.attn_levels) != n:
            raise ValueError("attn_levels must have one entry per block_out_channels entry")
        for c, a in zip(self.block_out_channels, self.attn_levels):
            if c % 64 != 0:
                raise ValueError(f"channel width {c} must be a multiple of 64 (K-tile of the MFMA kernels)")
            if c % self.norm_groups != 0:
                raise ValueError(f"channel width {c} not divisible by {self.norm_groups} groups")
            min_dh = 8 if self.temporal_plain else self.rotary_dim
            if a and (c % self.heads != 0 or c // self.heads < min_dh or (c // self.heads) % 8 != 0):
                raise ValueError(f"width {c}: head dim must be a multiple of 8 and >= rotary_dim")
        if self.cross_attention_dim % 64 != 0:
            raise ValueError("cross_attention_dim must be a multiple of 64")


BASE_CONFIG = UNetConfig()
# vsr/configs/unet_3d_config.json (UNet3DVSRModel): the SD x4-upscaler UNet with temporal modules; 4 noisy + 3 low-res channels
VSR_CONFIG = UNetConfig(sample_size=128, in_channels=7, block_out_channels=(256, 512, 512, 1024), cross_attention_dim=1024,
                        attn_levels=(False, True, True, True), vsr_blocks=True, only_cross_attention=(True, True, True, False),
                        vsr_temporal_modules=True, num_class_embeds=1000)
# `from_pretrained_2d(..., use_concat=True, copy_no_mask=True)` of the interpolation stage (interpolation/models/unet.py:
# 477-506, interpolation/configs/sample.yaml): noisy latent || copied low-frame-rate latent on 8 input channels
INTERPOLATION_CONFIG = UNetConfig(in_channels=8, sparse_causal_attn1=True, temporal_plain=True, ff_before_temporal=True)
