"""Host-side mirror of `vsr/models/unet.py` (reference): `UNet3DVSRModel` — IN PROGRESS.

Built so far on the engine: the VSR `Transformer3DModel` variant (vsr/models/attention.py:314-594): a `ResnetBlock3DCNN`
(3,1,1) in front of every transformer block, `only_cross_attention` levels whose attn1 attends to the text context,
`nn.Linear` proj_in / proj_out, temporal attention under the names attn_temporal / norm_temporal.  NOT built yet (the
constructor refuses them): `TemporalModule3D` after every block (down_temporal_idx / mid_temporal / up_temporal_idx),
the class-embedded noise level (`num_class_embeds`), 7 input channels.  Until then this class is the seam the parity
tests use for the transformer variant (`lavie_unet_transformer_forward`)."""
from typing import Tuple, Union

from ..unet import UNet3DConditionModel as _BaseUNet


class UNet3DVSRModel(_BaseUNet):
    _allow_vsr_options = True

    def __init__(self, *args, only_cross_attention: Union[bool, Tuple[bool, ...]] = False, use_linear_projection: bool = True,
                 num_class_embeds=None, **kw):
        if num_class_embeds is not None:
            raise NotImplementedError("UNet3DVSRModel: class-embedded noise level (num_class_embeds) is not built yet")
        if not use_linear_projection:
            raise NotImplementedError("UNet3DVSRModel runs with use_linear_projection=True (vsr/configs/unet_3d_config.json)")
        super().__init__(*args, only_cross_attention=only_cross_attention, use_linear_projection=use_linear_projection, **kw)

    def _vsr_config(self, only_cross_attention, use_linear_projection, levels: int) -> dict:
        if isinstance(only_cross_attention, bool):                       # vsr/models/unet.py:198-199
            only_cross_attention = (only_cross_attention,) * levels
        if len(only_cross_attention) != levels:
            raise ValueError("only_cross_attention must have one entry per down block")
        return dict(vsr_blocks=True, only_cross_attention=tuple(bool(v) for v in only_cross_attention))
