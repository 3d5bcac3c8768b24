"""Host-side mirror of `vsr/models/unet.py` (reference): `UNet3DVSRModel`, the video super-resolution UNet.

Same state-dict contract (names / shapes of vsr/models/unet.py:164-345 with `vsr/configs/unet_3d_config.json`) and the same
`forward(sample, timestep, low_res, encoder_hidden_states, class_labels).sample` surface (:408-600) on the gfx950 engine:
  * `Transformer3DModel` variant (vsr/models/attention.py:314-594): a `ResnetBlock3DCNN` (3,1,1) in front of every
    transformer block, `only_cross_attention` levels whose attn1 attends to the text context, `nn.Linear` proj_in / proj_out;
  * a `TemporalModule3D` (temporal_module.py:65-178: ResnetBlock3DCNN (5,1,1) -> ResnetBlock3D -> zero-initialised 1x1
    shift conv, residual) after every down block, the mid block and every up block;
  * `emb = time_embedding + class_embedding[noise level]` (:494-505);
  * 4 noisy + 3 low-resolution channels concatenated at the input (:453); the engine's first convolution packs channel
    pairs, so the 7-channel input and `conv_in.weight` are padded with one zero channel on the way in.
Not supported (the constructor refuses them): `video_condition=True`, temporal transformers inside TemporalModule3D
(`attention_block_types` other than ("", "")), partial `down/up_temporal_idx`, `use_first_frame`."""
import ctypes
from typing import Optional, Tuple, Union

import torch

from .. import _lib
from ..unet import UNet3DConditionModel as _BaseUNet
from ..unet import UNet3DConditionOutput


class UNet3DVSRModel(_BaseUNet):
    _allow_vsr_options = True

    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 7, out_channels: int = 4,
                 down_block_types: Tuple[str, ...] = ("DownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D"),
                 up_block_types: Tuple[str, ...] = ("CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "UpBlock3D"),
                 block_out_channels: Tuple[int, ...] = (256, 512, 512, 1024), layers_per_block: int = 2,
                 cross_attention_dim: int = 1024, attention_head_dim: int = 8,
                 only_cross_attention: Union[bool, Tuple[bool, ...]] = (True, True, True, False),
                 use_linear_projection: bool = True, num_class_embeds: Optional[int] = 1000,
                 down_temporal_idx=(0, 1, 2), mid_temporal: bool = False, up_temporal_idx=(0, 1, 2),
                 video_condition: bool = False, temporal_module_config: Optional[dict] = None, max_noise_level: int = 350, **kw):
        levels = len(block_out_channels)
        tm = temporal_module_config or {}
        if video_condition or tuple(tm.get("attention_block_types", ("", ""))) != ("", ""):
            raise NotImplementedError("UNet3DVSRModel: video_condition / temporal transformers inside TemporalModule3D are not built")
        temporal = (tuple(down_temporal_idx), bool(mid_temporal), tuple(up_temporal_idx))
        all_on = (tuple(range(levels)), True, tuple(range(levels)))
        none_on = ((), False, ())
        if temporal not in (all_on, none_on):
            raise NotImplementedError("UNet3DVSRModel: TemporalModule3D on every level (vsr/configs/unet_3d_config.json) or on none")
        if not use_linear_projection:
            raise NotImplementedError("UNet3DVSRModel runs with use_linear_projection=True (vsr/configs/unet_3d_config.json)")
        if in_channels % 2 == 0:
            raise NotImplementedError("UNet3DVSRModel: the 4 + 3 channel input (in_channels=7) is the supported layout")
        self._vsr_extra = dict(vsr_temporal_modules=temporal == all_on, num_class_embeds=int(num_class_embeds or 0))
        self.max_noise_level = max_noise_level
        super().__init__(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                         down_block_types=down_block_types, up_block_types=up_block_types, block_out_channels=block_out_channels,
                         layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim,
                         attention_head_dim=attention_head_dim, only_cross_attention=only_cross_attention,
                         use_linear_projection=use_linear_projection, **kw)
        self.config.max_noise_level = max_noise_level
        self.config.num_class_embeds = num_class_embeds

    def _vsr_config(self, only_cross_attention, use_linear_projection, levels: int) -> dict:
        if isinstance(only_cross_attention, bool):                       # vsr/models/unet.py:198-199
            only_cross_attention = (only_cross_attention,) * levels
        if len(only_cross_attention) != levels:
            raise ValueError("only_cross_attention must have one entry per down block")
        return dict(vsr_blocks=True, only_cross_attention=tuple(bool(v) for v in only_cross_attention), **self._vsr_extra)

    # ------------------------------------------------------------------ engine: one zero channel on the way in
    def _config_c(self):
        c = super()._config_c()
        c.in_channels = self.cfg.in_channels + 1
        return c

    def _engine_tensor(self, name: str, t: torch.Tensor) -> torch.Tensor:
        if name == "conv_in.weight":
            return torch.cat([t, torch.zeros_like(t[:, :1])], dim=1).contiguous()
        return t

    # ------------------------------------------------------------------ forward (vsr/models/unet.py:408-600)
    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep, low_res: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                class_labels=20, low_res_clean=None, attention_mask=None, return_dict: bool = True):
        if attention_mask is not None or low_res_clean is not None:
            raise NotImplementedError("attention_mask / low_res_clean are outside the MI355X path")
        if sample.dim() != 5 or low_res.dim() != 5:
            raise ValueError("sample [b, 4, f, h, w] and low_res [b, 3, f, h, w] are required")
        if encoder_hidden_states is None or encoder_hidden_states.dim() != 3:
            raise ValueError("encoder_hidden_states [b, n, cross_attention_dim] is required")
        b, c, f, h, w = sample.shape
        if c + low_res.shape[1] != self.cfg.in_channels or low_res.shape[0] != b or tuple(low_res.shape[2:]) != (f, h, w):
            raise ValueError("sample / low_res shapes do not match the model configuration")
        dev = self.device
        handle = self._ensure_engine()
        n_ctx = encoder_hidden_states.shape[1]
        self.prepare(b, f, h, w, n_ctx)
        x = torch.cat([sample.to(dev, torch.float16), low_res.to(dev, torch.float16),
                       torch.zeros(b, 1, f, h, w, dtype=torch.float16, device=dev)], dim=1).contiguous()       # :453 (+ pad)
        ctx = encoder_hidden_states.to(device=dev, dtype=torch.float16).contiguous()
        if torch.is_tensor(timestep):
            t = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
        else:
            t = torch.tensor([float(timestep)], dtype=torch.float32, device=dev)
        t = t.expand(b).contiguous()
        out = torch.empty(b, self.cfg.out_channels, f, h, w, dtype=torch.float16, device=dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            args = (handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(ctx.data_ptr()))
            if self.cfg.num_class_embeds:
                labels = torch.as_tensor(class_labels).reshape(-1).to("cpu", torch.int64)
                labels = labels.expand(b) if labels.numel() == 1 else labels
                if labels.numel() != b:
                    raise ValueError("class_labels must hold one noise level per video")
                if bool((labels > self.max_noise_level).any()):          # :498-499
                    raise ValueError(f"`noise_level` has to be <= {self.max_noise_level} but is {class_labels}")
                arr = (ctypes.c_int * b)(*[int(v) for v in labels])
                _lib.check(lib.lavie_unet_forward_labels(*args, arr, ctypes.c_void_p(out.data_ptr()), b, f, h, w, n_ctx, stream),
                           "lavie_unet_forward_labels")
            else:
                _lib.check(lib.lavie_unet_forward(*args, ctypes.c_void_p(out.data_ptr()), b, f, h, w, n_ctx, stream),
                           "lavie_unet_forward")
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)
