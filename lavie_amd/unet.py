"""Host-side mirror of `base/models/unet.py` (reference): `UNet3DConditionModel`.

Same constructor keywords (the ones live on the inference path), same state-dict key names and
shapes (unet.py:142-295), same `forward(sample, timestep, encoder_hidden_states, ...).sample`
surface (unet.py:366-375, 509-512) — but the module holds no compute of its own: forward hands the
fp16 parameters and inputs to the gfx950 engine in liblavie_hip.so.  There is no eager/CPU path."""
import ctypes
import json
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch
from torch import nn

from . import _lib, spec
from .config import UNetConfig
from .weights import rotary_freqs


@dataclass
class UNet3DConditionOutput:
    sample: torch.Tensor


def _attach(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    """Registers `param` under the reference's dotted state-dict name, creating plain containers."""
    parts = dotted.split(".")
    mod = root
    for name in parts[:-1]:
        child = mod._modules.get(name)
        if child is None:
            child = nn.Module()
            mod.add_module(name, child)
        mod = child
    mod.register_parameter(parts[-1], param)


class UNet3DConditionModel(nn.Module):
    _supported_down = ("CrossAttnDownBlock3D", "DownBlock3D")
    # transformer-block variant (UNetConfig fields); lavie_amd.interpolation.unet overrides it
    _block_variant = dict(sparse_causal_attn1=False, temporal_plain=False, ff_before_temporal=False)
    _allow_first_frame = False
    _allow_vsr_options = False            # only_cross_attention tuples / use_linear_projection (lavie_amd.vsr.unet)

    def __init__(
        self,
        sample_size: Optional[int] = None,
        in_channels: int = 4,
        out_channels: int = 4,
        center_input_sample: bool = False,
        flip_sin_to_cos: bool = True,
        freq_shift: int = 0,
        down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
        mid_block_type: str = "UNetMidBlock3DCrossAttn",
        up_block_types: Tuple[str, ...] = ("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
        only_cross_attention: bool = False,
        block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280),
        layers_per_block: int = 2,
        downsample_padding: int = 1,
        mid_block_scale_factor: float = 1,
        act_fn: str = "silu",
        norm_num_groups: int = 32,
        norm_eps: float = 1e-5,
        cross_attention_dim: int = 1280,
        attention_head_dim: int = 8,
        dual_cross_attention: bool = False,
        use_linear_projection: bool = False,
        class_embed_type: Optional[str] = None,
        num_class_embeds: Optional[int] = None,
        upcast_attention: bool = False,
        resnet_time_scale_shift: str = "default",
        use_first_frame: bool = False,
        use_relative_position: bool = False,
        init_weights: bool = True,
    ):
        """`init_weights=False` (not a reference keyword) skips the ~1 min CPU default initialisation of the
        909 M parameters when a state dict is about to be loaded anyway."""
        super().__init__()
        # options that leave the benchmarked inference path (SURVEY.md §8a "Not on the path")
        unsupported = {
            "center_input_sample": center_input_sample,
            "only_cross_attention": only_cross_attention and not self._allow_vsr_options,
            "dual_cross_attention": dual_cross_attention,
            "use_linear_projection": use_linear_projection and not self._allow_vsr_options,
            "class_embed_type": class_embed_type, "num_class_embeds": num_class_embeds,
            "upcast_attention": upcast_attention, "use_first_frame": use_first_frame and not self._allow_first_frame,
            "use_relative_position": use_relative_position,
        }
        bad = [k for k, v in unsupported.items() if v]
        if bad or not flip_sin_to_cos or freq_shift != 0 or resnet_time_scale_shift != "default" \
                or act_fn not in ("silu", "swish") or mid_block_type != "UNetMidBlock3DCrossAttn" \
                or downsample_padding != 1 or mid_block_scale_factor != 1 or not isinstance(attention_head_dim, int):
            raise NotImplementedError(f"UNet3DConditionModel option outside the MI355X path: {bad or 'see constructor'}")
        if len(down_block_types) != len(block_out_channels) or len(up_block_types) != len(block_out_channels):
            raise ValueError("down_block_types / up_block_types / block_out_channels must have equal lengths")
        attn = tuple(t == "CrossAttnDownBlock3D" for t in down_block_types)
        for t in down_block_types:
            if t not in self._supported_down:
                raise ValueError(f"{t} does not exist.")
        expect_up = tuple("CrossAttnUpBlock3D" if a else "UpBlock3D" for a in reversed(attn))
        if tuple(up_block_types) != expect_up:
            raise NotImplementedError("up_block_types must mirror down_block_types")

        self.sample_size = sample_size
        self.cfg = UNetConfig(sample_size=sample_size or 64, in_channels=in_channels, out_channels=out_channels,
                              block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                              heads=attention_head_dim, cross_attention_dim=cross_attention_dim,
                              norm_groups=norm_num_groups, norm_eps=norm_eps, attn_levels=attn,
                              **dict(self._block_variant, sparse_causal_attn1=bool(use_first_frame) and self._allow_first_frame),
                              **self._vsr_config(only_cross_attention, use_linear_projection, len(block_out_channels)))
        self.cfg.validate()
        self.config = SimpleNamespace(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            center_input_sample=False, block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
            cross_attention_dim=cross_attention_dim, attention_head_dim=attention_head_dim,
            norm_num_groups=norm_num_groups, norm_eps=norm_eps, class_embed_type=None,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types))
        self.num_upsamplers = len(block_out_channels) - 1

        # parameters under the reference's names; values follow the reference constructor's defaults
        for name, shape in spec.iter_params(self.cfg):
            if name.endswith("rotary_emb.freqs"):       # incl. the VSR model's shared `temporal_rotary_emb.freqs`
                p = nn.Parameter(rotary_freqs(self.cfg.rotary_dim), requires_grad=False)
            else:
                p = nn.Parameter(torch.empty(shape), requires_grad=False)
                if init_weights:
                    self._default_init(name, p.data)
            _attach(self, name, p)

        self._engine = None
        self._engine_key = None
        self._prepared = None
        self._cached_ctx = None
        self._graph = False
        self._graph_io = {}

    def _vsr_config(self, only_cross_attention, use_linear_projection, levels: int) -> dict:
        """UNetConfig fields of the VSR block variant; the base and interpolation models have none."""
        return {}

    # ------------------------------------------------------------------ init / bookkeeping
    @staticmethod
    def _default_init(name: str, t: torch.Tensor) -> None:
        if t.dim() == 1:
            (nn.init.ones_ if name.endswith("weight") else nn.init.zeros_)(t)
            if name.endswith("bias") and "norm" not in name:     # Linear/Conv bias: U(-1/sqrt(fan_in), ..) in torch;
                nn.init.zeros_(t)                                # kept at zero here, checkpoints overwrite it
        elif name.endswith("relative_attention_bias.weight"):
            nn.init.normal_(t)
        elif name.endswith("attn_temp.to_out.0.weight"):
            nn.init.zeros_(t)                                     # attention.py:475
        else:
            fan_in = t[0].numel()
            bound = 1.0 / fan_in ** 0.5
            nn.init.uniform_(t, -bound, bound)

    @property
    def dtype(self) -> torch.dtype:
        return self.conv_in.weight.dtype

    @property
    def device(self) -> torch.device:
        return self.conv_in.weight.device

    def _apply(self, fn, *a, **k):            # .to()/.half()/.cuda() replace parameter storage
        self._drop_engine()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._drop_engine()
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def _drop_engine(self):
        eng = self.__dict__.get("_engine")
        if eng:
            _lib.load().lavie_unet_destroy(eng)
        self.__dict__["_engine"] = None
        self.__dict__["_engine_key"] = None
        self.__dict__["_prepared"] = None
        self.__dict__["_cached_ctx"] = None
        self.__dict__["_graph_io"] = {}

    def __del__(self):
        try:
            self._drop_engine()
        except Exception:
            pass

    # ------------------------------------------------------------------ engine
    def _config_c(self) -> "_lib.UNetConfigC":
        c = _lib.UNetConfigC()
        c.struct_size = ctypes.sizeof(_lib.UNetConfigC)
        cfg = self.cfg
        c.in_channels, c.out_channels = cfg.in_channels, cfg.out_channels
        c.num_levels = len(cfg.block_out_channels)
        for i, (w, a) in enumerate(zip(cfg.block_out_channels, cfg.attn_levels)):
            c.block_out_channels[i] = w
            c.attn_levels[i] = int(a)
        c.layers_per_block, c.heads = cfg.layers_per_block, cfg.heads
        c.cross_attention_dim, c.norm_groups, c.norm_eps = cfg.cross_attention_dim, cfg.norm_groups, cfg.norm_eps
        c.rotary_dim, c.rel_buckets, c.rel_max_distance = cfg.rotary_dim, cfg.rel_buckets, cfg.rel_max_distance
        c.sparse_causal_attn1, c.temporal_plain = int(cfg.sparse_causal_attn1), int(cfg.temporal_plain)
        c.ff_before_temporal = int(cfg.ff_before_temporal)
        c.vsr_blocks = int(cfg.vsr_blocks)
        for i, v in enumerate(cfg.only_cross_attention):
            c.only_cross_attention[i] = int(v)
        c.vsr_temporal_modules, c.num_class_embeds = int(cfg.vsr_temporal_modules), int(cfg.num_class_embeds)
        return c

    def _ensure_engine(self):
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("UNet3DConditionModel runs on MI355X only: move the model to a HIP device "
                               "(model.to('cuda', torch.float16)); there is no CPU path")
        if self.dtype != torch.float16:
            raise RuntimeError("UNet3DConditionModel computes in fp16: call .half() / .to(dtype=torch.float16)")
        if self._engine is not None:        # invalidated by _apply()/load_state_dict(); see refresh_engine()
            return self._engine
        key = dev.index
        lib = _lib.load()
        handle = ctypes.c_void_p()
        cfg_c = self._config_c()
        with torch.cuda.device(dev):
            _lib.check(lib.lavie_unet_create(ctypes.byref(cfg_c), ctypes.byref(handle)), "lavie_unet_create")
            try:
                stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                keep = []            # tensors made for the engine (contiguous / padded copies) stay alive until finalize
                for name, p in self.named_parameters():
                    t = self._engine_tensor(name, p.data if p.data.is_contiguous() else p.data.contiguous())
                    keep.append(t)
                    _lib.check(lib.lavie_unet_set_param(handle, name.encode(), ctypes.c_void_p(t.data_ptr()), t.numel()),
                               f"lavie_unet_set_param({name})")
                _lib.check(lib.lavie_unet_finalize(handle, stream), "lavie_unet_finalize")
                torch.cuda.current_stream().synchronize()
            except Exception:
                lib.lavie_unet_destroy(handle)
                raise
        self.__dict__["_engine"] = handle
        self.__dict__["_engine_key"] = key
        return handle

    def _engine_tensor(self, name: str, t: torch.Tensor) -> torch.Tensor:
        """Hook: the tensor handed to the engine for state-dict entry `name` (identity here)."""
        return t

    def refresh_engine(self):
        """Re-packs the weights after parameters were modified in place."""
        self._drop_engine()
        return self._ensure_engine()

    def engine_handle(self):
        """The `lavie_unet_t` behind this module (built on first use)."""
        return self._ensure_engine()

    def prepare(self, batch: int, frames: int, height: int, width: int, ctx_len: int = 77) -> None:
        """Sizes the engine workspace for latents [batch, C, frames, height, width] (allocates)."""
        handle = self._ensure_engine()
        want = (batch, frames, height, width, ctx_len)
        if self._prepared == want:
            return
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().lavie_unet_prepare(handle, *want), "lavie_unet_prepare")
        self.__dict__["_prepared"] = want

    def cache_context(self, encoder_hidden_states: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """Computes the text keys / values of every transformer block once for this context (lavie_unet_cache_context) and
        returns the fp16 device tensor to pass as `encoder_hidden_states` while it is cached — a denoise loop calls this
        before its first step and `cache_context(None)` after its last.  The tensor must not be modified meanwhile."""
        handle = self._ensure_engine()
        lib = _lib.load()
        with torch.cuda.device(self.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            if encoder_hidden_states is None:
                self.__dict__["_cached_ctx"] = None
                _lib.check(lib.lavie_unet_cache_context(handle, None, 0, 0, stream), "lavie_unet_cache_context")
                return None
            if self._prepared is None:
                raise RuntimeError("cache_context: call prepare(batch, frames, height, width, ctx_len) first")
            ctx = encoder_hidden_states.to(device=self.device, dtype=torch.float16).contiguous()
            _lib.check(lib.lavie_unet_cache_context(handle, ctypes.c_void_p(ctx.data_ptr()), ctx.shape[0], ctx.shape[1], stream),
                       "lavie_unet_cache_context")
        self.__dict__["_cached_ctx"] = ctx          # keeps the tensor (and so its address) alive while cached
        return ctx

    def set_cfg_shared_input(self, on: bool) -> None:
        """Classifier-free guidance feeds the UNet `torch.cat([latents] * 2)` (pipeline_videogen.py:666): with on=True the caller
        vouches that sample[b] == sample[b + B/2] (and the timesteps likewise) in the forwards that follow, and the engine computes
        the layers in front of the first text cross-attention once (lavie_unet_set_cfg_shared_input).  The denoise loops that
        build the duplicated input themselves switch it on around their steps and off afterwards."""
        handle = self._ensure_engine()
        _lib.check(_lib.load().lavie_unet_set_cfg_shared_input(handle, 1 if on else 0), "lavie_unet_set_cfg_shared_input")

    def enable_graph(self, on: bool = True) -> None:
        """Replay the forward from a hipGraph (lavie_unet_forward_graph).  While on, the timestep and the output live in
        per-shape buffers owned by this module, so the captured addresses repeat from call to call: the returned tensor is
        OVERWRITTEN by the next forward of the same shape (a denoise loop consumes it at once), and `sample` /
        `encoder_hidden_states` should be the same fp16 device tensors on every step (new addresses only cost a re-capture)."""
        self.__dict__["_graph"] = bool(on)
        if not on:
            self.__dict__["_graph_io"] = {}

    # ------------------------------------------------------------------ forward (unet.py:366-512)
    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                encoder_hidden_states: torch.Tensor = None, class_labels: Optional[torch.Tensor] = None,
                attention_mask: Optional[torch.Tensor] = None, use_image_num: int = 0, return_dict: bool = True):
        if class_labels is not None or attention_mask is not None or use_image_num:
            raise NotImplementedError("class_labels / attention_mask / use_image_num are outside the MI355X path")
        if sample.dim() != 5:
            raise ValueError(f"Expected sample to have ndim=5 [b, c, f, h, w], got ndim={sample.dim()}")
        if encoder_hidden_states is None or encoder_hidden_states.dim() != 3:
            raise ValueError("encoder_hidden_states [b, n, cross_attention_dim] is required")
        b, c, f, h, w = sample.shape
        if c != self.cfg.in_channels or encoder_hidden_states.shape[0] != b \
                or encoder_hidden_states.shape[2] != self.cfg.cross_attention_dim:
            raise ValueError("sample / encoder_hidden_states shapes do not match the model configuration")
        handle = self._ensure_engine()
        dev = self.device
        n_ctx = encoder_hidden_states.shape[1]
        self.prepare(b, f, h, w, n_ctx)

        x = sample.to(device=dev, dtype=torch.float16).contiguous()
        ctx = encoder_hidden_states.to(device=dev, dtype=torch.float16).contiguous()
        if torch.is_tensor(timestep):
            t = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
        else:
            t = torch.tensor([float(timestep)], dtype=torch.float32, device=dev)
        lib = _lib.load()
        if self._graph:
            io = self._graph_io.get((b, f, h, w))
            if io is None:
                io = (torch.empty(b, dtype=torch.float32, device=dev),
                      torch.empty(b, self.cfg.out_channels, f, h, w, dtype=torch.float16, device=dev))
                self._graph_io[(b, f, h, w)] = io
            io[0].copy_(t.expand(b))                       # unet.py:426; contents change, the address does not
            t, out = io
            entry, what = lib.lavie_unet_forward_graph, "lavie_unet_forward_graph"
        else:
            t = t.expand(b).contiguous()                   # unet.py:426
            out = torch.empty(b, self.cfg.out_channels, f, h, w, dtype=torch.float16, device=dev)
            entry, what = lib.lavie_unet_forward, "lavie_unet_forward"
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(entry(handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(t.data_ptr()),
                             ctypes.c_void_p(ctx.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                             b, f, h, w, n_ctx, stream), what)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)

    # ------------------------------------------------------------------ checkpoint contract (unet.py:540-588)
    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path: str, subfolder: Optional[str] = None):
        """SD-1.x `unet/config.json` + `diffusion_pytorch_model.bin`; temporal ('_temp') tensors keep this
        model's own initialisation, exactly as the reference does before `lavie_base.pt` is loaded on top."""
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")
        with open(config_file, "r") as fh:
            config = json.load(fh)
        keep = ("sample_size", "in_channels", "out_channels", "block_out_channels", "layers_per_block",
                "norm_num_groups", "norm_eps", "cross_attention_dim", "attention_head_dim")
        model = cls(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in config.items() if k in keep})
        model_file = os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin")
        if not os.path.isfile(model_file):
            raise RuntimeError(f"{model_file} does not exist")
        state = torch.load(model_file, map_location="cpu", weights_only=True)
        for k, v in model.state_dict().items():
            if "_temp" in k:
                state[k] = v
        model.load_state_dict(state)
        return model
