"""Host-side mirror of `interpolation/models/unet.py` (reference): the frame-interpolation `UNet3DConditionModel`.

Same engine and state-dict contract as `lavie_amd.unet`, with the three block deltas of the interpolation model
(interpolation/models/attention.py:456-606): `use_first_frame` turns attn1 into SparseCausalAttention (keys/values =
first frame || previous frame, :609-665), the block order is spatial -> text -> feed-forward -> temporal (:566-606),
and attn_temp is the plain CrossAttention over frames (no rotary embedding, no relative-position bias, no such
tensors in the state dict; :525-533).  `from_pretrained_2d(use_concat=True, copy_no_mask=True)` widens conv_in to
8 channels exactly as interpolation/models/unet.py:477-555 does."""
import json
import os
from typing import Optional, Union

import torch

from ..unet import UNet3DConditionModel as _BaseUNet
from ..unet import UNet3DConditionOutput


class UNet3DConditionModel(_BaseUNet):
    _block_variant = dict(sparse_causal_attn1=False, temporal_plain=True, ff_before_temporal=True)
    _allow_first_frame = True

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                encoder_hidden_states: torch.Tensor = None, class_labels: Optional[torch.Tensor] = None,
                attention_mask: Optional[torch.Tensor] = None, return_dict: bool = True):
        """interpolation/models/unet.py:311-452 (no `use_image_num` keyword on this model)."""
        return super().forward(sample, timestep, encoder_hidden_states, class_labels=class_labels,
                               attention_mask=attention_mask, return_dict=return_dict)

    @torch.no_grad()
    def forward_with_cfg(self, x, t, encoder_hidden_states=None, class_labels: Optional[torch.Tensor] = None,
                         cfg_scale: float = 4.0):
        """interpolation/models/unet.py:454-474: the first half of the batch is run twice (conditional text first,
        unconditional second), guidance is applied to the first four output channels and both halves return the
        guided prediction."""
        half = x[: len(x) // 2]
        combined = torch.cat([half, half], dim=0)
        model_out = self.forward(combined, t, encoder_hidden_states, class_labels).sample
        eps, rest = model_out[:, :4], model_out[:, 4:]
        cond_eps, uncond_eps = torch.split(eps.float(), len(eps) // 2, dim=0)
        half_eps = uncond_eps + cfg_scale * (cond_eps - uncond_eps)
        eps = torch.cat([half_eps, half_eps], dim=0)
        return torch.cat([eps, rest.float()], dim=1)

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path: str, subfolder: Optional[str] = None, use_concat: bool = False,
                           copy_no_mask: bool = False):
        """SD-1.x `unet/config.json` + `diffusion_pytorch_model.bin` (interpolation/models/unet.py:477-555): forces
        `use_first_frame`, widens conv_in to 8 (copy_no_mask) or 9 (use_concat) input channels with the 2-D weights in
        the first four and zeros elsewhere, and keeps this model's own initialisation for every `_temp.` tensor."""
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")
        with open(config_file, "r") as fh:
            config = json.load(fh)
        keep = ("sample_size", "in_channels", "out_channels", "block_out_channels", "layers_per_block",
                "norm_num_groups", "norm_eps", "cross_attention_dim", "attention_head_dim")
        kwargs = {k: (tuple(v) if isinstance(v, list) else v) for k, v in config.items() if k in keep}
        kwargs["use_first_frame"] = True
        if copy_no_mask:
            kwargs["in_channels"] = 8
        elif use_concat:
            kwargs["in_channels"] = 9
        model = cls(**kwargs)
        model_file = os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin")
        if not os.path.isfile(model_file):
            raise RuntimeError(f"{model_file} does not exist")
        state = torch.load(model_file, map_location="cpu", weights_only=True)
        own = model.state_dict()
        if use_concat:
            w2d = state["conv_in.weight"]
            wide = torch.zeros((w2d.shape[0], kwargs["in_channels"], *w2d.shape[2:]), dtype=w2d.dtype)
            wide[:, :4] = w2d[:, :4]
            new_state = {"conv_in.weight": wide, "conv_in.bias": state["conv_in.bias"]}
            for k, v in own.items():          # :537-543 — every other tensor keeps the constructor's value
                if "conv_in" not in k:
                    new_state[k] = v
            model.load_state_dict(new_state)
        else:
            for k, v in own.items():
                if "_temp." in k:
                    state[k] = v
            model.load_state_dict(state)
        return model


__all__ = ["UNet3DConditionModel", "UNet3DConditionOutput"]
