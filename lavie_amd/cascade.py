"""The three-stage cascade of the reference as one on-device driver (BASELINE.json configs[4]): base T2V
(16 x 320 x 512) -> frame interpolation (61 frames) -> video super-resolution (61 x 1280 x 2048).

The reference runs three scripts that hand mp4 files to each other (base/pipelines/sample.py, interpolation/sample.py,
vsr/sample.py); the tensors that cross a stage boundary are the decoded frames in [-1, 1], so the driver keeps them on the
device instead of writing files.  Each step cites the script line it reproduces.  The UNets and samplers are the HIP-path
objects of `lavie_amd`; text encoders and VAEs are stock PyTorch-ROCm objects passed in by the caller (outside the latents
metric)."""
from typing import Optional

import numpy as np
import torch

from .vsr.pipeline import upscale_in_chunks


def decode_frames(vae, latents: torch.Tensor, scaling: Optional[float] = None, chunk: int = 4) -> torch.Tensor:
    """[b, 4, f, h, w] latents -> frames [b, 3, f, H, W] in [-1, 1] (pipeline_videogen.py:422-427; vsr pipeline :354-358)."""
    scaling = vae.config.scaling_factor if scaling is None else scaling
    b, c, f, h, w = latents.shape
    flat = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w).to(next(vae.parameters()).dtype) / scaling
    out = torch.cat([vae.decode(flat[i:i + chunk]).sample.clamp(-1, 1) for i in range(0, b * f, chunk)], dim=0)
    return out.reshape(b, f, *out.shape[1:]).permute(0, 2, 1, 3, 4)


def interpolation_condition(vae, frames: torch.Tensor, num_frames: int = 61, generator=None) -> torch.Tensor:
    """interpolation/sample.py:74-79, 135-149: the 16 base frames are resampled to `num_frames` by index
    (np.linspace(0, 15, num_frames, dtype=int)), VAE-encoded (x 0.18215), every 4th latent frame is kept and repeated 4x, and
    frames [1:-2] of that are the `copied_video` conditioning.  frames: [b, 3, 16, H, W] in [-1, 1]."""
    b, _, f, H, W = frames.shape
    idx = torch.from_numpy(np.linspace(0, f - 1, num_frames, dtype=int)).to(frames.device)
    vid = frames.index_select(2, idx).permute(0, 2, 1, 3, 4).reshape(b * num_frames, 3, H, W).to(next(vae.parameters()).dtype)
    lat = torch.cat([vae.encode(vid[i:i + 8]).latent_dist.sample(generator) for i in range(0, b * num_frames, 8)], dim=0) * 0.18215
    lat = lat.reshape(b, num_frames, *lat.shape[1:]).permute(0, 2, 1, 3, 4)
    lr = torch.arange(0, num_frames + 1, 4, device=lat.device)
    lr = lr[lr < num_frames]
    copied = torch.repeat_interleave(lat.index_select(2, lr), 4, dim=2)[:, :, 1:-2]
    return copied.float()


@torch.no_grad()
def text_to_video_cascade(base_pipe, interp_unet, interp_diffusion, vsr_pipe, vae, vsr_vae, prompt_embeds, negative_prompt_embeds,
                          vsr_prompt_embeds, vsr_negative_prompt_embeds, interp_prompt_embeds, interp_negative_prompt_embeds,
                          height: int = 320, width: int = 512, base_steps: int = 50, guidance_scale: float = 7.5,
                          interp_frames: int = 61, interp_cfg_scale: float = 4.0, vsr_steps: int = 50,
                          vsr_guidance_scale: float = 7.5, noise_level: int = 150, generator=None, decode_final: bool = True):
    """Returns (base_latents, interp_latents, vsr_latents, frames or None).  Text embeddings are passed per stage (the base
    and interpolation stages use SD-1.4's CLIP, 768 wide; the VSR stage the x4-upscaler's, 1024 wide)."""
    dev = base_pipe.device
    # 1. base T2V (base/pipelines/sample.py:78-91)
    base = base_pipe(prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds, height=height, width=width,
                     video_length=16, num_inference_steps=base_steps, guidance_scale=guidance_scale, generator=generator,
                     output_type="latent").video
    frames16 = decode_frames(vae, base, 0.18215)
    # 2. interpolation 16 -> 61 frames (interpolation/sample.py:135-174)
    copied = interpolation_condition(vae, frames16, interp_frames)
    z = torch.randn(1, 4, interp_frames, height // 8, width // 8, device=dev)
    z2, copied2 = torch.cat([z] * 2), torch.cat([copied] * 2)
    ctx = torch.cat([interp_prompt_embeds, interp_negative_prompt_embeds], dim=0)                 # prompt first (:157)
    interp = interp_diffusion.ddim_sample_loop(
        interp_unet.forward_with_cfg, z2.shape, z2, clip_denoised=False,
        model_kwargs=dict(encoder_hidden_states=ctx, class_labels=None, cfg_scale=interp_cfg_scale), device=dev, mask=None,
        x_start=copied2, use_concat=True, copy_no_mask=True).chunk(2, dim=0)[0]
    frames61 = decode_frames(vae, interp, 0.18215)
    # 3. video super-resolution in 8-frame chunks (vsr/sample.py:90-123): the decoded frames are the low-res conditioning
    up = upscale_in_chunks(vsr_pipe, frames61, short_seq=8, prompt_embeds=vsr_prompt_embeds,
                           negative_prompt_embeds=vsr_negative_prompt_embeds, num_inference_steps=vsr_steps,
                           guidance_scale=vsr_guidance_scale, noise_level=noise_level, generator=generator)
    frames = decode_frames(vsr_vae, up, None, chunk=1) if decode_final else None
    return base, interp, up, frames
