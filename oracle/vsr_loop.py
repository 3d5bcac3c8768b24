"""CPU restatement of the VSR stage's denoise loop (TEST INFRASTRUCTURE, see oracle/__init__.py):
pipeline_stable_diffusion_upscale_video_3d.py:629-735 with the DDIM scheduler `vsr/sample.py:48-52` installs.

Pinned parts: the UNet (oracle/vsr_blocks.py, against the imported UNet3DVSRModel) and the DDIM step (oracle/ddim.py,
against the reference's vendored class).  Parity unpinned: the low-res scheduler's cosine beta table is the x4-upscaler's
`low_res_scheduler` config, which is not part of the reference tree; restated here independently of the product
(closed form on float64) from the published schedule, the same function the reference states in-tree at
interpolation/diffusion/gaussian_diffusion.py:116-140."""
import math
from typing import Callable, List, Optional

import numpy as np
import torch

from .ddim import DDIMSchedule


def cosine_alphas_cumprod(n: int = 1000, max_beta: float = 0.999) -> np.ndarray:
    bar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = np.array([min(1 - bar((i + 1) / n) / bar(i / n), max_beta) for i in range(n)], dtype=np.float64)
    return np.cumprod(1.0 - betas)


def add_low_res_noise(image: torch.Tensor, noise: torch.Tensor, noise_level: int) -> torch.Tensor:
    """:629-633: q(x_t | x_0) of the low-res DDPM scheduler at t = noise_level."""
    ab = float(cosine_alphas_cumprod()[noise_level])
    return ab ** 0.5 * image + (1 - ab) ** 0.5 * noise


def vsr_denoise_loop(unet: Callable, latents, image_noised, prompt_embeds, negative_embeds, noise_level: int,
                     num_steps: int, guidance_scale: float, eta: float = 0.0, noises: Optional[List[torch.Tensor]] = None,
                     schedule: Optional[DDIMSchedule] = None):
    """:706-735: `unet(x4, low_res3, t, ctx, labels) -> eps`; negative half first (:640-641, 721-723)."""
    sch = schedule or DDIMSchedule()
    sch.set_timesteps(num_steps)
    ctx = torch.cat([negative_embeds, prompt_embeds], dim=0)
    low = torch.cat([image_noised, image_noised], dim=0)
    labels = torch.full((2 * latents.shape[0],), noise_level, dtype=torch.long)
    x = latents
    for i, t in enumerate(sch.timesteps):
        eps = unet(torch.cat([x, x], dim=0), low, t, ctx, labels)
        e_u, e_c = eps.chunk(2)
        x = sch.step(e_u + guidance_scale * (e_c - e_u), t, x, eta=eta, noise=None if noises is None else noises[i])
    return x
