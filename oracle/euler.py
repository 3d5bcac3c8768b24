"""CPU restatement of the Euler (discrete) sampler loop (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference takes `EulerDiscreteScheduler` from diffusers 0.16.0 (base/pipelines/sample.py:50-55); neither the package
nor a vendored copy is in the reference tree and the reference holds no fixture for it: PARITY UNPINNED.  Restated from
the published algorithm (Karras et al. 2022, Alg. 2 with churn 0, on the discrete DDPM noise levels), written
independently of `lavie_amd/scheduling_euler_discrete.py` (closed-form x_prev, float64 tables)."""
from typing import Callable, Optional

import numpy as np
import torch


class EulerSchedule:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02):
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        abar = torch.cumprod(1.0 - betas, dim=0).double().numpy()
        self.train_sigmas = np.sqrt((1 - abar) / abar)
        self.n_train = num_train_timesteps

    def set_timesteps(self, n: int):
        self.timesteps = np.linspace(0, self.n_train - 1, n)[::-1].copy()
        lo = np.floor(self.timesteps).astype(int)
        hi = np.minimum(lo + 1, self.n_train - 1)
        w = self.timesteps - lo
        self.sigmas = np.append((1 - w) * self.train_sigmas[lo] + w * self.train_sigmas[hi], 0.0)
        self.init_noise_sigma = float(self.sigmas.max())


def cfg_euler_loop(unet: Callable, latents: torch.Tensor, prompt_embeds, negative_embeds, num_steps: int = 50,
                   guidance_scale: float = 7.5, max_steps: Optional[int] = None):
    """VideoGenPipeline.__call__ steps 4-7 (pipeline_videogen.py:641-689) with the Euler scheduler: latents are scaled by
    init_noise_sigma (:509), the model sees x / sqrt(sigma^2 + 1) (:667), guidance with the unconditional half first
    (:679), x <- x + (sigma_next - sigma) eps."""
    sch = EulerSchedule()
    sch.set_timesteps(num_steps)
    ctx = torch.cat([negative_embeds, prompt_embeds], dim=0)
    x = latents * sch.init_noise_sigma
    for i, t in enumerate(sch.timesteps):
        if max_steps is not None and i >= max_steps:
            break
        sigma, sigma_next = float(sch.sigmas[i]), float(sch.sigmas[i + 1])
        xin = x / (sigma ** 2 + 1) ** 0.5
        eps = unet(torch.cat([xin, xin], dim=0), float(t), ctx)
        e_u, e_c = eps[0:1], eps[1:2]
        x = x + (sigma_next - sigma) * (e_u + guidance_scale * (e_c - e_u))
    return x
