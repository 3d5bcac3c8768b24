"""fp32 CPU restatement of the DDPM sampler step and the CFG denoise loop (TEST INFRASTRUCTURE).

The scheduler is third-party (diffusers 0.16.0 `DDPMScheduler`, constructed at
/root/reference/base/pipelines/sample.py:56-61 with beta linear 1e-4 -> 0.02, 1000 train steps;
SD-1.4's scheduler_config carries clip_sample=false, variance_type defaults to "fixed_small").
Its source is not under /root/reference and the package is not installed: PARITY UNPINNED.
The restated algorithm is the published DDPM ancestral step generalised to a strided schedule;
`tests/test_oracle_ddpm.py` cross-checks the posterior-mean coefficients against the reference's
independent in-tree implementation (interpolation/diffusion/gaussian_diffusion.py:232-252, 362-394).
"""
from typing import Callable, List, Optional

import torch


class DDPMSchedule:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02):
        self.num_train_timesteps = num_train_timesteps
        self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps: List[int] = []

    def set_timesteps(self, n: int):
        """`set_timesteps(50)` -> 980, 960, ..., 0 (pipeline_videogen.py:641-642)."""
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        self.timesteps = [i * ratio for i in range(n)][::-1]

    def coefficients(self, t: int):
        """(c_x0_from_xt, c_x0_from_eps, c_mean_x0, c_mean_xt, sigma) for one step t -> t - stride."""
        stride = self.num_train_timesteps // (self.num_inference_steps or self.num_train_timesteps)
        prev = t - stride
        a_t = self.alphas_cumprod[t].item()
        a_prev = self.alphas_cumprod[prev].item() if prev >= 0 else 1.0
        b_t, b_prev = 1.0 - a_t, 1.0 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1.0 - cur_alpha
        c_x0 = (a_prev ** 0.5) * cur_beta / b_t
        c_xt = (cur_alpha ** 0.5) * b_prev / b_t
        var = max(b_prev / b_t * cur_beta, 1e-20)                  # fixed_small, clamped
        sigma = var ** 0.5 if t > 0 else 0.0
        return 1.0 / a_t ** 0.5, (b_t ** 0.5) / a_t ** 0.5, c_x0, c_xt, sigma

    def step(self, eps: torch.Tensor, t: int, x: torch.Tensor, noise: Optional[torch.Tensor]) -> torch.Tensor:
        """x_{t-stride} = c0 * x0_hat + ct * x_t + sigma * noise, x0_hat = (x_t - sqrt(1-abar) eps)/sqrt(abar)."""
        k_x, k_e, c_x0, c_xt, sigma = self.coefficients(t)
        x0 = k_x * x - k_e * eps
        out = c_x0 * x0 + c_xt * x
        if t > 0:
            out = out + sigma * noise
        return out


def cfg_denoise_loop(unet: Callable, latents: torch.Tensor, prompt_embeds: torch.Tensor,
                     negative_embeds: torch.Tensor, noises: List[torch.Tensor], num_steps: int = 50,
                     guidance_scale: float = 7.5, schedule: Optional[DDPMSchedule] = None,
                     max_steps: Optional[int] = None):
    """VideoGenPipeline.__call__ steps 4-7 (pipeline_videogen.py:641-689) for one prompt:
    `unet(x[2,...], t, ctx[2,77,768]) -> eps[2,...]`; CFG `u + s (c - u)` with the UNCONDITIONAL
    half first (line 679: chunk(2) of cat([negative, prompt])); DDPM step with caller-supplied noise."""
    sch = schedule or DDPMSchedule()
    sch.set_timesteps(num_steps)
    ctx = torch.cat([negative_embeds, prompt_embeds], dim=0)
    x = latents * sch.init_noise_sigma
    for i, t in enumerate(sch.timesteps):
        if max_steps is not None and i >= max_steps:
            break
        eps = unet(torch.cat([x, x], dim=0), t, ctx)
        e_u, e_c = eps[0:1], eps[1:2]
        guided = e_u + guidance_scale * (e_c - e_u)
        x = sch.step(guided, t, x, noises[i] if t > 0 else None)
    return x
