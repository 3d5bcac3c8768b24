"""DDPM scheduler with the surface `VideoGenPipeline` uses (pipeline_videogen.py:112-137, 431-446,
509, 641-642, 662, 667, 683): set_timesteps / timesteps / scale_model_input / step(...).prev_sample /
init_noise_sigma / order / config.

The reference gets this object from diffusers 0.16.0 (`DDPMScheduler.from_pretrained(sd_path,
subfolder="scheduler", beta_start, beta_end, beta_schedule)`, base/pipelines/sample.py:56-61); the
package is not part of the reference tree, so the arithmetic below is the published DDPM ancestral
step for a strided schedule (epsilon prediction, variance "fixed_small", no sample clipping — SD-1.4's
scheduler_config.json carries clip_sample=false).  `coefficients(t)` exposes the five scalars of one
step so the hot loop can run the fused HIP kernel (lavie_cfg_ddpm_step) instead of `step`."""
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch


@dataclass
class DDPMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


def randn_tensor(shape, generator=None, device=None, dtype=torch.float32):
    """Noise helper with diffusers' contract (pipeline_videogen.py:504): a CPU generator draws on the
    CPU and the result is moved, so trajectories are reproducible across devices."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    if isinstance(generator, (list, tuple)):               # one generator per batch entry, results concatenated
        if len(generator) != shape[0]:
            raise ValueError(f"got {len(generator)} generators for a batch of {shape[0]}")
        return torch.cat([randn_tensor((1,) + tuple(shape[1:]), g, device, dtype) for g in generator], dim=0)
    gen_dev = generator.device.type if generator is not None else device.type
    if gen_dev == "cpu" and device.type != "cpu":
        return torch.randn(shape, generator=generator, dtype=dtype).to(device)
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)


class DDPMScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02,
                 beta_schedule: str = "linear", variance_type: str = "fixed_small", clip_sample: bool = False,
                 prediction_type: str = "epsilon", steps_offset: int = 0):
        if variance_type != "fixed_small" or clip_sample or prediction_type != "epsilon":
            raise NotImplementedError("only epsilon prediction, fixed_small variance, clip_sample=False are supported")
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "squaredcos_cap_v2":
            # the cosine schedule of the x4-upscaler's low-res scheduler; in-tree statement of the same function:
            # interpolation/diffusion/gaussian_diffusion.py:116-140 (betas_for_alpha_bar, max_beta 0.999)
            import math
            bar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
            betas = torch.tensor([min(1 - bar((i + 1) / num_train_timesteps) / bar(i / num_train_timesteps), 0.999)
                                  for i in range(num_train_timesteps)], dtype=torch.float32)
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for DDPMScheduler")
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, variance_type=variance_type, clip_sample=clip_sample,
                                      prediction_type=prediction_type, steps_offset=steps_offset)

    def set_timesteps(self, num_inference_steps: int, device: Union[str, torch.device, None] = None):
        n_train = self.config.num_train_timesteps
        if num_inference_steps > n_train:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than {n_train}")
        self.num_inference_steps = num_inference_steps
        ratio = n_train // num_inference_steps
        steps = (torch.arange(0, num_inference_steps, dtype=torch.int64) * ratio).flip(0)      # 980, 960, ..., 0
        self.timesteps = steps.to(device) if device is not None else steps

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """q(x_t | x_0) = sqrt(abar_t) x_0 + sqrt(1 - abar_t) noise, per batch entry — the low-res conditioning of the VSR
        pipeline (pipeline_stable_diffusion_upscale_video_3d.py:631-633)."""
        ab = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        t = torch.as_tensor(timesteps, device=original_samples.device).reshape(-1)
        shape = (-1,) + (1,) * (original_samples.dim() - 1)
        return ab[t].sqrt().reshape(shape) * original_samples + (1 - ab[t]).sqrt().reshape(shape) * noise

    def coefficients(self, timestep: int) -> Tuple[float, float, float, float, float]:
        """(k_x, k_eps, c_x0, c_xt, sigma): x0 = k_x x - k_eps eps;  x_prev = c_x0 x0 + c_xt x + sigma z."""
        t = int(timestep)
        steps = self.num_inference_steps or self.config.num_train_timesteps
        prev = t - self.config.num_train_timesteps // steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev]) if prev >= 0 else 1.0
        b_t, b_prev = 1.0 - a_t, 1.0 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1.0 - cur_alpha
        c_x0 = (a_prev ** 0.5) * cur_beta / b_t
        c_xt = (cur_alpha ** 0.5) * b_prev / b_t
        var = max(b_prev / b_t * cur_beta, 1e-20)
        sigma = var ** 0.5 if t > 0 else 0.0
        return 1.0 / a_t ** 0.5, (b_t ** 0.5) / a_t ** 0.5, c_x0, c_xt, sigma

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator=None,
             return_dict: bool = True):
        """Generic (stock-PyTorch) form of one ancestral step; the pipeline's hot loop uses the fused kernel."""
        t = int(timestep)
        k_x, k_e, c_x0, c_xt, sigma = self.coefficients(t)
        x0 = k_x * sample - k_e * model_output
        prev = c_x0 * x0 + c_xt * sample
        if t > 0:
            noise = randn_tensor(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            prev = prev + sigma * noise
        if not return_dict:
            return (prev,)
        return DDPMSchedulerOutput(prev_sample=prev, pred_original_sample=x0)
