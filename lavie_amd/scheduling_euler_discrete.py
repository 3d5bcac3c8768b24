"""Euler (discrete) scheduler with the surface `VideoGenPipeline` uses — the `sample_method == 'eulerdiscrete'` branch of
the reference (base/pipelines/sample.py:50-55: `EulerDiscreteScheduler.from_pretrained(sd_path, subfolder="scheduler",
beta_start, beta_end, beta_schedule)`; call sites pipeline_videogen.py:509, 641-642, 667, 683).

The class is diffusers 0.16.0's; neither the package nor a vendored copy is in the reference tree, so this file restates the
published algorithm (Karras et al. 2022, Algorithm 2 without churn, on the discrete DDPM noise levels) with the defaults
`from_pretrained` leaves in place: epsilon prediction, linear timestep interpolation, no Karras sigmas, s_churn = 0.
PARITY UNPINNED (no source, fixture or test for it inside the reference).

  sigma_j = sqrt((1 - abar_j) / abar_j);  timesteps = linspace(0, T-1, n)[::-1] (fractional), sigmas interpolated there, then 0
  init_noise_sigma = max sigma;  scale_model_input(x, t) = x / sqrt(sigma_t^2 + 1)
  step: x0 = x - sigma eps;  x_prev = x + (sigma_next - sigma) (x - x0) / sigma

`coefficients(t)` gives the step in the fused HIP kernel's form: k_x = 1, k_eps = sigma, c_x0 = 1 - sigma_next / sigma,
c_xt = sigma_next / sigma, no noise; `model_input_scale(t)` is the factor the kernel applies to the fp16 UNet input."""
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import numpy as np
import torch


@dataclass
class EulerDiscreteSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


class EulerDiscreteScheduler:
    order = 1
    fractional_timesteps = True          # timesteps are floats (linspace), passed to the UNet as they are

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02,
                 beta_schedule: str = "linear", prediction_type: str = "epsilon", interpolation_type: str = "linear",
                 use_karras_sigmas: bool = False):
        if prediction_type != "epsilon" or interpolation_type != "linear" or use_karras_sigmas:
            raise NotImplementedError("only epsilon prediction with linear sigma interpolation is supported")
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for EulerDiscreteScheduler")
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self._train_sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy().astype(np.float64)
        self.num_inference_steps: Optional[int] = None
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type=prediction_type,
                                      interpolation_type=interpolation_type, use_karras_sigmas=use_karras_sigmas)
        self.set_timesteps(num_train_timesteps)
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps: int, device: Union[str, torch.device, None] = None):
        n_train = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(0, n_train - 1, num_inference_steps, dtype=np.float64)[::-1].copy()
        sig = np.interp(ts, np.arange(0, n_train), self._train_sigmas)
        self._sigmas = np.concatenate([sig, [0.0]])
        self.sigmas = torch.from_numpy(self._sigmas.astype(np.float32))
        self.timesteps = torch.from_numpy(ts.astype(np.float32))
        if device is not None:
            self.sigmas, self.timesteps = self.sigmas.to(device), self.timesteps.to(device)
        self.init_noise_sigma = float(self._sigmas.max())

    def _index(self, timestep) -> int:
        t = float(timestep)
        idx = int(np.argmin(np.abs(self.timesteps.cpu().numpy().astype(np.float64) - t)))
        if abs(float(self.timesteps[idx]) - t) > 1e-3:
            raise ValueError(f"timestep {t} is not one of the scheduler's timesteps: call set_timesteps first")
        return idx

    def model_input_scale(self, timestep) -> float:
        sigma = self._sigmas[self._index(timestep)]
        return float(1.0 / np.sqrt(sigma ** 2 + 1.0))

    def scale_model_input(self, sample: torch.Tensor, timestep) -> torch.Tensor:
        return sample * self.model_input_scale(timestep)

    def coefficients(self, timestep) -> Tuple[float, float, float, float, float]:
        """(k_x, k_eps, c_x0, c_xt, sigma_noise) of the fused kernel form; see the module docstring."""
        i = self._index(timestep)
        sigma, sigma_next = self._sigmas[i], self._sigmas[i + 1]
        return 1.0, float(sigma), float(1.0 - sigma_next / sigma), float(sigma_next / sigma), 0.0

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, s_churn: float = 0.0, s_tmin: float = 0.0,
             s_tmax: float = float("inf"), s_noise: float = 1.0, generator=None, return_dict: bool = True):
        """Generic (stock-PyTorch) form of one Euler step; the pipeline's hot loop runs the fused kernel instead."""
        if s_churn != 0.0:
            raise NotImplementedError("s_churn > 0 (stochastic Euler) is not supported")
        i = self._index(timestep)
        sigma, sigma_next = float(self._sigmas[i]), float(self._sigmas[i + 1])
        pred_original_sample = sample - sigma * model_output
        derivative = (sample - pred_original_sample) / sigma
        prev_sample = sample + derivative * (sigma_next - sigma)
        if not return_dict:
            return (prev_sample,)
        return EulerDiscreteSchedulerOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)
