"""DDIM scheduler with the surface `VideoGenPipeline` uses (pipeline_videogen.py:112-137, 431-446, 509, 641-642,
662, 667, 683) — the `sample_method == 'ddim'` branch of the reference (base/pipelines/sample.py:44-49).

The reference takes `DDIMScheduler` from diffusers 0.16.0 (not in its tree), but it also VENDORS that class's text
in `vsr/diffusion/scheduling_ddim.py`; this file follows the vendored text: constructor :137-182, `_get_variance`
:198-207, `set_timesteps` :243-265 (the stock "leading" spacing, kept there as a comment) and :267-285 (the
linspace variant the VSR stage runs, `timestep_spacing="vsr_linspace"`), `step` :286-405.  Epsilon prediction, no
sample clipping / thresholding (SD-1.4's scheduler_config.json: clip_sample=false, set_alpha_to_one=false,
steps_offset=1).

`coefficients(t, eta)` exposes the five scalars of one step in the form of the fused HIP kernel
(`lavie_cfg_ddpm_step`: x0 = k_x x - k_eps eps;  x_prev = c_x0 x0 + c_xt x + sigma z): with a = sqrt(abar_prev) and
b = sqrt(1 - abar_prev - sigma^2), DDIM's  x_prev = a x0 + b eps + sigma z  equals that form for
c_x0 = a - b / k_eps and c_xt = b k_x / k_eps  (eps = (k_x x - x0) / k_eps)."""
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import numpy as np
import torch

from .scheduling_ddpm import randn_tensor


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02,
                 beta_schedule: str = "linear", clip_sample: bool = False, set_alpha_to_one: bool = False,
                 steps_offset: int = 1, prediction_type: str = "epsilon", thresholding: bool = False,
                 timestep_spacing: str = "leading"):
        if clip_sample or thresholding or prediction_type != "epsilon":
            raise NotImplementedError("only epsilon prediction without sample clipping / thresholding is supported")
        if timestep_spacing not in ("leading", "vsr_linspace"):
            raise NotImplementedError(f"timestep_spacing={timestep_spacing!r}")
        if beta_schedule == "linear":                                                   # scheduling_ddim.py:155-156
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":                                          # :157-161
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for DDIMScheduler")
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)                          # :168-169
        # :173-175 — the "previous" alpha of the final step: 1, or the alpha of step 0
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0                                                      # :178
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                                      steps_offset=steps_offset, prediction_type=prediction_type, thresholding=thresholding,
                                      timestep_spacing=timestep_spacing)

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:    # :184-196
        return sample

    def set_timesteps(self, num_inference_steps: int, device: Union[str, torch.device, None] = None):
        n_train = self.config.num_train_timesteps
        if num_inference_steps > n_train:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.config.train_timesteps`:"
                             f" {n_train} as the unet model trained with this scheduler can only handle maximal {n_train} timesteps.")
        self.num_inference_steps = num_inference_steps
        if self.config.timestep_spacing == "leading":                                    # :259-265 (stock diffusers 0.16.0)
            step_ratio = n_train // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        else:                                                                            # :281-284 (VSR variant)
            ts = np.linspace(self.config.steps_offset, n_train, num_inference_steps).round()[::-1].copy().astype(np.int64)
        steps = torch.from_numpy(ts) + self.config.steps_offset
        self.timesteps = steps.to(device) if device is not None else steps

    def _alphas(self, t: int) -> Tuple[float, float]:
        prev = t - self.config.num_train_timesteps // self.num_inference_steps             # :343
        a_t = float(self.alphas_cumprod[t])                                                # :346
        a_prev = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)   # :347
        return a_t, a_prev

    def _variance(self, a_t: float, a_prev: float) -> float:                              # :198-207
        return ((1.0 - a_prev) / (1.0 - a_t)) * (1.0 - a_t / a_prev)

    def coefficients(self, timestep: int, eta: float = 0.0) -> Tuple[float, float, float, float, float]:
        """(k_x, k_eps, c_x0, c_xt, sigma) of the fused kernel form; see the module docstring."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        a_t, a_prev = self._alphas(int(timestep))
        sigma = eta * max(self._variance(a_t, a_prev), 0.0) ** 0.5                         # :381-382
        k_x, k_e = 1.0 / a_t ** 0.5, ((1.0 - a_t) ** 0.5) / a_t ** 0.5                     # :354
        a = a_prev ** 0.5
        b = max(1.0 - a_prev - sigma * sigma, 0.0) ** 0.5                                  # :389
        return k_x, k_e, a - b / k_e, b * k_x / k_e, sigma

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0,
             use_clipped_model_output: bool = False, generator=None, variance_noise: Optional[torch.Tensor] = None,
             return_dict: bool = True):
        """Generic (stock-PyTorch) form of one DDIM step, statement by statement as :286-405; the pipeline's hot loop
        runs the fused kernel with `coefficients` instead."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        a_t, a_prev = self._alphas(int(timestep))
        beta_prod_t = 1.0 - a_t
        pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / a_t ** 0.5     # :354
        pred_epsilon = model_output
        std_dev_t = eta * self._variance(a_t, a_prev) ** 0.5                                  # :381-382
        pred_sample_direction = (1.0 - a_prev - std_dev_t ** 2) ** 0.5 * pred_epsilon         # :389
        prev_sample = a_prev ** 0.5 * pred_original_sample + pred_sample_direction            # :392
        if eta > 0:                                                                           # :394-407
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` or"
                                 " `variance_noise` stays `None`.")
            if variance_noise is None:
                variance_noise = randn_tensor(model_output.shape, generator=generator, device=model_output.device,
                                              dtype=model_output.dtype)
            prev_sample = prev_sample + std_dev_t * variance_noise
        if not return_dict:
            return (prev_sample,)
        return DDIMSchedulerOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)
