"""fp32 CPU restatement of the DDIM sampler step (TEST INFRASTRUCTURE; `sample_method == 'ddim'`,
/root/reference/base/pipelines/sample.py:44-49).

The reference imports `DDIMScheduler` from diffusers 0.16.0 (absent here) but vendors the same class text in
/root/reference/vsr/diffusion/scheduling_ddim.py; each function cites the lines it follows.  PINNED: in the build
container `tests/test_oracle_vs_reference.py::test_ddim_*` runs this restatement against that vendored class (imported
under tests/refshim, which supplies plumbing symbols only), and `tests/golden/ddim_steps.pt` freezes the vendored
class's outputs for the GPU box."""
from typing import List, Optional

import numpy as np
import torch


class DDIMSchedule:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02,
                 beta_schedule: str = "linear", set_alpha_to_one: bool = False, steps_offset: int = 1,
                 timestep_spacing: str = "leading", prediction_type: str = "epsilon"):
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        if beta_schedule == "linear":                                   # scheduling_ddim.py:155-156
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:                                                           # "scaled_linear", :157-161
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)    # :168-169
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else self.alphas_cumprod[0].item()   # :175
        self.steps_offset = steps_offset
        self.timestep_spacing = timestep_spacing
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps: List[int] = []

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        if self.timestep_spacing == "leading":                          # :259-265 (stock text, kept as a comment there)
            ratio = self.num_train_timesteps // n
            ts = (np.arange(0, n) * ratio).round()[::-1].astype(np.int64)
        else:                                                           # :281-284 (the variant the VSR stage runs)
            ts = np.linspace(self.steps_offset, self.num_train_timesteps, n).round()[::-1].astype(np.int64)
        self.timesteps = [int(t) + self.steps_offset for t in ts]

    def step(self, eps: torch.Tensor, t: int, x: torch.Tensor, eta: float = 0.0, noise: Optional[torch.Tensor] = None):
        prev = t - self.num_train_timesteps // self.num_inference_steps                       # :343
        a_t = self.alphas_cumprod[t].item()                                                   # :346
        a_prev = self.alphas_cumprod[prev].item() if prev >= 0 else self.final_alpha_cumprod  # :347
        if self.prediction_type == "epsilon":
            x0 = (x - (1.0 - a_t) ** 0.5 * eps) / a_t ** 0.5                                  # :354
        elif self.prediction_type == "sample":                                                # :356-358 (`eps` = model output)
            x0 = eps
            eps = (x - a_t ** 0.5 * x0) / (1.0 - a_t) ** 0.5
        else:                                                                                 # "v_prediction", :359-361
            v = eps
            x0 = a_t ** 0.5 * x - (1.0 - a_t) ** 0.5 * v
            eps = a_t ** 0.5 * v + (1.0 - a_t) ** 0.5 * x
        var = ((1.0 - a_prev) / (1.0 - a_t)) * (1.0 - a_t / a_prev)                           # :198-207
        std = eta * var ** 0.5                                                                # :382
        out = a_prev ** 0.5 * x0 + (1.0 - a_prev - std ** 2) ** 0.5 * eps                     # :389-392
        if eta > 0:
            out = out + std * noise                                                           # :394-407
        return out
