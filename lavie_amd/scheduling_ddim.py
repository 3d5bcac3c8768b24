"""DDIM scheduler with the surface `VideoGenPipeline` uses (pipeline_videogen.py:112-137, 431-446, 509, 641-642,
662, 667, 683) — the `sample_method == 'ddim'` branch of the reference (base/pipelines/sample.py:44-49).

The reference takes `DDIMScheduler` from diffusers 0.16.0 (not in its tree), but it also VENDORS that class's text
in `vsr/diffusion/scheduling_ddim.py`; this file follows the vendored text: constructor :137-182, `_get_variance`
:198-207, `set_timesteps` :243-265 (the stock "leading" spacing, kept there as a comment), `step` :286-405 with its
three prediction types (:354-363).  No sample clipping / thresholding (SD-1.4's scheduler_config.json: clip_sample=false,
set_alpha_to_one=false, steps_offset=1).

`timestep_spacing="vsr_linspace"` reproduces the vendored file's live `set_timesteps` (:267-285: linspace + offset).
Nothing in the reference runs that text: vsr/sample.py:19,53 installs the STOCK `diffusers.DDIMScheduler.from_config(
x4-upscaler scheduler_config.json)` with only beta_schedule overridden, and the vendored variant's first timestep
(1001 at 50 steps, 1000 with steps_offset 0) lies past its own 1000-entry alpha table.  The mode is kept so the
timestep list can be compared with the vendored class; `coefficients` / `step` raise for a timestep outside the table.

`coefficients(t, eta)` exposes the five scalars of one step in the form of the fused HIP kernel
(`lavie_cfg_ddpm_step`: x0 = k_x x - k_m m;  x_prev = c_x0 x0 + c_xt x + sigma z, m = the guided model output);
derivation in its docstring."""
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
        if clip_sample or thresholding:
            raise NotImplementedError("sample clipping / thresholding are not supported")
        if prediction_type not in ("epsilon", "sample", "v_prediction"):
            raise ValueError(f"prediction_type given as {prediction_type} must be one of `epsilon`, `sample`, or `v_prediction`")
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


    _CONFIG_KEYS = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "clip_sample", "set_alpha_to_one",
                    "steps_offset", "prediction_type", "thresholding", "timestep_spacing")

    @classmethod
    def from_config(cls, config, **overrides):
        """Builds the scheduler from the fields of a diffusers `scheduler_config.json` (a dict or a path to the file), as
        `DDIMScheduler.from_config(pipeline.scheduler.config, beta_schedule=...)` does at vsr/sample.py:49-53 and
        base/pipelines/sample.py:45-49.  Keys this class does not model (`_class_name`, `trained_betas`, ...) are ignored;
        `overrides` win over the file.  The VSR stage's scheduler comes from the x4-upscaler checkpoint's
        scheduler/scheduler_config.json, which is NOT in the reference tree: its `prediction_type`, betas and
        `steps_offset` must be taken from the user's file — the constructor defaults (epsilon, SD-1.4 betas) are not a
        statement about what the reference's VSR stage runs (parity unpinned for that configuration)."""
        if isinstance(config, (str, bytes)) or hasattr(config, "__fspath__"):
            import json
            with open(config) as fh:
                config = json.load(fh)
        cfg = {k: config[k] for k in cls._CONFIG_KEYS if k in config}
        cfg.update(overrides)
        unknown = set(overrides) - set(cls._CONFIG_KEYS)
        if unknown:
            raise TypeError(f"from_config: unknown scheduler fields {sorted(unknown)}")
        # Keys outside _CONFIG_KEYS are dropped only while they are inert.  One that would change the schedule or the step if a
        # diffusers scheduler read it (and this class does not model it) is an error, not a silently different sampler.
        merged = dict(config)
        merged.update(overrides)
        for key, inert in cls._UNMODELLED_DEFAULTS.items():
            if key in merged and merged[key] not in inert:
                raise NotImplementedError(f"from_config: {key}={merged[key]!r} changes the schedule / step and is not modelled here "
                                          f"(accepted: {list(inert)})")
        return cls(**cfg)

    # key -> values under which it does nothing (diffusers' own defaults); anything else raises in from_config
    _UNMODELLED_DEFAULTS = {"trained_betas": (None,), "rescale_betas_zero_snr": (False,), "dynamic_thresholding_ratio": (0.995,),
                            "clip_sample_range": (1.0,), "sample_max_value": (1.0,)}

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
        if not 0 <= t < self.config.num_train_timesteps:
            raise ValueError(f"timestep {t} is outside the {self.config.num_train_timesteps}-entry alpha table"
                             " (timestep_spacing='vsr_linspace' starts past it: see the module docstring)")
        prev = t - self.config.num_train_timesteps // self.num_inference_steps             # :343
        a_t = float(self.alphas_cumprod[t])                                                # :346
        a_prev = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)   # :347
        return a_t, a_prev

    def _variance(self, a_t: float, a_prev: float) -> float:                              # :198-207
        return ((1.0 - a_prev) / (1.0 - a_t)) * (1.0 - a_t / a_prev)

    def coefficients(self, timestep: int, eta: float = 0.0) -> Tuple[float, float, float, float, float]:
        """(k_x, k_m, c_x0, c_xt, sigma) of the fused kernel form (m = the model output after guidance):
        x0 = k_x x - k_m m;  x_prev = c_x0 x0 + c_xt x + sigma z.  With a = sqrt(abar_prev), b = sqrt(1 - abar_prev -
        sigma^2) DDIM's update is x_prev = a x0 + b eps + sigma z (:389-392), and eps is linear in (x, x0) for every
        prediction type (:354-363):
          epsilon       x0 = (x - sqrt(1-abar) m) / sqrt(abar)      eps = (x - sqrt(abar) x0) / sqrt(1-abar)
          v_prediction  x0 = sqrt(abar) x - sqrt(1-abar) m          eps = sqrt(abar) m + sqrt(1-abar) x   (same eps(x, x0))
          sample        x0 = m                                      eps = (x - sqrt(abar) x0) / sqrt(1-abar)
        so c_x0 = a - b sqrt(abar) / sqrt(1-abar) and c_xt = b / sqrt(1-abar) in all three."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        a_t, a_prev = self._alphas(int(timestep))
        sigma = eta * max(self._variance(a_t, a_prev), 0.0) ** 0.5                         # :381-382
        sa, sb = a_t ** 0.5, (1.0 - a_t) ** 0.5
        kind = self.config.prediction_type
        if kind == "epsilon":
            k_x, k_m = 1.0 / sa, sb / sa                                                   # :354
        elif kind == "v_prediction":
            k_x, k_m = sa, sb                                                              # :360
        else:
            k_x, k_m = 0.0, -1.0                                                           # :357  x0 = m
        a = a_prev ** 0.5
        b = max(1.0 - a_prev - sigma * sigma, 0.0) ** 0.5                                  # :389
        return k_x, k_m, a - b * sa / sb, b / sb, sigma

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0,
             use_clipped_model_output: bool = False, generator=None, variance_noise: Optional[torch.Tensor] = None,
             return_dict: bool = True):
        """Generic (stock-PyTorch) form of one DDIM step, statement by statement as :286-405; the pipeline's hot loop
        runs the fused kernel with `coefficients` instead."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        a_t, a_prev = self._alphas(int(timestep))
        beta_prod_t = 1.0 - a_t
        if self.config.prediction_type == "epsilon":
            pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / a_t ** 0.5     # :354
            pred_epsilon = model_output
        elif self.config.prediction_type == "sample":                                         # :356-358
            pred_original_sample = model_output
            pred_epsilon = (sample - a_t ** 0.5 * pred_original_sample) / beta_prod_t ** 0.5
        else:                                                                                 # v_prediction, :359-361
            pred_original_sample = a_t ** 0.5 * sample - beta_prod_t ** 0.5 * model_output
            pred_epsilon = a_t ** 0.5 * model_output + beta_prod_t ** 0.5 * sample
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
