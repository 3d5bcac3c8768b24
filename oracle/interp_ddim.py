"""CPU restatement of the interpolation stage's sampler (TEST INFRASTRUCTURE, see oracle/__init__.py): the OpenAI-style
respaced Gaussian diffusion the reference vendors in-tree, as `interpolation/sample.py:138-174` drives it —
`create_diffusion("50")` + `ddim_sample_loop(model.forward_with_cfg, ..., x_start=copied_video, use_concat=True,
copy_no_mask=True)`.

PINNED: the source is inside the reference (`interpolation/diffusion/{__init__,respace,gaussian_diffusion}.py`), is
imported in the build container by tests/test_oracle_vs_reference.py and frozen into tests/golden/interp_ddim.pt.

  space_timesteps        respace.py:15-68      (integer section counts)
  SpacedSchedule         respace.py:71-100, gaussian_diffusion.py:98-115, 153-201, __init__.py:10-46
  ddim_step              gaussian_diffusion.py:254-360 (epsilon model, fixed variance), 362-394, 587-642
  forward_with_cfg       interpolation/models/unet.py:454-474
  ddim_sample_loop       gaussian_diffusion.py:682-778
"""
from typing import Callable, List, Optional

import numpy as np
import torch


def space_timesteps(num_timesteps: int, section_counts) -> List[int]:
    """respace.py:15-68 for comma-separated / list section counts (the sampler passes str(num_sampling_steps))."""
    if isinstance(section_counts, str):
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return sorted(set(steps))


class SpacedSchedule:
    """SpacedDiffusion's constants in float64: the retained steps of the linear 1e-4 -> 0.02 schedule, their cumulative
    alphas (respace.py:82-96 re-derives betas so that the cumulative products match the base process at those steps)."""

    def __init__(self, timestep_respacing="50", diffusion_steps: int = 1000):
        scale = 1000 / diffusion_steps
        betas = np.linspace(scale * 0.0001, scale * 0.02, diffusion_steps, dtype=np.float64)
        base = np.cumprod(1.0 - betas, axis=0)
        self.timestep_map = space_timesteps(diffusion_steps, timestep_respacing)
        last, new_betas = 1.0, []
        for i in self.timestep_map:
            new_betas.append(1 - base[i] / last)
            last = base[i]
        self.betas = np.array(new_betas, dtype=np.float64)
        self.num_timesteps = len(self.betas)
        self.alphas_cumprod = np.cumprod(1.0 - self.betas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])


def forward_with_cfg(unet: Callable, x, t, ctx, cfg_scale: float):
    """interpolation/models/unet.py:454-474: first half run twice; CONDITIONAL text is the first half of ctx."""
    half = x[: len(x) // 2]
    out = unet(torch.cat([half, half], dim=0), t, ctx)
    eps, rest = out[:, :4], out[:, 4:]
    cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
    g = uncond + cfg_scale * (cond - uncond)
    return torch.cat([torch.cat([g, g], dim=0), rest], dim=1)


def ddim_step(sch: SpacedSchedule, eps, x, i: int, eta: float = 0.0, noise=None):
    """gaussian_diffusion.py:587-642 with clip_denoised=False: x0 = sqrt(1/abar) x - sqrt(1/abar - 1) eps (:386-389), eps
    re-derived from x0 (:623), x_prev = sqrt(abar_prev) x0 + sqrt(1 - abar_prev - sigma^2) eps + [i != 0] sigma noise."""
    ab, ab_prev = sch.alphas_cumprod[i], sch.alphas_cumprod_prev[i]
    k_x, k_e = float(np.sqrt(1.0 / ab)), float(np.sqrt(1.0 / ab - 1))
    x0 = k_x * x - k_e * eps
    eps = (k_x * x - x0) / k_e
    sigma = float(eta * np.sqrt((1 - ab_prev) / (1 - ab)) * np.sqrt(1 - ab / ab_prev))
    out = x0 * float(np.sqrt(ab_prev)) + float(np.sqrt(1 - ab_prev - sigma ** 2)) * eps
    if i != 0 and sigma != 0.0:
        out = out + sigma * noise
    return out


def ddim_sample_loop(unet: Callable, z, x_start, ctx, sch: SpacedSchedule, cfg_scale: float = 4.0, eta: float = 0.0,
                     noises: Optional[List[torch.Tensor]] = None, max_steps: Optional[int] = None):
    """gaussian_diffusion.py:723-778 + p_mean_variance's concat (:282-285) + _WrappedModel's timestep map
    (respace.py:117-128).  z, x_start: [2k, 4, f, h, w] (the k videos duplicated for guidance, sample.py:150-155);
    ctx [2k, n, c] = prompts then negatives; `unet(x8, t_long[2k], ctx) -> eps`.  `noises[j]` is used at loop position j
    when eta > 0."""
    x = z
    for j, i in enumerate(reversed(range(sch.num_timesteps))):
        if max_steps is not None and j >= max_steps:
            break
        t = torch.full((x.shape[0],), sch.timestep_map[i], dtype=torch.long)
        eps = forward_with_cfg(unet, torch.cat([x, x_start], dim=1), t, ctx, cfg_scale)
        x = ddim_step(sch, eps, x, i, eta, None if noises is None else noises[j])
    return x
