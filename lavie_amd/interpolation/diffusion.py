"""Host-side mirror of `interpolation/diffusion/` (reference): `create_diffusion` and the respaced DDIM sample loop the
interpolation stage runs (`interpolation/sample.py:138-174`: `create_diffusion("50")`, `ddim_sample_loop(
model.forward_with_cfg, z.shape, z, clip_denoised=False, model_kwargs, mask=None, x_start=copied_video, use_concat=True,
copy_no_mask=True)`).

Schedule constants are float64 numpy, statement by statement as respace.py:15-100 and gaussian_diffusion.py:98-115,
153-201.  When `model` is the HIP UNet's `forward_with_cfg`, the loop runs the engine with the fused guidance + update
kernel (`lavie_cfg_ddpm_step`): with k_x = sqrt(1/abar), k_e = sqrt(1/abar - 1), a = sqrt(abar_prev) and
b = sqrt(1 - abar_prev - sigma^2), DDIM's  x_prev = a x0 + b eps + sigma z  is the kernel's
x_prev = c_x0 x0 + c_xt x + sigma z  for c_x0 = a - b / k_e, c_xt = b k_x / k_e.  Any other callable runs the same
arithmetic with stock PyTorch ops on whatever device its tensors live (scheduler arithmetic only: the UNet itself has no
such path)."""
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from .. import ops


def space_timesteps(num_timesteps: int, section_counts) -> set:
    """respace.py:15-68."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start_idx = 0
    all_steps: List[int] = []
    for i, section_count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < section_count:
            raise ValueError(f"cannot divide section of {size} steps into {section_count}")
        frac_stride = 1 if section_count <= 1 else (size - 1) / (section_count - 1)
        cur_idx = 0.0
        for _ in range(section_count):
            all_steps.append(start_idx + round(cur_idx))
            cur_idx += frac_stride
        start_idx += size
    return set(all_steps)


def get_named_beta_schedule(schedule_name: str, num_diffusion_timesteps: int) -> np.ndarray:
    """gaussian_diffusion.py:98-122 ("linear" only: the schedule create_diffusion uses)."""
    if schedule_name != "linear":
        raise NotImplementedError(f"unknown beta schedule: {schedule_name}")
    scale = 1000 / num_diffusion_timesteps
    return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)


class SpacedDiffusion:
    """respace.py:71-100 over gaussian_diffusion.py:153-201: epsilon prediction, fixed variance, retained timesteps."""

    def __init__(self, use_timesteps, betas):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(betas)
        base_cumprod = np.cumprod(1.0 - np.array(betas, dtype=np.float64), axis=0)
        self.timestep_map: List[int] = []
        last, new_betas = 1.0, []
        for i, alpha_cumprod in enumerate(base_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - alpha_cumprod / last)
                last = alpha_cumprod
                self.timestep_map.append(i)
        self.betas = np.array(new_betas, dtype=np.float64)
        assert (self.betas > 0).all() and (self.betas <= 1).all()
        self.num_timesteps = int(self.betas.shape[0])
        self.alphas_cumprod = np.cumprod(1.0 - self.betas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)

    # ------------------------------------------------------------------ one DDIM step (gaussian_diffusion.py:587-642)
    def ddim_coefficients(self, i: int, eta: float = 0.0) -> Tuple[float, float, float, float, float]:
        """(k_x, k_eps, c_x0, c_xt, sigma) of step index i in the fused kernel's form (module docstring)."""
        ab, ab_prev = self.alphas_cumprod[i], self.alphas_cumprod_prev[i]
        sigma = eta * np.sqrt((1 - ab_prev) / (1 - ab)) * np.sqrt(1 - ab / ab_prev)                 # :627-631
        k_x, k_e = self.sqrt_recip_alphas_cumprod[i], self.sqrt_recipm1_alphas_cumprod[i]           # :386-389
        a, b = np.sqrt(ab_prev), np.sqrt(1 - ab_prev - sigma ** 2)                                  # :634-637
        if i == 0:
            sigma = 0.0                                                                             # :638-641 nonzero_mask
        return float(k_x), float(k_e), float(a - b / k_e), float(b * k_x / k_e), float(sigma)

    def ddim_sample(self, model: Callable, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                    eta=0.0, mask=None, x_start=None, use_concat=False, copy_no_mask=False):
        """Generic (stock-PyTorch) form of one step; `t` holds step INDICES, the model sees timestep_map[t]
        (respace.py:117-128)."""
        self._check(clip_denoised, denoised_fn, cond_fn, mask, x_start, use_concat, copy_no_mask)
        i = int(t[0])
        mapped = torch.tensor(self.timestep_map, device=t.device, dtype=t.dtype)[t]
        inp = torch.cat([x, x_start], dim=1) if use_concat else x                                   # :282-296
        out = model(inp, mapped, **(model_kwargs or {}))
        eps = out.sample if hasattr(out, "sample") else out
        k_x, k_e, _, _, sigma = self.ddim_coefficients(i, eta)
        ab_prev = self.alphas_cumprod_prev[i]
        pred_xstart = k_x * x - k_e * eps
        eps = (k_x * x - pred_xstart) / k_e                                                          # :623
        noise = torch.randn_like(x)                                                                  # :633 (always drawn)
        sample = pred_xstart * float(np.sqrt(ab_prev)) + float(np.sqrt(1 - ab_prev - sigma ** 2)) * eps
        if i != 0:
            sample = sample + sigma * noise
        return {"sample": sample, "pred_xstart": pred_xstart}

    @staticmethod
    def _check(clip_denoised, denoised_fn, cond_fn, mask, x_start, use_concat, copy_no_mask):
        if clip_denoised or denoised_fn is not None or cond_fn is not None or mask is not None:
            raise NotImplementedError("clip_denoised / denoised_fn / cond_fn / mask are outside the MI355X path "
                                      "(interpolation/sample.py passes clip_denoised=False, mask=None)")
        if use_concat and (x_start is None or not copy_no_mask):
            raise NotImplementedError("use_concat needs x_start and copy_no_mask=True (the 8-channel model)")

    # ------------------------------------------------------------------ the loop (gaussian_diffusion.py:682-778)
    @torch.no_grad()
    def ddim_sample_loop(self, model: Callable, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, mask=None, x_start=None,
                         use_concat=False, copy_no_mask=False):
        self._check(clip_denoised, denoised_fn, cond_fn, mask, x_start, use_concat, copy_no_mask)
        unet = getattr(model, "__self__", None)
        from .unet import UNet3DConditionModel
        if isinstance(unet, UNet3DConditionModel) and getattr(model, "__name__", "") == "forward_with_cfg":
            return self._ddim_loop_hip(unet, shape, noise, model_kwargs or {}, eta, x_start, use_concat)
        img = noise if noise is not None else torch.randn(*shape, device=device)
        for i in list(range(self.num_timesteps))[::-1]:
            t = torch.tensor([i] * shape[0], device=img.device)
            img = self.ddim_sample(model, img, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, eta, mask, x_start,
                                   use_concat, copy_no_mask)["sample"]
        return img

    def _ddim_loop_hip(self, unet, shape, noise, model_kwargs, eta, x_start, use_concat, max_steps: Optional[int] = None):
        """forward_with_cfg only ever reads the first half of x (interpolation/models/unet.py:462-463) and the caller
        keeps the first half of the result (sample.py:171), so the loop carries the k = B/2 videos once, in fp32, and
        returns them duplicated.  The engine batch is ordered [unconditional | conditional] (the fused kernel's eps2
        order); both halves see identical latents, so this is the reference's cat([half, half]) with the text rows
        swapped to match."""
        ctx = model_kwargs.get("encoder_hidden_states")
        if model_kwargs.get("class_labels") is not None or ctx is None:
            raise NotImplementedError("model_kwargs must hold encoder_hidden_states and no class_labels")
        cfg_scale = float(model_kwargs.get("cfg_scale", 4.0))
        dev = unet.device
        if shape[0] % 2 != 0:
            raise ValueError("forward_with_cfg needs an even batch: videos duplicated for guidance")
        k = shape[0] // 2
        z = noise if noise is not None else torch.randn(*shape, device=dev)
        x = z[:k].to(device=dev, dtype=torch.float32).contiguous().clone()
        ctx = ctx.to(device=dev, dtype=torch.float16)
        ctx = torch.cat([ctx[k:], ctx[:k]], dim=0).contiguous()             # [negative | prompt]
        model_in = torch.empty((2 * k,) + tuple(x.shape[1:]), dtype=torch.float16, device=dev)
        ops.latents_to_model_input(x, model_in)
        if use_concat:
            xs = x_start[:k].to(device=dev, dtype=torch.float16)
            inp = torch.empty((2 * k, x.shape[1] + xs.shape[1]) + tuple(x.shape[2:]), dtype=torch.float16, device=dev)
            inp[:k, x.shape[1]:] = xs
            inp[k:, x.shape[1]:] = xs
        else:
            inp = model_in
        unet.prepare(2 * k, x.shape[2], x.shape[3], x.shape[4], ctx.shape[1])
        ctx = unet.cache_context(ctx)            # text keys / values once per clip, not once per block and step
        noise_dev = torch.empty_like(x)
        steps = list(range(self.num_timesteps))[::-1]
        t_dev = torch.tensor([float(self.timestep_map[i]) for i in steps], dtype=torch.float32, device=dev)
        try:                                     # an exception in the loop must not leave the engine holding ctx
            for j, i in enumerate(steps):
                if max_steps is not None and j >= max_steps:
                    break
                if use_concat:
                    inp[:, : x.shape[1]].copy_(model_in)                          # th.concat([x, x_start], dim=1), :285
                eps = unet(inp, t_dev[j], encoder_hidden_states=ctx).sample
                coeffs = self.ddim_coefficients(i, eta)
                step_noise = noise_dev.normal_() if coeffs[4] != 0.0 else None
                ops.cfg_ddpm_step(eps, x, step_noise, model_in, cfg_scale, coeffs)
        finally:
            unet.cache_context(None)
        return torch.cat([x, x], dim=0)


def create_diffusion(timestep_respacing, noise_schedule="linear", use_kl=False, sigma_small=False, predict_xstart=False,
                     learn_sigma=False, rescale_learned_sigmas=False, diffusion_steps=1000) -> SpacedDiffusion:
    """interpolation/diffusion/__init__.py:10-46.  Only the configuration the UNet stage samples with is supported:
    epsilon prediction, fixed (not learned) variance; the loss-type switches do not exist at inference."""
    if predict_xstart or learn_sigma:
        raise NotImplementedError("predict_xstart / learn_sigma are outside the MI355X path")
    betas = get_named_beta_schedule(noise_schedule, diffusion_steps)
    if timestep_respacing is None or timestep_respacing == "":
        timestep_respacing = [diffusion_steps]
    return SpacedDiffusion(use_timesteps=space_timesteps(diffusion_steps, timestep_respacing), betas=betas)
