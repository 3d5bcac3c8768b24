"""Generates tests/golden/ddpm_50step.pt: a full 50-step CFG + DDPM trajectory with the REFERENCE UNet
(full width, 909 M parameters, under tests/refshim) as the denoiser.  Build container only:
    python tests/golden/make_golden_traj.py

The loop below is written directly from /root/reference/base/pipelines/pipeline_videogen.py:662-689
(duplicate latents for CFG 666, scale_model_input 667, UNet 670-675, `u + s (c - u)` 679-680,
scheduler.step with the caller's generator 683) and does NOT go through oracle/: the fixture is a
reference-side trajectory, so `oracle.ddpm` and the HIP pipeline are both checked against it.
`scheduler.step` is diffusers 0.16.0's DDPMScheduler (absent from the reference tree and the image:
PARITY UNPINNED for that class): `_step` restates its published arithmetic with fp32 0-dim tensors
as that class does, independently of oracle/ddpm.py (python floats) and lavie_amd/scheduling_ddpm.py.
Noise is drawn the way diffusers' randn_tensor draws it for a CPU generator: torch.randn(shape,
generator=g, dtype=float32) once per step with t > 0."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import refbuild  # noqa: E402
from lavie_amd import spec, weights  # noqa: E402
from lavie_amd.config import UNetConfig  # noqa: E402

STEPS, TRAIN, GUIDANCE = 50, 1000, 7.5
WEIGHT_SEED, LATENT_SEED, NOISE_SEED = 0, 2000, 3000          # SURVEY.md §8d seeds for prompt 0
KEEP = (0, 1, 2, 4, 9, 19, 29, 39, 48, 49)                    # loop indices whose latents are kept


def q16(t):
    return t.to(torch.float16).to(torch.float32)


def _step(alphas_cumprod, eps, t, sample, generator):
    """One DDPMScheduler.step (epsilon prediction, fixed_small variance, clip_sample False), fp32 tensor arithmetic."""
    prev_t = t - TRAIN // STEPS
    one = torch.tensor(1.0)
    a_t = alphas_cumprod[t]
    a_prev = alphas_cumprod[prev_t] if prev_t >= 0 else one
    b_t, b_prev = 1 - a_t, 1 - a_prev
    cur_alpha = a_t / a_prev
    cur_beta = 1 - cur_alpha
    x0 = (sample - b_t ** 0.5 * eps) / a_t ** 0.5
    prev = (a_prev ** 0.5 * cur_beta) / b_t * x0 + cur_alpha ** 0.5 * b_prev / b_t * sample
    if t > 0:
        noise = torch.randn(eps.shape, generator=generator, dtype=eps.dtype)
        variance = torch.clamp(b_prev / b_t * cur_beta, min=1e-20)
        prev = prev + variance ** 0.5 * noise
    return prev


@torch.no_grad()
def main():
    cfg = UNetConfig()
    net, _ = refbuild.reference_unet(cfg, WEIGHT_SEED)
    net.load_state_dict({k: q16(v) for k, v in weights.synth_state_dict(spec.param_shapes(cfg), WEIGHT_SEED).items()})
    g = torch.Generator().manual_seed(4321)
    pe, ne = q16(torch.randn(1, 77, 768, generator=g)), q16(torch.randn(1, 77, 768, generator=g))
    latents = torch.randn(1, 4, 16, 8, 8, generator=torch.Generator().manual_seed(LATENT_SEED))
    gen = torch.Generator().manual_seed(NOISE_SEED)

    betas = torch.linspace(1e-4, 0.02, TRAIN, dtype=torch.float32)          # base/pipelines/sample.py:56-61
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    timesteps = (torch.arange(0, STEPS) * (TRAIN // STEPS)).flip(0)         # 980 ... 0
    prompt_embeds = torch.cat([ne, pe])                                     # :418 unconditional half first
    x = latents * 1.0                                                       # init_noise_sigma
    kept = {}
    t0 = time.time()
    for i, t in enumerate(timesteps):
        model_in = torch.cat([x] * 2)                                       # :666 (scale_model_input is the identity)
        noise_pred = net(model_in, t, encoder_hidden_states=prompt_embeds).sample
        uncond, text = noise_pred.chunk(2)
        noise_pred = uncond + GUIDANCE * (text - uncond)                    # :679-680
        x = _step(alphas_cumprod, noise_pred, int(t), x, gen)               # :683
        if i in KEEP:
            kept[i] = x.clone()
        print(f"step {i} t={int(t)} |x|={x.norm():.4f} ({time.time() - t0:.0f}s)", flush=True)
    path = os.path.join(HERE, "ddpm_50step.pt")
    torch.save(dict(weight_seed=WEIGHT_SEED, noise_seed=NOISE_SEED, guidance_scale=GUIDANCE, steps=STEPS, latents=latents,
                    prompt=pe.half(), negative=ne.half(), kept=kept, y=x), path)
    print(f"wrote ddpm_50step.pt: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
