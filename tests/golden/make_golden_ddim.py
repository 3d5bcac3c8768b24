"""Generates tests/golden/ddim_steps.pt by running the REFERENCE's vendored DDIM scheduler class
(/root/reference/vsr/diffusion/scheduling_ddim.py, imported under tests/refshim) in the build container.
Run from the repo root:  python tests/golden/make_golden_ddim.py

Data only: seeded inputs and the reference class's outputs.  The scheduler is configured as the base pipeline's
`sample_method == 'ddim'` branch sees it (base/pipelines/sample.py:44-49 with base/configs/sample.yaml: beta linear
1e-4 .. 0.02; SD-1.4 scheduler_config.json: set_alpha_to_one=false, steps_offset=1, clip_sample=false)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import refimport  # noqa: E402


@torch.no_grad()
def main():
    DDIM = refimport.load_vsr_ddim()
    sch = DDIM(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, beta_schedule="linear", clip_sample=False,
               set_alpha_to_one=False, steps_offset=1)
    sch.set_timesteps(50)                                   # the vendored (VSR) spacing: linspace + offset
    vsr_timesteps = sch.timesteps.clone()
    g = torch.Generator().manual_seed(4321)
    cases = []
    for t in (981, 961, 501, 21, 1):                        # 1 -> previous timestep < 0: final_alpha_cumprod branch
        for eta in (0.0, 0.5):
            x = torch.randn(1, 4, 4, 8, 8, generator=g)
            eps = torch.randn(1, 4, 4, 8, 8, generator=g)
            z = torch.randn(1, 4, 4, 8, 8, generator=g)
            out = sch.step(eps, t, x, eta=eta, variance_noise=z if eta > 0 else None)
            cases.append(dict(t=t, eta=eta, x=x, eps=eps, noise=z, prev=out.prev_sample, x0=out.pred_original_sample))
    # a short deterministic chain (eta = 0) over the first five stock-spaced timesteps with a fixed "model"
    x = torch.randn(1, 4, 4, 8, 8, generator=g)
    chain_in = x.clone()
    ts = [981, 961, 941, 921, 901]
    for t in ts:
        eps = torch.tanh(x * 0.7 + 0.01 * t / 1000.0)       # any deterministic function of (x, t)
        x = sch.step(eps, t, x, eta=0.0).prev_sample
    # the other two prediction types of the vendored step (:356-363); v_prediction is what the x4-upscaler's own
    # scheduler_config.json is published with (the file is not in the reference tree)
    pcases = []
    for kind in ("v_prediction", "sample"):
        psch = DDIM(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, beta_schedule="linear", clip_sample=False,
                    set_alpha_to_one=False, steps_offset=1, prediction_type=kind)
        psch.set_timesteps(50)
        for t in (981, 501, 1):
            for eta in (0.0, 0.5):
                xx = torch.randn(1, 4, 4, 8, 8, generator=g)
                m = torch.randn(1, 4, 4, 8, 8, generator=g)
                z = torch.randn(1, 4, 4, 8, 8, generator=g)
                out = psch.step(m, t, xx, eta=eta, variance_noise=z if eta > 0 else None)
                pcases.append(dict(kind=kind, t=t, eta=eta, x=xx, model_output=m, noise=z, prev=out.prev_sample,
                                   x0=out.pred_original_sample))
    path = os.path.join(HERE, "ddim_steps.pt")
    torch.save(dict(vsr_timesteps_50=vsr_timesteps, alphas_cumprod=sch.alphas_cumprod.clone(), cases=cases,
                    chain=dict(x=chain_in, timesteps=ts, y=x), prediction_cases=pcases), path)
    print(f"wrote ddim_steps.pt: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
