"""Generates tests/golden/interp_*.pt by running the REFERENCE's frame-interpolation modules
(/root/reference/interpolation/models, under tests/refshim) in the build container.
Run from the repo root:  python tests/golden/make_golden_interp.py

Same conventions as make_golden.py: data only — seeded inputs rounded to fp16, the reference's fp32 outputs, and the
(shapes, seed) recipe of lavie_amd.weights.synth_state_dict instead of weights.  The GEGLU feed-forward and the
timestep embedding come from the shim restatements (parity-unpinned against the real diffusers 0.16.0)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import refbuild  # noqa: E402
import refimport  # noqa: E402
from lavie_amd import spec  # noqa: E402
from lavie_amd.config import INTERPOLATION_CONFIG  # noqa: E402
from make_golden import module_shapes, q16, save, synth16  # noqa: E402


@torch.no_grad()
def main():
    m = refimport.load("interpolation")
    g = torch.Generator().manual_seed(4321)

    # (1) SparseCausalAttention (interpolation/models/attention.py:609-665) at the three head widths; token counts that
    #     are not multiples of the kernel's 64-key tile, so tiles straddle the first-frame / previous-frame boundary
    cases = []
    for c, d, frames in ((320, 40, 5), (640, 160, 3), (1280, 24, 4), (320, 256, 2)):
        att = m.attention.SparseCausalAttention(query_dim=c, heads=8, dim_head=c // 8).eval()
        shapes = module_shapes(att)
        seed = 600 + c + d
        att.load_state_dict(synth16(shapes, seed))
        x = q16(torch.randn(2 * frames, d, c, generator=g))
        cases.append(dict(c=c, d=d, frames=frames, shapes=shapes, seed=seed, x=x.half(), y=att(x, video_length=frames).half()))
    save("interp_sparse_causal.pt", dict(cases=cases))

    # (2) Transformer3DModel of the interpolation stage: sparse-causal attn1, order spatial -> text -> FF -> temporal,
    #     plain temporal attention (no rotary / relative-position bias); a short clip and the 61-frame clip
    cases = []
    for frames, h, w in ((7, 4, 4), (61, 2, 2)):
        tr = m.attention.Transformer3DModel(8, 40, in_channels=320, num_layers=1, cross_attention_dim=768,
                                            norm_num_groups=32, use_first_frame=True).eval()
        shapes = module_shapes(tr)
        seed = 700 + frames
        tr.load_state_dict(synth16(shapes, seed))
        x = q16(torch.randn(2, 320, frames, h, w, generator=g) * torch.linspace(0.5, 2.0, frames).reshape(1, 1, frames, 1, 1))
        ctx = q16(torch.randn(2, 77, 768, generator=g))
        cases.append(dict(frames=frames, shapes=shapes, seed=seed, x=x.half(), ctx=ctx.half(),
                          y=tr(x, encoder_hidden_states=ctx).sample))
    save("interp_transformer3d.pt", dict(cases=cases))

    # (3) whole interpolation UNet at full width (in_channels 8, use_first_frame), latent 8x8, F=5, B=2
    cfg = INTERPOLATION_CONFIG
    seed = 3
    net = refbuild.reference_interpolation_unet(cfg, seed)
    net.load_state_dict(synth16(spec.param_shapes(cfg), seed))
    x = q16(torch.randn(2, 8, 5, 8, 8, generator=g))
    ctx = q16(torch.randn(2, 77, 768, generator=g))
    outs = {t: net(x, torch.tensor(t), encoder_hidden_states=ctx).sample for t in (900, 20)}
    # forward_with_cfg (interpolation/models/unet.py:454-474): conditional half first, guidance 4.0
    cfg_out = net.forward_with_cfg(x, torch.tensor([500, 500]), encoder_hidden_states=ctx, cfg_scale=4.0)
    save("interp_unet_full_8x8.pt", dict(seed=seed, x=x.half(), ctx=ctx.half(), y=outs, y_cfg=cfg_out))

    # (4) the sampler: create_diffusion's respaced schedule tables and a whole 4-step ddim_sample_loop driven exactly as
    #     interpolation/sample.py:138-174 does (forward_with_cfg, x_start = copied low-frame-rate latent, use_concat,
    #     copy_no_mask, clip_denoised=False), with the reference UNet from (3) as the denoiser
    rd = refimport.load_interp_diffusion()
    tables = {}
    for n in ("50", "4"):
        d = rd.create_diffusion(n)
        tables[n] = dict(timestep_map=torch.tensor(d.timestep_map), alphas_cumprod=torch.from_numpy(d.alphas_cumprod),
                         alphas_cumprod_prev=torch.from_numpy(d.alphas_cumprod_prev))
    z = q16(torch.randn(1, 4, 5, 8, 8, generator=g))
    xs = q16(torch.randn(1, 4, 5, 8, 8, generator=g))
    ctx2 = q16(torch.randn(2, 77, 768, generator=g))          # [prompt, negative] (sample.py:157)
    z2, xs2 = torch.cat([z] * 2), torch.cat([xs] * 2)
    d = rd.create_diffusion("4")
    out = d.ddim_sample_loop(net.forward_with_cfg, z2.shape, z2, clip_denoised=False,
                             model_kwargs=dict(encoder_hidden_states=ctx2, class_labels=None), progress=False, device="cpu",
                             mask=None, x_start=xs2, use_concat=True, copy_no_mask=True)
    save("interp_ddim.pt", dict(seed=seed, tables=tables, z=z, x_start=xs, ctx=ctx2.half(), steps="4", y=out))


if __name__ == "__main__":
    main()
