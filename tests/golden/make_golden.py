"""Generates tests/golden/*.pt by running the REFERENCE modules (under tests/refshim) in the build
container.  Run from the repo root:  python tests/golden/make_golden.py

Every fixture is data only: seeded inputs (already rounded to fp16 so the GPU sees the same bits),
the reference's fp32 outputs, and — instead of weights — the (shapes, seed) recipe of
lavie_amd.weights.synth_state_dict; weights are rounded to fp16 before the reference runs.
What each fixture pins is listed in tests/golden/README.md.  The third-party arithmetic supplied by
the shim (GEGLU, Timesteps, rotary) is part of these outputs and is parity-unpinned
against the real packages."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import refbuild  # noqa: E402
import refimport  # noqa: E402
from lavie_amd import spec, weights  # noqa: E402
from lavie_amd.config import UNetConfig  # noqa: E402


def q16(t):
    return t.to(torch.float16).to(torch.float32)


def synth16(shapes, seed):
    return {k: q16(v) for k, v in weights.synth_state_dict(shapes, seed).items()}


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


def module_shapes(mod):
    return {k: tuple(v.shape) for k, v in mod.state_dict().items()}


@torch.no_grad()
def main():
    m = refimport.load()
    from rotary_embedding_torch import RotaryEmbedding
    g = torch.Generator().manual_seed(1234)

    # (1) ResnetBlock3D, 5-D GroupNorm domain, with and without shortcut; Upsample3D / Downsample3D   [T1]
    cases = []
    for cin, cout in ((64, 128), (128, 128)):
        blk = m.resnet.ResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=256, groups=32, eps=1e-5).eval()
        shapes = module_shapes(blk)
        seed = 100 + cin
        blk.load_state_dict(synth16(shapes, seed))
        x = q16(torch.randn(2, cin, 4, 8, 8, generator=g) * torch.linspace(0.5, 2.0, 4).reshape(1, 1, 4, 1, 1))
        temb = q16(torch.randn(2, 256, generator=g))
        cases.append(dict(cin=cin, cout=cout, shapes=shapes, seed=seed, x=x.half(), temb=temb.half(), y=blk(x, temb)))
    up = m.resnet.Upsample3D(64, use_conv=True, out_channels=64).eval()
    dn = m.resnet.Downsample3D(64, use_conv=True, out_channels=64, padding=1, name="op").eval()
    up_shapes, dn_shapes = module_shapes(up), module_shapes(dn)
    up.load_state_dict(synth16(up_shapes, 201))
    dn.load_state_dict(synth16(dn_shapes, 202))
    xs = q16(torch.randn(1, 64, 4, 8, 8, generator=g))
    save("resnet.pt", dict(cases=cases, sampler=dict(x=xs.half(), up_shapes=up_shapes, up_seed=201, up=up(xs),
                                                     dn_shapes=dn_shapes, dn_seed=202, dn=dn(xs))))

    # (2) TemporalAttention at the three head widths, non-zero to_out and bias table                  [T2]
    cases = []
    for c, frames in ((320, 16), (640, 16), (1280, 16), (320, 61)):
        att = m.attention.TemporalAttention(query_dim=c, heads=8, dim_head=c // 8, rotary_emb=RotaryEmbedding(32)).eval()
        shapes = module_shapes(att)
        seed = 300 + c + frames
        att.load_state_dict(synth16(shapes, seed))
        nseq = 12 if frames == 16 else 6          # not a multiple of the kernel's pixel tile: exercises tails
        x = q16(torch.randn(nseq, frames, c, generator=g))
        cases.append(dict(c=c, frames=frames, shapes=shapes, seed=seed, x=x.half(), y=att(x).half()))
    save("temporal_attention.pt", dict(cases=cases))

    # (3) CrossAttention: spatial self-attention and 77-token text cross-attention                    [T2]
    cases = []
    for c, d, ctx_dim in ((1280, 48, None), (640, 160, None), (320, 64, 768)):
        att = m.attention.CrossAttention(query_dim=c, cross_attention_dim=ctx_dim, heads=8, dim_head=c // 8).eval()
        shapes = module_shapes(att)
        seed = 400 + c
        att.load_state_dict(synth16(shapes, seed))
        x = q16(torch.randn(2, d, c, generator=g))
        ctx = None if ctx_dim is None else q16(torch.randn(2, 77, ctx_dim, generator=g))
        y = att(x, encoder_hidden_states=ctx)
        cases.append(dict(c=c, d=d, shapes=shapes, seed=seed, x=x.half(), ctx=None if ctx is None else ctx.half(), y=y.half()))
    save("cross_attention.pt", dict(cases=cases))

    # (4) Transformer3DModel (per-frame GN eps 1e-6, order spatial -> text -> temporal -> FF)          [T2]
    tr = m.attention.Transformer3DModel(8, 40, in_channels=320, num_layers=1, cross_attention_dim=768,
                                        norm_num_groups=32, rotary_emb=RotaryEmbedding(32)).eval()
    shapes = module_shapes(tr)
    tr.load_state_dict(synth16(shapes, 500))
    x = q16(torch.randn(2, 320, 16, 4, 4, generator=g) * torch.linspace(0.5, 2.0, 16).reshape(1, 1, 16, 1, 1))
    ctx = q16(torch.randn(2, 77, 768, generator=g))
    save("transformer3d.pt", dict(shapes=shapes, seed=500, x=x.half(), ctx=ctx.half(),
                                  y=tr(x, encoder_hidden_states=ctx, use_image_num=0).sample))

    # (6) relative-position bucket tables                                                             [T2]
    tables = {}
    for n in (16, 61):
        q = torch.arange(n)
        rel = q.reshape(1, n) - q.reshape(n, 1)
        tables[n] = m.attention.RelativePositionBias._relative_position_bucket(rel, num_buckets=32, max_distance=32)
    save("relpos_buckets.pt", tables)

    # (5) whole UNet, full width (909 M parameters), latent 8x8, F=16, B=2, three timesteps           [T2]
    cfg = UNetConfig()
    seed = 0
    net, _ = refbuild.reference_unet(cfg, seed)
    net.load_state_dict(synth16(spec.param_shapes(cfg), seed))
    x = q16(torch.randn(2, 4, 16, 8, 8, generator=g))
    ctx = q16(torch.randn(2, 77, 768, generator=g))
    outs = {t: net(x, torch.tensor(t), encoder_hidden_states=ctx).sample for t in (980, 500, 0)}
    save("unet_full_8x8.pt", dict(seed=seed, x=x.half(), ctx=ctx.half(), y=outs))

    # (7) three CFG + DDPM steps on [1,4,16,8,8] with the reference UNet as the denoiser               [T2 + T3]
    from oracle.ddpm import cfg_denoise_loop
    lat = torch.randn(1, 4, 16, 8, 8, generator=g)
    pe, ne = q16(torch.randn(1, 77, 768, generator=g)), q16(torch.randn(1, 77, 768, generator=g))
    noises = [torch.randn(1, 4, 16, 8, 8, generator=g) for _ in range(3)]
    fn = lambda xx, t, c: net(xx, torch.tensor(t), encoder_hidden_states=c).sample
    out = cfg_denoise_loop(fn, lat, pe, ne, noises, num_steps=50, guidance_scale=7.5, max_steps=3)
    save("ddpm_3step.pt", dict(seed=seed, latents=lat, prompt=pe.half(), negative=ne.half(), noises=noises, y=out))


if __name__ == "__main__":
    main()
