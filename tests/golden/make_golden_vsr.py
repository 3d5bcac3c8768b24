"""Generates tests/golden/vsr_*.pt by running the REFERENCE's VSR-stage modules in the build container.
Run from the repo root:  python tests/golden/make_golden_vsr.py

`vsr/models/resnet.py` imports only torch / einops, so `ResnetBlock3DCNN` is the reference's own class, imported
directly (tier T1).  Same conventions as make_golden.py (data only; weights as a (shapes, seed) recipe)."""
import importlib.util
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

from make_golden import module_shapes, q16, save, synth16  # noqa: E402


def load_vsr_resnet():
    spec = importlib.util.spec_from_file_location("ref_vsr_resnet", "/root/reference/vsr/models/resnet.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@torch.no_grad()
def main():
    m = load_vsr_resnet()
    g = torch.Generator().manual_seed(777)
    # ResnetBlock3DCNN as TemporalModule3D builds it (vsr/models/temporal_module.py:112: kernel (5,1,1), in == out) at two
    # widths of the VSR UNet, a clip shorter than the kernel reach (F = 3), the 8-frame chunk and an odd clip; plus the
    # default (3,1,1) kernel.  temb_channels = 4 * 256 (vsr unet block_out_channels[0] = 256).
    cases = []
    for c, kern, frames, h, w in ((256, (5, 1, 1), 8, 6, 8), (512, (5, 1, 1), 3, 4, 4), (256, (3, 1, 1), 5, 5, 7),
                                  (1024, (5, 1, 1), 8, 2, 4)):
        blk = m.ResnetBlock3DCNN(in_channels=c, out_channels=c, kernel=kern, temb_channels=1024).eval()
        shapes = module_shapes(blk)
        seed = 900 + c + frames
        blk.load_state_dict(synth16(shapes, seed))
        x = q16(torch.randn(2, c, frames, h, w, generator=g) * torch.linspace(0.5, 2.0, frames).reshape(1, 1, frames, 1, 1))
        temb = q16(torch.randn(2, 1024, generator=g))
        cases.append(dict(c=c, taps=kern[0], shapes=shapes, seed=seed, x=x.half(), temb=temb.half(), y=blk(x, temb)))
    save("vsr_resnet3dcnn.pt", dict(cases=cases))

    # VSR Transformer3DModel (vsr/models/attention.py:314-594, under tests/refshim): resblock_temporal in front, attn1 as text
    # cross-attention (only_cross_attention levels) or spatial self-attention, Linear projections, temporal attention with
    # rotary + relative-position bias; widths / head dims of the VSR UNet (512 -> 64, 1024 -> 128), context width 1024
    import refimport
    mv = refimport.load_vsr_blocks()                    # puts tests/refshim on sys.path
    from rotary_embedding_torch import RotaryEmbedding
    cases = []
    for c, only_cross, frames, h, w in ((512, True, 8, 4, 4), (1024, False, 5, 2, 4)):
        tr = mv.attention.Transformer3DModel(8, c // 8, in_channels=c, num_layers=1, cross_attention_dim=1024,
                                             norm_num_groups=32, use_linear_projection=True, only_cross_attention=only_cross,
                                             rotary_emb=RotaryEmbedding(32)).eval()
        shapes = module_shapes(tr)
        seed = 950 + c
        tr.load_state_dict(synth16(shapes, seed))
        x = q16(torch.randn(2, c, frames, h, w, generator=g) * torch.linspace(0.5, 2.0, frames).reshape(1, 1, frames, 1, 1))
        ctx = q16(torch.randn(2, 77, 1024, generator=g))
        cases.append(dict(c=c, only_cross=only_cross, shapes=shapes, seed=seed, x=x.half(), ctx=ctx.half(),
                          y=tr(x, encoder_hidden_states=ctx).sample))
    save("vsr_transformer3d.pt", dict(cases=cases))

    # whole UNet3DVSRModel at full width (vsr/configs/unet_3d_config.json: 691 M parameters), 8x8, F = 5, B = 2, two noise levels
    import refbuild
    from lavie_amd import spec
    from lavie_amd.config import VSR_CONFIG
    seed = 6
    net = refbuild.reference_vsr_unet(VSR_CONFIG)
    net.load_state_dict(synth16(spec.param_shapes(VSR_CONFIG), seed))
    x = q16(torch.randn(2, 4, 5, 8, 8, generator=g))
    low = q16(torch.randn(2, 3, 5, 8, 8, generator=g))
    ctx = q16(torch.randn(2, 77, 1024, generator=g))
    labels = torch.tensor([20, 250])
    outs = {t: net(x, torch.tensor(t), low, encoder_hidden_states=ctx, class_labels=labels).sample for t in (900, 20)}
    save("vsr_unet_full_8x8.pt", dict(seed=seed, x=x.half(), low_res=low.half(), ctx=ctx.half(), labels=labels, y=outs))


if __name__ == "__main__":
    main()
