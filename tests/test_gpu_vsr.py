"""-m gpu: first pieces of the VSR stage through the HIP path: the (T,1,1) temporal convolution as an
implicit GEMM with a frame-tap table, and ResnetBlock3DCNN composed from it, against fixtures the reference's own
vsr/models/resnet.py produced and against the fp32 oracle."""
import pytest
import torch

import golden_util as G
from gpu_util import TOL_BLOCK, TOL_OP, f32, h16, rel_l2

pytestmark = pytest.mark.gpu


def to_rows(x):
    b, c, f, h, w = x.shape
    return x.permute(0, 2, 3, 4, 1).reshape(-1, c).contiguous()


def from_rows(r, b, f, h, w):
    return r.reshape(b, f, h, w, -1).permute(0, 4, 1, 2, 3)


@pytest.mark.parametrize("taps", [3, 5])
@pytest.mark.parametrize("shape", [(2, 64, 128, 8, 24), (1, 320, 320, 5, 33), (3, 128, 64, 2, 7), (1, 64, 640, 16, 40)])
def test_temporal_conv_vs_oracle(shape, taps):
    """frames fewer than the kernel reach, pixel counts that are not tile multiples, batch boundaries (frame 0 of video 1
    must not see the last frames of video 0), bias + per-video bias + residual."""
    from lavie_amd import ops
    from oracle import vsr_blocks as V
    b, cin, cout, frames, d = shape
    g = torch.Generator().manual_seed(cin + cout + frames + taps)
    x = torch.randn(b, cin, frames, 1, d, generator=g).half().float()
    w = (torch.randn(cout, cin, taps, 1, 1, generator=g) / (cin * taps) ** 0.5).half().float()
    bias = torch.randn(cout, generator=g)
    bias2 = torch.randn(b, cout, generator=g)
    res = torch.randn(b, cout, frames, 1, d, generator=g).half().float()
    ref = V.temporal_conv(x, w, bias) + bias2[:, :, None, None, None] + res
    wp = ops.pack_temporal_conv(h16(w))
    y = ops.temporal_conv(h16(to_rows(x)), wp, f32(bias), b, frames, d, taps, bias2=f32(bias2), residual=h16(to_rows(res)))
    assert rel_l2(from_rows(y.float().cpu(), b, frames, 1, d), ref) < TOL_OP
    # plain call (no per-video bias / residual)
    y0 = ops.temporal_conv(h16(to_rows(x)), wp, f32(bias), b, frames, d, taps)
    assert rel_l2(from_rows(y0.float().cpu(), b, frames, 1, d), V.temporal_conv(x, w, bias)) < TOL_OP


def test_vsr_resnet_block_3dcnn_golden():
    """ResnetBlock3DCNN (vsr/models/resnet.py:220-315) composed from the C-ABI operators: video-domain GroupNorm + SiLU,
    temporal conv with the time-embedding projection as per-video bias, GroupNorm + SiLU, temporal conv + residual."""
    import torch.nn.functional as F
    from lavie_amd import ops
    for c in G.load("vsr_resnet3dcnn.pt")["cases"]:
        sd = {k: v.cuda() for k, v in G.synth16(c["shapes"], c["seed"]).items()}
        x = c["x"].float()
        b, C, frames, h, w = x.shape
        d = h * w
        xr = h16(to_rows(x))
        tproj = F.linear(F.silu(c["temb"].float().cuda()), sd["time_emb_proj.weight"], sd["time_emb_proj.bias"]).contiguous()
        w1, w2 = ops.pack_temporal_conv(sd["conv1.weight"].half()), ops.pack_temporal_conv(sd["conv2.weight"].half())
        n1 = ops.group_norm(xr, sd["norm1.weight"], sd["norm1.bias"], b, 32, 1e-6, True)
        h1 = ops.temporal_conv(n1, w1, sd["conv1.bias"], b, frames, d, c["taps"], bias2=tproj)
        n2 = ops.group_norm(h1, sd["norm2.weight"], sd["norm2.bias"], b, 32, 1e-6, True)
        y = ops.temporal_conv(n2, w2, sd["conv2.bias"], b, frames, d, 3, residual=xr)
        assert rel_l2(from_rows(y.float().cpu(), b, frames, h, w), c["y"]) < TOL_BLOCK, (C, c["taps"], frames)


def test_vsr_transformer_golden():
    """The VSR Transformer3DModel variant through the engine seam (lavie_unet_transformer_forward): temporal resblock,
    attn1 as text cross-attention / self-attention, Linear projections — against the reference's outputs."""
    from lavie_amd import ops, spec
    from lavie_amd.config import UNetConfig
    from lavie_amd.vsr import UNet3DVSRModel
    for c in G.load("vsr_transformer3d.pt")["cases"]:
        C, oc = c["c"], c["only_cross"]
        cfg = UNetConfig(block_out_channels=(C,), attn_levels=(True,), layers_per_block=1, cross_attention_dim=1024,
                         vsr_blocks=True, only_cross_attention=(oc,))
        sd = G.synth16(spec.param_shapes(cfg), 5)
        blk = G.synth16(c["shapes"], c["seed"], "down_blocks.0.attentions.0.")
        assert set(blk) <= set(sd), set(blk) - set(sd)
        sd.update(blk)
        net = UNet3DVSRModel(init_weights=False, sample_size=8, block_out_channels=(C,), cross_attention_dim=1024,
                             layers_per_block=1, down_block_types=("CrossAttnDownBlock3D",),
                             up_block_types=("CrossAttnUpBlock3D",), only_cross_attention=(oc,))
        net.load_state_dict({k: v.half() for k, v in sd.items()})
        net = net.to("cuda", torch.float16)
        b, _, f, h, w = c["x"].shape
        y = ops.unet_transformer(net, "down_blocks.0.attentions.0", h16(to_rows(c["x"].float())), h16(c["ctx"]), b, f, h, w)
        assert rel_l2(from_rows(y.float().cpu(), b, f, h, w), c["y"]) < TOL_BLOCK, (C, oc)
