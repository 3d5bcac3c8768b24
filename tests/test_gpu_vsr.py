"""-m gpu: first pieces of the VSR stage through the HIP path: the (T,1,1) temporal convolution as an
implicit GEMM with a frame-tap table, and ResnetBlock3DCNN composed from it, against fixtures the reference's own
vsr/models/resnet.py produced and against the fp32 oracle."""
import pytest
import torch

import golden_util as G
from gpu_util import TOL_BLOCK, TOL_OP, TOL_UNET, f32, h16, rel_l2

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


@pytest.mark.parametrize("taps", [3, 5])
@pytest.mark.parametrize("shape", [(2, 64, 128, 8, 80), (1, 192, 256, 5, 128), (2, 128, 128, 4, 160), (1, 64, 128, 16, 60)])
def test_temporal_conv_patch_kernel(shape, taps):
    """The halo-patch kernel's temporal mode (a tile = every frame of 320 / F pixels, staged once per 64-channel slab and
    re-read at the T frame taps) forced on shapes it accepts — F = 8 / 5 / 4 / 16, two videos (frame 0 of video 1 must read
    zeros, not video 0), bias + per-video bias + residual, split-K over slabs — against the oracle and, bit for bit, against
    the default kernel."""
    from lavie_amd import _lib, ops
    from oracle import vsr_blocks as V
    lib = _lib.load()
    b, cin, cout, frames, d = shape
    g = torch.Generator().manual_seed(cin + cout + frames + taps + 1)
    x = torch.randn(b, cin, frames, 1, d, generator=g).half().float()
    w = (torch.randn(cout, cin, taps, 1, 1, generator=g) / (cin * taps) ** 0.5).half().float()
    bias = torch.randn(cout, generator=g)
    bias2 = torch.randn(b, cout, generator=g)
    res = torch.randn(b, cout, frames, 1, d, generator=g).half().float()
    ref = V.temporal_conv(x, w, bias) + bias2[:, :, None, None, None] + res
    wp = ops.pack_temporal_conv(h16(w))
    args = dict(bias2=f32(bias2), residual=h16(to_rows(res)))
    y_auto = ops.temporal_conv(h16(to_rows(x)), wp, f32(bias), b, frames, d, taps, **args)
    outs = []
    try:
        for splits in (0, cin // 64 if cin > 64 else 0):
            lib.lavie_debug_force_tile(5)
            lib.lavie_debug_force_splits(splits)
            outs.append(ops.temporal_conv(h16(to_rows(x)), wp, f32(bias), b, frames, d, taps, **args))
    finally:
        lib.lavie_debug_force_tile(0)
        lib.lavie_debug_force_splits(0)
    assert rel_l2(from_rows(outs[0].float().cpu(), b, frames, 1, d), ref) < TOL_OP
    assert torch.equal(outs[0], y_auto)                   # same summation order as the other kernels
    assert rel_l2(from_rows(outs[1].float().cpu(), b, frames, 1, d), ref) < TOL_OP


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
        cfg = UNetConfig(in_channels=7, block_out_channels=(C,), attn_levels=(True,), layers_per_block=1,
                         cross_attention_dim=1024, vsr_blocks=True, only_cross_attention=(oc,))
        sd = G.synth16(spec.param_shapes(cfg), 5)
        blk = G.synth16(c["shapes"], c["seed"], "down_blocks.0.attentions.0.")
        assert set(blk) <= set(sd), set(blk) - set(sd)
        sd.update(blk)
        net = UNet3DVSRModel(init_weights=False, sample_size=8, block_out_channels=(C,), cross_attention_dim=1024,
                             layers_per_block=1, down_block_types=("CrossAttnDownBlock3D",),
                             up_block_types=("CrossAttnUpBlock3D",), only_cross_attention=(oc,), num_class_embeds=None,
                             down_temporal_idx=(), mid_temporal=False, up_temporal_idx=())
        net.load_state_dict({k: v.half() for k, v in sd.items()})
        net = net.to("cuda", torch.float16)
        b, _, f, h, w = c["x"].shape
        y = ops.unet_transformer(net, "down_blocks.0.attentions.0", h16(to_rows(c["x"].float())), h16(c["ctx"]), b, f, h, w)
        assert rel_l2(from_rows(y.float().cpu(), b, f, h, w), c["y"]) < TOL_BLOCK, (C, oc)


SMALL_VSR = dict(sample_size=8, block_out_channels=(256, 512), down_block_types=("DownBlock3D", "CrossAttnDownBlock3D"),
                 up_block_types=("CrossAttnUpBlock3D", "UpBlock3D"), only_cross_attention=(True, False), layers_per_block=1,
                 cross_attention_dim=128, attention_head_dim=8, down_temporal_idx=(0, 1), mid_temporal=True,
                 up_temporal_idx=(0, 1))


def build_vsr(sd, **kw):
    from lavie_amd.vsr import UNet3DVSRModel
    net = UNet3DVSRModel(init_weights=False, **kw)
    net.load_state_dict({k: v.to(torch.float16) for k, v in sd.items()})
    return net.to("cuda", torch.float16)


@pytest.fixture(scope="module")
def small_vsr():
    from lavie_amd import spec
    from lavie_amd.config import UNetConfig
    cfg = UNetConfig(in_channels=7, block_out_channels=(256, 512), cross_attention_dim=128, attn_levels=(False, True),
                     layers_per_block=1, vsr_blocks=True, only_cross_attention=(True, False), vsr_temporal_modules=True,
                     num_class_embeds=1000)
    sd = G.synth16(spec.param_shapes(cfg), 31)
    return build_vsr(sd, **SMALL_VSR), sd


@pytest.mark.parametrize("shape", [(2, 5, 8, 8, 77), (1, 8, 8, 16, 77), (3, 2, 4, 4, 10)])
def test_vsr_whole_unet_small_vs_oracle(small_vsr, shape):
    """temporal modules after every block, class-embedded noise level, 4 + 3 input channels; ragged clip lengths."""
    from oracle import vsr_blocks as V
    net, sd = small_vsr
    b, f, h, w, n = shape
    g = torch.Generator().manual_seed(b * 100 + f)
    x, low = torch.randn(b, 4, f, h, w, generator=g).half(), torch.randn(b, 3, f, h, w, generator=g).half()
    ctx = torch.randn(b, n, 128, generator=g).half()
    t = torch.tensor([41.0 * (i + 1) for i in range(b)])
    labels = torch.tensor([20, 250, 7][:b])
    ref = V.vsr_unet_forward(sd, x.float(), low.float(), t, ctx.float(), labels, block_out_channels=(256, 512),
                             attn_levels=(False, True), only_cross_attention=(True, False), layers_per_block=1, heads=8)
    got = net(x.cuda(), t.cuda(), low.cuda(), encoder_hidden_states=ctx.cuda(), class_labels=labels).sample
    assert rel_l2(got, ref) < TOL_UNET


def test_vsr_whole_unet_full_width_golden():
    """UNet3DVSRModel at vsr/configs/unet_3d_config.json's width (691 M parameters) against the reference's own output."""
    from lavie_amd import spec
    from lavie_amd.config import VSR_CONFIG
    fx = G.load("vsr_unet_full_8x8.pt")
    sd = G.synth16(spec.param_shapes(VSR_CONFIG), fx["seed"])
    net = build_vsr(sd, sample_size=128, down_temporal_idx=(0, 1, 2, 3), mid_temporal=True, up_temporal_idx=(0, 1, 2, 3))
    for t, ref in fx["y"].items():
        got = net(fx["x"].cuda(), int(t), fx["low_res"].cuda(), encoder_hidden_states=fx["ctx"].cuda(),
                  class_labels=fx["labels"]).sample
        assert rel_l2(got, ref) < TOL_UNET, t
    with pytest.raises(ValueError):
        net(fx["x"].cuda(), 5, fx["low_res"].cuda(), encoder_hidden_states=fx["ctx"].cuda(), class_labels=torch.tensor([400, 1]))


def test_vsr_pipeline_loop_vs_oracle(small_vsr):
    """VideoUpscalePipeline.__call__ (noise-level conditioning of the low-res frames, DDIM loop through the fused
    guidance + step kernel, class labels = noise level) against oracle/vsr_loop.py with the same host noise; and the 8-frame
    chunk driver of vsr/sample.py:104-123."""
    from lavie_amd.scheduling_ddim import DDIMScheduler
    from lavie_amd.vsr import VideoUpscalePipeline, upscale_in_chunks
    from oracle import vsr_blocks as V
    from oracle.ddim import DDIMSchedule
    from oracle.vsr_loop import add_low_res_noise, vsr_denoise_loop
    net, sd = small_vsr
    pipe = VideoUpscalePipeline(unet=net, scheduler=DDIMScheduler())
    g = torch.Generator().manual_seed(77)
    pe, ne = torch.randn(1, 77, 128, generator=g).half().float(), torch.randn(1, 77, 128, generator=g).half().float()
    frames = torch.randn(1, 3, 4, 8, 8, generator=g).clamp(-1, 1)
    lat = torch.randn(1, 4, 4, 8, 8, generator=g)
    seen = []
    out = pipe(image=frames, prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, num_inference_steps=3, guidance_scale=9.0,
               noise_level=20, generator=torch.Generator().manual_seed(5), callback=lambda i, t, x: seen.append(t)).images
    assert seen == [667, 334, 1] and out.shape == (1, 4, 4, 8, 8) and torch.isfinite(out).all()
    noise = torch.randn(frames.shape, generator=torch.Generator().manual_seed(5))
    img = add_low_res_noise(frames, noise, 20)
    unet = lambda x, low, t, ctx, labels: V.vsr_unet_forward(sd, x, low, t, ctx, labels, block_out_channels=(256, 512),
                                                            attn_levels=(False, True), only_cross_attention=(True, False),
                                                            layers_per_block=1, heads=8)
    ref = vsr_denoise_loop(unet, lat, img, pe, ne, 20, 3, 9.0, schedule=DDIMSchedule())
    assert rel_l2(out, ref) < 3e-2
    # chunk driver: 10 frames -> chunks of 8 + 2, concatenated along the frame axis
    long = torch.randn(1, 3, 10, 8, 8, generator=g).clamp(-1, 1)
    up = upscale_in_chunks(pipe, long, short_seq=8, prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=2,
                           guidance_scale=9.0, noise_level=20, generator=torch.Generator().manual_seed(1))
    assert up.shape == (1, 4, 10, 8, 8) and torch.isfinite(up).all()
    with pytest.raises(ValueError, match="noise_level"):
        pipe(image=frames, prompt_embeds=pe, negative_prompt_embeds=ne, noise_level=400)


def test_pingpong_256_wide_tile_bit_identical():
    """The VSR widths (256 / 512 / 1024) are not multiples of 320: the ping-pong GEMM runs them with its 160x256 tile.
    Kernel choice must not change numerics: automatic selection == the 128-row kernel, bit for bit (plain GEMM with bias +
    residual, and a gathered 3x3 conv), and both match torch within the operator tolerance."""
    import torch.nn.functional as F
    from lavie_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    M, N, K = 20480, 512, 1024
    a = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = (torch.randn(M, N, generator=g) * 0.5).half().cuda()
    outs = []
    for mode in (0, 4):
        lib.lavie_debug_force_tile(mode)
        outs.append(ops.linear(a, w, bias=bias, residual=res))
    lib.lavie_debug_force_tile(0)
    assert torch.equal(outs[0], outs[1])
    ref = a.float() @ w.float().t() + bias + res.float()
    assert rel_l2(outs[0], ref) < TOL_OP
    ni, h, wd, c = 16, 32, 40, 256
    x = (torch.randn(ni, c, h, wd, generator=g) * 0.5).half()
    wt = (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5).half()
    xr = x.permute(0, 2, 3, 1).reshape(-1, c).contiguous().cuda()
    wp = ops.pack_conv3x3(wt.cuda())
    cb = torch.randn(c, generator=g).cuda()
    outs = []
    for mode in (0, 4):
        lib.lavie_debug_force_tile(mode)
        outs.append(ops.conv3x3(xr, wp, cb, ni, h, wd))
    lib.lavie_debug_force_tile(0)
    assert torch.equal(outs[0], outs[1])
    ref = F.conv2d(x.float(), wt.float(), cb.cpu(), padding=1).permute(0, 2, 3, 1).reshape(-1, c)
    assert rel_l2(outs[0], ref) < TOL_OP


def test_vsr_production_shape_properties():
    """The VSR UNet at the size vsr/sample.py runs it (691 M parameters, one 8-frame chunk of 320x512 latents, guidance
    batch 2; ~31 GB workspace) through size-independent properties: bit-reproducible, identical halves give bit-identical
    halves, finite, timestep / noise-level sensitive.  Also the shorter last chunk (F = 5: 61 = 7 x 8 + 5) right after it
    on the same handle: the per-F tables are cached, the answer is reproducible after switching back and forth."""
    from lavie_amd import spec, weights
    from lavie_amd.config import VSR_CONFIG
    from lavie_amd.vsr import UNet3DVSRModel
    sd = weights.synth_state_dict(spec.param_shapes(VSR_CONFIG), 0)
    net = UNet3DVSRModel(init_weights=False, sample_size=128, down_temporal_idx=(0, 1, 2, 3), mid_temporal=True,
                         up_temporal_idx=(0, 1, 2, 3))
    net.load_state_dict({k: v.half() for k, v in sd.items()})
    del sd
    net = net.to("cuda", torch.float16)
    g = torch.Generator().manual_seed(8)
    x1 = torch.randn(1, 4, 8, 320, 512, generator=g).half()
    l1 = torch.randn(1, 3, 8, 320, 512, generator=g).half()
    c1 = torch.randn(1, 77, 1024, generator=g).half()
    x, low, ctx = torch.cat([x1, x1]).cuda(), torch.cat([l1, l1]).cuda(), torch.cat([c1, c1]).cuda()
    labels = torch.tensor([20, 20])
    y = net(x, 500, low, encoder_hidden_states=ctx, class_labels=labels).sample
    assert torch.isfinite(y).all() and torch.equal(y[0], y[1])
    y5 = net(x[:, :, :5], 500, low[:, :, :5], encoder_hidden_states=ctx, class_labels=labels).sample      # last chunk: F = 5
    assert torch.isfinite(y5).all() and torch.equal(y5[0], y5[1])
    yb = net(x, 500, low, encoder_hidden_states=ctx, class_labels=labels).sample
    assert torch.equal(y, yb)
    y5b = net(x[:, :, :5], 500, low[:, :, :5], encoder_hidden_states=ctx, class_labels=labels).sample
    assert torch.equal(y5, y5b)
    assert rel_l2(net(x, 20, low, encoder_hidden_states=ctx, class_labels=labels).sample, y) > 1e-3
    assert rel_l2(net(x, 500, low, encoder_hidden_states=ctx, class_labels=torch.tensor([300, 300])).sample, y) > 1e-4


def test_vsr_groupnorm_statistics_from_producers_verified():
    """Round 4: in the VSR UNet the GroupNorms behind the temporal (T,1,1) convs, behind `TemporalModule3D`'s shift conv and behind the
    module's inputs fold the producing kernel's epilogue sums too (temporal-conv tiles of the halo-patch kernel: a block's rows are
    scattered over ONE video, `GnColStat::span`).  Verify mode (bit 6 of lavie_debug_fused_mask) runs the statistics pass beside every
    such norm and compares (mean, rstd) per (batch, group) on the host; the forward's result must not depend on the mode.  Production
    width at a reduced frame size (64 x 64: every level's tiles, split-K reduces at the deep levels) with 8 and 5 frames, and the
    full 320 x 512 chunk."""
    from lavie_amd import _lib, spec, weights
    from lavie_amd.config import VSR_CONFIG
    from lavie_amd.vsr import UNet3DVSRModel
    lib = _lib.load()
    DEF = _lib.FUSED_DEFAULT
    sd = weights.synth_state_dict(spec.param_shapes(VSR_CONFIG), 0)
    net = UNet3DVSRModel(init_weights=False, sample_size=128, down_temporal_idx=(0, 1, 2, 3), mid_temporal=True, up_temporal_idx=(0, 1, 2, 3))
    net.load_state_dict({k: v.half() for k, v in sd.items()})
    del sd
    net = net.to("cuda", torch.float16)
    g = torch.Generator().manual_seed(12)
    labels = torch.tensor([20, 20])
    try:
        for F_, H, W in ((8, 64, 64), (5, 64, 64), (3, 64, 64), (8, 320, 512)):
            x = torch.randn(2, 4, F_, H, W, generator=g).half().cuda()
            low = torch.randn(2, 3, F_, H, W, generator=g).half().cuda()
            ctx = torch.randn(2, 77, 1024, generator=g).half().cuda()
            outs = {}
            for mask in (DEF & ~32, DEF):
                _lib.check(lib.lavie_debug_fused_mask(mask), "lavie_debug_fused_mask")
                outs[mask] = net(x, 500, low, encoder_hidden_states=ctx, class_labels=labels).sample.clone()
            assert torch.isfinite(outs[DEF]).all()
            assert rel_l2(outs[DEF], outs[DEF & ~32]) < 2e-3, (F_, H, W, rel_l2(outs[DEF], outs[DEF & ~32]))
            n0 = lib.lavie_debug_gn_producer_count()
            _lib.check(lib.lavie_debug_fused_mask(DEF | 64), "lavie_debug_fused_mask")
            got = net(x, 500, low, encoder_hidden_states=ctx, class_labels=labels).sample
            n = lib.lavie_debug_gn_producer_count() - n0
            assert torch.equal(got, outs[DEF])
            assert n >= 90, (F_, H, W, n)          # of the 129 GroupNorms of the VSR forward (conv_in's consumers and per-frame norms whose frames are no whole blocks keep the pass)
            del x, low, ctx, outs, got
    finally:
        lib.lavie_debug_fused_mask(DEF)
