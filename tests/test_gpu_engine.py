"""-m gpu: the engine (packed weights + kernel sequencing) against reference-generated fixtures and the oracle."""
import pytest
import torch

import golden_util as G
from gpu_util import TOL_BLOCK, TOL_OP, TOL_UNET, f32, h16, rel_l2

pytestmark = pytest.mark.gpu

SMALL_KW = dict(sample_size=8, block_out_channels=(256, 512), cross_attention_dim=128,
                down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"), up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"))


def build(sd, **kw):
    from lavie_amd.unet import UNet3DConditionModel
    net = UNet3DConditionModel(init_weights=False, **kw)
    net.load_state_dict({k: v.to(torch.float16) for k, v in sd.items()})
    return net.to("cuda", torch.float16)


@pytest.fixture(scope="module")
def small():
    from lavie_amd import spec, weights
    from lavie_amd.config import UNetConfig
    cfg = UNetConfig(block_out_channels=(256, 512), cross_attention_dim=128, attn_levels=(True, False))
    sd = G.synth16(spec.param_shapes(cfg), 11)
    return build(sd, **SMALL_KW), sd


@pytest.fixture(scope="module")
def full():
    from lavie_amd import spec
    sd = G.synth16(spec.param_shapes(), 0)
    net = build(sd, sample_size=64, cross_attention_dim=768)
    return net, sd


def ocfg_small():
    from oracle import unet_fp32 as O
    return O.UNetConfig(block_out_channels=(256, 512), cross_attention_dim=128, attn_levels=(True, False))


def to_rows(x):
    b, c, f, h, w = x.shape
    return x.permute(0, 2, 3, 4, 1).reshape(-1, c).contiguous()


def from_rows(r, b, f, h, w):
    return r.reshape(b, f, h, w, -1).permute(0, 4, 1, 2, 3)


# ------------------------------------------------------------------ sub-module seams
def test_resnet_blocks_vs_oracle(small):
    from lavie_amd import ops
    from oracle import unet_fp32 as O
    net, sd = small
    g = torch.Generator().manual_seed(1)
    b, f, h, w = 2, 4, 8, 8
    temb = torch.randn(b, 1024, generator=g)
    cases = [("down_blocks.0.resnets.0", 256, 0, h, w),       # cin == cout: plain residual
             ("down_blocks.1.resnets.0", 256, 0, 4, 4),       # 256 -> 512: fused 1x1 shortcut
             ("up_blocks.1.resnets.2", 256, 256, h, w),       # skip concat [x | skip] + shortcut
             ("up_blocks.0.resnets.0", 512, 512, 4, 4)]
    for prefix, c1, c2, hh, ww in cases:
        x1 = G.synth16({"x": (b, c1, f, hh, ww)}, 50)["x"] * 3
        x2 = G.synth16({"x": (b, c2, f, hh, ww)}, 51)["x"] * 3 if c2 else None
        xin = x1 if x2 is None else torch.cat([x1, x2], 1)
        ref = O.resnet_block(sd, prefix + ".", xin, temb, ocfg_small())
        y = ops.unet_resnet_block(net, prefix, h16(to_rows(x1)), None if x2 is None else h16(to_rows(x2)), f32(temb),
                                  b, f, hh, ww)
        assert rel_l2(from_rows(y.float().cpu(), b, f, hh, ww), ref) < TOL_BLOCK, prefix


def test_transformer_vs_oracle(small):
    from lavie_amd import ops
    from oracle import unet_fp32 as O
    net, sd = small
    g = torch.Generator().manual_seed(2)
    b, f, h, w = 2, 16, 4, 6
    x = (torch.randn(b, 256, f, h, w, generator=g) * torch.linspace(0.5, 2, f).reshape(1, 1, f, 1, 1)).half().float()
    ctx = torch.randn(b, 77, 128, generator=g).half().float()
    ref = O.transformer3d(sd, "down_blocks.0.attentions.0.", x, ctx, ocfg_small())
    y = ops.unet_transformer(net, "down_blocks.0.attentions.0", h16(to_rows(x)), h16(ctx), b, f, h, w)
    assert rel_l2(from_rows(y.float().cpu(), b, f, h, w), ref) < TOL_BLOCK


def test_resnet_golden_full_width_seams():
    """reference-generated ResnetBlock3D fixture (64->128 with shortcut, 128->128) through conv/norm operators."""
    from lavie_amd import ops
    import torch.nn.functional as F
    fx = G.load("resnet.pt")
    for c in fx["cases"]:
        sd = G.synth16(c["shapes"], c["seed"])
        x, temb = c["x"].float(), c["temb"].float()
        b, cin, f, h, w = x.shape
        cout = c["cout"]
        xr = h16(to_rows(x))
        tproj = F.linear(F.silu(temb), sd["time_emb_proj.weight"], sd["time_emb_proj.bias"])
        n1 = ops.group_norm(xr, f32(sd["norm1.weight"]), f32(sd["norm1.bias"]), nb=b, groups=32, eps=1e-5, silu=True)
        h1 = ops.conv3x3(n1, ops.pack_conv3x3(h16(sd["conv1.weight"])), f32(sd["conv1.bias"]), b * f, h, w,
                         bias2=f32(tproj), rows_per_batch=f * h * w)
        n2 = ops.group_norm(h1, f32(sd["norm2.weight"]), f32(sd["norm2.bias"]), nb=b, groups=32, eps=1e-5, silu=True)
        if cin != cout:
            wp = ops.pack_conv3x3(h16(sd["conv2.weight"]), h16(sd["conv_shortcut.weight"]))
            y = ops.conv3x3(n2, wp, f32(sd["conv2.bias"] + sd["conv_shortcut.bias"]), b * f, h, w, sc1=xr)
        else:
            y = ops.conv3x3(n2, ops.pack_conv3x3(h16(sd["conv2.weight"])), f32(sd["conv2.bias"]), b * f, h, w, residual=xr)
        assert rel_l2(from_rows(y.float().cpu(), b, f, h, w), c["y"]) < TOL_BLOCK
    s = fx["sampler"]
    x = s["x"].float()
    b, cch, f, h, w = x.shape
    up = G.synth16(s["up_shapes"], s["up_seed"])
    dn = G.synth16(s["dn_shapes"], s["dn_seed"])
    yu = ops.conv3x3(h16(to_rows(x)), ops.pack_conv3x3(h16(up["conv.weight"])), f32(up["conv.bias"]), b * f, h, w, ups=1)
    yd = ops.conv3x3(h16(to_rows(x)), ops.pack_conv3x3(h16(dn["conv.weight"])), f32(dn["conv.bias"]), b * f, h, w, stride=2)
    assert rel_l2(from_rows(yu.float().cpu(), b, f, 2 * h, 2 * w), s["up"]) < TOL_OP
    assert rel_l2(from_rows(yd.float().cpu(), b, f, h // 2, w // 2), s["dn"]) < TOL_OP


def test_temporal_attention_golden():
    """reference TemporalAttention fixtures (F = 16 and the interpolation model's F = 61) through the projection GEMMs +
    the temporal core kernel."""
    from lavie_amd import ops
    from oracle import unet_fp32 as O
    for c in G.load("temporal_attention.pt")["cases"]:
        sd = G.synth16(c["shapes"], c["seed"])
        x = c["x"].float()                            # [(b d), f, C] with b = 1
        nseq, f, ch = x.shape
        tokens = x.permute(1, 0, 2).reshape(f * nseq, ch)             # (f, pixel) order, b = 1
        wqkv = torch.cat([sd["to_q.weight"], sd["to_k.weight"], sd["to_v.weight"]], 0)
        qkv = ops.linear(h16(tokens), h16(wqkv))
        bias = sd["time_rel_pos_bias.relative_attention_bias.weight"][O.rel_pos_bucket_table(f, 32, 32)].permute(2, 0, 1)
        cos, sin = ops.rotary_tables(f, 32)
        o = ops.temporal_attention(qkv, 1, f, nseq, 8, f32(bias.contiguous()), cos, sin)
        y = ops.linear(o, h16(sd["to_out.0.weight"]), bias=f32(sd["to_out.0.bias"]))
        got = y.float().cpu().reshape(f, nseq, ch).permute(1, 0, 2)
        assert rel_l2(got, c["y"]) < TOL_BLOCK, (ch, f)


def test_cross_attention_golden():
    from lavie_amd import ops
    for c in G.load("cross_attention.pt")["cases"]:
        sd = G.synth16(c["shapes"], c["seed"])
        x = c["x"].float()
        nb, d, ch = x.shape
        xr = h16(x.reshape(-1, ch))
        if c["ctx"] is None:
            wqkv = torch.cat([sd["to_q.weight"], sd["to_k.weight"], sd["to_v.weight"]], 0)
            qkv = ops.linear(xr, h16(wqkv))
            o = ops.attention(qkv[:, :ch], qkv[:, ch:2 * ch], qkv[:, 2 * ch:], nb=nb, lq=d, lk=d, heads=8)
        else:
            ctx = h16(c["ctx"].float().reshape(-1, c["ctx"].shape[-1]))
            q = ops.linear(xr, h16(sd["to_q.weight"]))
            kv = ops.linear(ctx, h16(torch.cat([sd["to_k.weight"], sd["to_v.weight"]], 0)))
            o = ops.attention(q, kv[:, :ch], kv[:, ch:], nb=nb, lq=d, lk=77, heads=8)
        y = ops.linear(o, h16(sd["to_out.0.weight"]), bias=f32(sd["to_out.0.bias"]))
        assert rel_l2(y.float().cpu().reshape(nb, d, ch), c["y"]) < TOL_BLOCK


# ------------------------------------------------------------------ whole UNet
@pytest.mark.parametrize("t", [980, 500, 0])
def test_whole_unet_small_vs_oracle(small, t):
    from oracle import unet_fp32 as O
    net, sd = small
    g = torch.Generator().manual_seed(100 + t)
    x = torch.randn(2, 4, 16, 8, 8, generator=g).half()
    ctx = torch.randn(2, 77, 128, generator=g).half()
    ref = O.unet_forward(sd, x.float(), t, ctx.float(), ocfg_small())
    got = net(x.cuda(), t, encoder_hidden_states=ctx.cuda()).sample
    assert got.dtype == torch.float16 and got.shape == ref.shape
    assert rel_l2(got, ref) < TOL_UNET
    got2 = net(x.cuda(), torch.tensor(t), encoder_hidden_states=ctx.cuda(), return_dict=False)[0]
    assert torch.equal(got, got2)                                       # no atomics anywhere: bit-reproducible


def test_layernorm_fold_matches_explicit_layernorm(small):
    """The folded LayerNorm path (default) and the explicit LayerNorm kernels agree, and both match the oracle."""
    from lavie_amd import _lib
    from oracle import unet_fp32 as O
    net, sd = small
    g = torch.Generator().manual_seed(808)
    x = torch.randn(2, 4, 16, 8, 8, generator=g).half()
    ctx = torch.randn(2, 77, 128, generator=g).half()
    ref = O.unet_forward(sd, x.float(), 640, ctx.float(), ocfg_small())
    lib, handle = _lib.load(), net.engine_handle()
    try:
        _lib.check(lib.lavie_unet_set_ln_fold(handle, 0))
        explicit = net(x.cuda(), 640, encoder_hidden_states=ctx.cuda()).sample
    finally:
        _lib.check(lib.lavie_unet_set_ln_fold(handle, 1))
    folded = net(x.cuda(), 640, encoder_hidden_states=ctx.cuda()).sample
    assert rel_l2(explicit, ref) < TOL_UNET and rel_l2(folded, ref) < TOL_UNET
    assert rel_l2(folded, explicit) < 5e-3


def test_whole_unet_other_shapes(small):
    """ragged sizes: F=5, 8x16 latent, batch 1 and 3, 10 context tokens."""
    from oracle import unet_fp32 as O
    net, sd = small
    for b, f, h, w, n in ((1, 5, 8, 16, 77), (3, 2, 4, 4, 10)):
        g = torch.Generator().manual_seed(b * 10 + f)
        x = torch.randn(b, 4, f, h, w, generator=g).half()
        ctx = torch.randn(b, n, 128, generator=g).half()
        t = torch.tensor([10.0 * (i + 1) for i in range(b)])
        ref = O.unet_forward(sd, x.float(), t, ctx.float(), ocfg_small())
        got = net(x.cuda(), t.cuda(), encoder_hidden_states=ctx.cuda()).sample
        assert rel_l2(got, ref) < TOL_UNET


def test_whole_unet_long_clip(small):
    """F = 24 frames (> 16): the 64-frame temporal tile path inside the whole UNet."""
    from oracle import unet_fp32 as O
    net, sd = small
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, 4, 24, 8, 8, generator=g).half()
    ctx = torch.randn(1, 77, 128, generator=g).half()
    ref = O.unet_forward(sd, x.float(), 321, ctx.float(), ocfg_small())
    got = net(x.cuda(), 321, encoder_hidden_states=ctx.cuda()).sample
    assert rel_l2(got, ref) < TOL_UNET


def test_unet_rejects_bad_shapes(small):
    net, _ = small
    with pytest.raises(RuntimeError, match="multiples"):
        net(torch.zeros(2, 4, 4, 7, 8, device="cuda"), 1, encoder_hidden_states=torch.zeros(2, 77, 128, device="cuda"))
    with pytest.raises(ValueError):
        net(torch.zeros(2, 4, 8, 8, device="cuda"), 1, encoder_hidden_states=torch.zeros(2, 77, 128, device="cuda"))


def test_whole_unet_full_width_golden(full):
    """909 M-parameter model, latent 8x8, F=16, B=2 against the fixture the reference produced."""
    net, _ = full
    fx = G.load("unet_full_8x8.pt")
    for t, ref in fx["y"].items():
        got = net(fx["x"].cuda(), int(t), encoder_hidden_states=fx["ctx"].cuda()).sample
        assert rel_l2(got, ref) < TOL_UNET, t


def test_full_size_properties(full):
    """BASELINE.json's full shape (909 M parameters, F=16, latent 40x64) through size-independent properties:
    bit-reproducibility, batch independence (the CFG batch of two identical halves returns identical halves, and a
    batch-1 call — different GEMM tile / split-K choices at half the rows — agrees with them), finite outputs."""
    net, _ = full
    g = torch.Generator().manual_seed(5)
    x1 = torch.randn(1, 4, 16, 40, 64, generator=g).half()
    c1 = torch.randn(1, 77, 768, generator=g).half()
    x2, c2 = torch.cat([x1, x1]).cuda(), torch.cat([c1, c1]).cuda()
    y2 = net(x2, 500, encoder_hidden_states=c2).sample
    y2b = net(x2, 500, encoder_hidden_states=c2).sample
    assert torch.equal(y2, y2b)
    assert torch.isfinite(y2).all()
    assert torch.equal(y2[0], y2[1])
    y1 = net(x1.cuda(), 500, encoder_hidden_states=c1.cuda()).sample
    assert rel_l2(y1[0], y2[0]) < 5e-3
    # a different timestep / context must change the answer (guards against stale-buffer reuse)
    y3 = net(x2, 20, encoder_hidden_states=c2).sample
    assert rel_l2(y3, y2) > 1e-2


def test_full_size_forward_vs_oracle(full):
    """BASELINE.json configs[1]'s own forward — 909 M parameters, CFG batch 2, F = 16, latent 40x64, bench.py's
    synth_inputs(0) — against oracle.unet_fp32.unet_forward (the fp32 restatement of unet.py:366-512, pinned to the
    imported reference in tests/test_oracle_vs_reference.py) on the host cores (~35 s on 16 threads).  The only place where
    the kernels the planner selects at 40x64 and nowhere below meet the oracle: the persistent ping-pong GEMM
    (>= 256 tiles), the halo-patch conv on 512-workgroup grids, 40-key-tile self-attention, multi-tile temporal
    streaming, and the fused temporal / GEGLU sub-block kernels at their production tile counts."""
    import bench
    from lavie_amd import _lib
    from oracle import unet_fp32 as O
    net, sd = full
    pe, ne, lat = bench.synth_inputs(0, "cpu")
    ctx = torch.cat([ne, pe]).half()
    x = torch.cat([lat, lat]).half()
    torch.set_num_threads(bench.host_cores())
    for t in (500,):
        with torch.no_grad():
            ref = O.unet_forward(sd, x.float(), t, ctx.float())
        got = net(x.cuda(), t, encoder_hidden_states=ctx.cuda()).sample
        assert got.shape == ref.shape == (2, 4, 16, 40, 64)
        assert torch.isfinite(got).all()
        assert rel_l2(got, ref) < TOL_UNET, (t, rel_l2(got, ref))
        # the two CFG halves see different text: both must match on their own
        assert rel_l2(got[0], ref[0]) < TOL_UNET and rel_l2(got[1], ref[1]) < TOL_UNET
        # the same forward as the denoise loop runs it: text K / V cached per prompt, which is what lets the level-0 blocks take
        # the fused text cross-attention kernel (K / V inside its weight stream) — counted, so that a silent fallback fails here
        lib = _lib.load()
        cc = net.cache_context(ctx.cuda())
        bench.profile_begin(lib, 1 << 10, 64)
        got_c = net(x.cuda(), t, encoder_hidden_states=cc).sample
        rows = bench.profile_end(lib)
        net.cache_context(None)
        assert rows[10]["launches"] == 5, rows[10]            # down 0 (x2), up 3 (x3)
        assert rel_l2(got_c, ref) < TOL_UNET, rel_l2(got_c, ref)
        assert rel_l2(got_c[0], ref[0]) < TOL_UNET and rel_l2(got_c[1], ref[1]) < TOL_UNET
        # and as the guided denoise loop runs it: both halves are the same latents, the layers in front of the first text
        # cross-attention computed once (lavie_unet_set_cfg_shared_input)
        try:
            net.set_cfg_shared_input(True)
            cc = net.cache_context(ctx.cuda())
            got_s = net(x.cuda(), t, encoder_hidden_states=cc).sample
        finally:
            net.cache_context(None)
            net.set_cfg_shared_input(False)
        assert rel_l2(got_s, ref) < TOL_UNET, rel_l2(got_s, ref)
        assert rel_l2(got_s[0], ref[0]) < TOL_UNET and rel_l2(got_s[1], ref[1]) < TOL_UNET


def test_full_size_graph_replay_matches_eager(full):
    """hipGraph replay at the production shape, the way the guided loop would drive it: cached context + shared CFG prefix, so the
    capture contains the three row-resident level-0 kernels, the parity-form upsample convs and the half-batch copies of the shared
    prefix (none of which the reduced-width graph tests reach).  Eager, capture, replay: bit-equal to the eager forward."""
    import bench
    net, _ = full
    pe, ne, lat = bench.synth_inputs(0, "cpu")
    ctx = torch.cat([ne, pe]).half().cuda()
    xs = [torch.cat([lat, lat]).half().cuda() * s for s in (1.0, 0.5, 0.25, 0.75)]
    ts = [900, 600, 300, 100]
    try:
        cc = net.cache_context(ctx)
        net.set_cfg_shared_input(True)
        ref = [net(x, t, encoder_hidden_states=cc).sample.clone() for x, t in zip(xs, ts)]
        net.enable_graph(True)
        buf = torch.empty_like(xs[0])
        for i, (x, t) in enumerate(zip(xs, ts)):          # call 0 eager, 1 captures, 2.. replay
            buf.copy_(x)
            assert torch.equal(net(buf, t, encoder_hidden_states=cc).sample, ref[i]), f"step {i}"
    finally:
        net.enable_graph(False)
        net.set_cfg_shared_input(False)
        net.cache_context(None)


def test_debug_check_shared_catches_unequal_halves(small, monkeypatch):
    """LAVIE_DEBUG_CHECK_SHARED=1: with set_cfg_shared_input(True) the engine verifies sample[b] == sample[b + B/2] before it
    computes the shared layers once; equal halves pass, unequal halves are an error instead of a silently wrong conditional half."""
    net, _ = small
    g = torch.Generator().manual_seed(5)
    half = torch.randn(1, 4, 4, 8, 8, generator=g).half().cuda()
    other = torch.randn(1, 4, 4, 8, 8, generator=g).half().cuda()
    ctx = torch.randn(2, 77, 128, generator=g).half().cuda()
    monkeypatch.setenv("LAVIE_DEBUG_CHECK_SHARED", "1")
    try:
        net.set_cfg_shared_input(True)
        net(torch.cat([half, half]), 500, encoder_hidden_states=ctx)
        with pytest.raises(RuntimeError, match="halves of the sample differ"):
            net(torch.cat([half, other]), 500, encoder_hidden_states=ctx)
        with pytest.raises(RuntimeError, match="even batch"):            # the switch on a batch it cannot apply to
            net(half, 500, encoder_hidden_states=ctx[:1])
    finally:
        net.set_cfg_shared_input(False)


def test_groupnorm_statistics_from_producers_match_the_statistics_pass(full, small):
    """Round 4: the kernels that write a GroupNorm's input leave its per-channel sums in their epilogues (conv / GEMM epilogue, the
    persistent GEMM's finish pass, the split-K reduce, the parity-form upsample conv) and the GroupNorm folds those instead of reading
    the tensor for statistics (lavie_debug_fused_mask bit 5).  Both paths sum the same rounded fp16 values in fp32, in different
    orders: the forward must agree to well below the fp16 resolution of the activations, at the production shape (every producer
    kernel the planner picks there, skip concatenations with straddling groups, per-frame and per-video domains, the shared CFG
    prefix) and at the reduced width (ragged tiles: the fallback to the statistics pass)."""
    import bench
    from lavie_amd import _lib
    lib = _lib.load()
    DEF = _lib.FUSED_DEFAULT
    net, _ = full
    pe, ne, lat = bench.synth_inputs(0, "cpu")
    ctx = torch.cat([ne, pe]).half().cuda()
    x = torch.cat([lat, lat]).half().cuda()
    try:
        outs = {}
        for shared in (False, True):
            cc = net.cache_context(ctx)
            net.set_cfg_shared_input(shared)
            for mask in (DEF & ~32, DEF):
                _lib.check(lib.lavie_debug_fused_mask(mask), "lavie_debug_fused_mask")
                outs[(shared, mask)] = net(x, 500, encoder_hidden_states=cc).sample.clone()
            assert torch.equal(outs[(shared, DEF)], net(x, 500, encoder_hidden_states=cc).sample)      # fixed summation order
            # two equivalent fp32 summation orders flip a few fp16 roundings, which the following 100+ layers decorrelate: the
            # level every other equivalent kernel switch of this engine shows (7e-4); the DIRECT check is the verify pass below
            assert rel_l2(outs[(shared, DEF)], outs[(shared, DEF & ~32)]) < 2e-3, (shared, rel_l2(outs[(shared, DEF)], outs[(shared, DEF & ~32)]))
            # bit 6: every GroupNorm that takes producer statistics also runs the statistics pass and compares (mean, rstd) per
            # (batch, group) on the host.  58 of the 61 GroupNorms qualify: not the two that read conv_in's output (its kernel
            # leaves no statistics) and not the mid block's per-frame norm (40 rows per frame: no whole statistics block)
            n0 = lib.lavie_debug_gn_producer_count()
            _lib.check(lib.lavie_debug_fused_mask(DEF | 64), "lavie_debug_fused_mask")
            got = net(x, 500, encoder_hidden_states=cc).sample
            assert lib.lavie_debug_gn_producer_count() - n0 == 58, lib.lavie_debug_gn_producer_count() - n0
            assert torch.equal(got, outs[(shared, DEF)])
            net.set_cfg_shared_input(False)
            net.cache_context(None)
        snet, _ = small
        g = torch.Generator().manual_seed(9)
        xs = torch.randn(2, 4, 4, 8, 8, generator=g).half().cuda()
        cs = torch.randn(2, 77, 128, generator=g).half().cuda()
        ys = {}
        for mask in (DEF & ~32, DEF):
            _lib.check(lib.lavie_debug_fused_mask(mask), "lavie_debug_fused_mask")
            ys[mask] = snet(xs, 300, encoder_hidden_states=cs).sample.clone()
        assert rel_l2(ys[DEF], ys[DEF & ~32]) < 2e-3
    finally:
        lib.lavie_debug_fused_mask(DEF)
        net.set_cfg_shared_input(False)
        net.cache_context(None)


def test_three_ddpm_steps_golden(full):
    """VideoGenPipeline loop (CFG + fused DDPM step) for 3 steps against the reference-UNet trajectory."""
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    from lavie_amd.scheduling_ddpm import DDPMScheduler
    net, _ = full
    fx = G.load("ddpm_3step.pt")
    pipe = VideoGenPipeline(unet=net, scheduler=DDPMScheduler())
    noises = iter(fx["noises"])

    class Replay(torch.Generator):          # hands the fixture's noise tensors to the loop, in order
        pass

    # drive the loop manually so that the fixture's noise is used verbatim
    from lavie_amd import ops
    sch = pipe.scheduler
    sch.set_timesteps(50)
    x = fx["latents"].cuda().float().contiguous()
    ctx = torch.cat([fx["negative"], fx["prompt"]]).cuda().half().contiguous()
    model_in = torch.empty((2,) + tuple(x.shape[1:]), dtype=torch.float16, device="cuda")
    ops.latents_to_model_input(x, model_in)
    for i, t in enumerate([int(v) for v in sch.timesteps][:3]):
        eps = net(model_in, t, encoder_hidden_states=ctx).sample
        ops.cfg_ddpm_step(eps, x, next(noises).cuda().float().contiguous(), model_in, 7.5, sch.coefficients(t))
    assert rel_l2(x, fx["y"]) < TOL_UNET


def test_fifty_ddpm_steps_reference_trajectory(full):
    """BASELINE's north star: "outputs match the reference CPU path on the same seed/prompt within a stated fp16
    tolerance".  All 50 CFG + DDPM steps of VideoGenPipeline.__call__ (latents=, prompt_embeds=, CPU generator=) at full
    width against tests/golden/ddpm_50step.pt — the reference loop (pipeline_videogen.py:662-689) around the imported
    reference UNet in fp32, noise from the same CPU generator seed.  Stated tolerance for fp16 storage over the whole
    stochastic trajectory: rel-L2 <= 5e-3 and cosine >= 0.9999 on the final latents, rel-L2 <= 5e-3 on every kept
    intermediate (measured on MI355X: 0.9e-3 .. 1.1e-3 at every kept step, cosine 0.9999994; SURVEY.md §8c had expected
    1e-2 .. 5e-2)."""
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    from lavie_amd.scheduling_ddpm import DDPMScheduler
    net, _ = full
    fx = G.load("ddpm_50step.pt")
    pipe = VideoGenPipeline(unet=net, scheduler=DDPMScheduler())
    seen = {}
    out = pipe(prompt_embeds=fx["prompt"], negative_prompt_embeds=fx["negative"], latents=fx["latents"], height=64, width=64,
               video_length=16, num_inference_steps=fx["steps"], guidance_scale=fx["guidance_scale"],
               generator=torch.Generator().manual_seed(fx["noise_seed"]), output_type="latent",
               callback=lambda i, t, x: seen.__setitem__(i, x.float().cpu().clone())).video.float().cpu()
    errs = {i: rel_l2(seen[i], ref) for i, ref in fx["kept"].items()}
    cos = torch.nn.functional.cosine_similarity(out.flatten(), fx["y"].flatten(), dim=0).item()
    print("trajectory rel-L2 per kept step:", {i: f"{e:.2e}" for i, e in errs.items()}, "final cosine", cos)
    assert torch.equal(out, seen[49])
    assert max(errs.values()) < 5e-3, errs
    assert rel_l2(out, fx["y"]) < 5e-3 and cos > 0.9999


def test_transformer3d_reference_fixture_base_width():
    """tests/golden/transformer3d.pt — the reference's Transformer3DModel at the base model's first-level width
    (C = 320, head dim 40, per-frame GroupNorm eps 1e-6, order spatial -> text -> temporal -> FF) — through
    lavie_unet_transformer_forward; the fixture's block weights replace those of a two-level model's first block."""
    from lavie_amd import ops, spec
    from lavie_amd.config import UNetConfig
    fx = G.load("transformer3d.pt")
    cfg = UNetConfig(block_out_channels=(320, 640), attn_levels=(True, False))
    sd = G.synth16(spec.param_shapes(cfg), 3)
    prefix = "down_blocks.0.attentions.0"
    blk = G.synth16(fx["shapes"], fx["seed"], prefix + ".")
    assert set(blk) == {k for k in sd if k.startswith(prefix + ".")}
    sd.update(blk)
    net = build(sd, sample_size=8, block_out_channels=(320, 640), cross_attention_dim=768,
                down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"), up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"))
    x = fx["x"].float()
    b, c, f, h, w = x.shape
    y = ops.unet_transformer(net, prefix, h16(to_rows(x)), h16(fx["ctx"]), b, f, h, w)
    assert rel_l2(from_rows(y.float().cpu(), b, f, h, w), fx["y"]) < TOL_BLOCK


def test_pipeline_call_surface(small):
    """__call__ with prompt_embeds / latents / CPU generator / callback, output_type='latent'; deterministic."""
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    net, sd = small
    pipe = VideoGenPipeline(unet=net)
    g = torch.Generator().manual_seed(9)
    pe, ne = torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
    lat = torch.randn(1, 4, 4, 8, 8, generator=g)
    seen = []
    outs = []
    for _ in range(2):
        out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, height=64, width=64, video_length=4,
                   num_inference_steps=4, guidance_scale=7.5, generator=torch.Generator().manual_seed(3),
                   output_type="latent", callback=lambda i, t, x: seen.append((i, t)), callback_steps=1).video
        outs.append(out.float().cpu())
    assert outs[0].shape == (1, 4, 4, 8, 8) and torch.isfinite(outs[0]).all()
    assert torch.equal(outs[1], outs[0])
    assert seen[:4] == [(0, 750), (1, 500), (2, 250), (3, 0)]
    # same trajectory from the oracle loop with the same host noise
    from oracle import unet_fp32 as O
    from oracle.ddpm import cfg_denoise_loop
    gen = torch.Generator().manual_seed(3)
    noises = [torch.randn(lat.shape, generator=gen) for _ in range(3)] + [None]
    fn = lambda x, t, c: O.unet_forward(sd, x, t, c, ocfg_small())
    ref = cfg_denoise_loop(fn, lat, pe.half().float(), ne.half().float(), noises, num_steps=4, guidance_scale=7.5)
    assert rel_l2(outs[0], ref) < 3e-2
    with pytest.raises(ValueError):
        pipe(prompt="a horse", height=64, width=64)            # no text encoder attached


def test_pipeline_without_guidance(small):
    """guidance_scale <= 1: do_classifier_free_guidance is False (pipeline_videogen.py:626) — the model batch is the
    latents alone (:666), eps is used as it is (:678) and no negative embeddings are needed; against the oracle UNet
    driven by the same loop arithmetic, and equal to guidance 1.0 + epsilon with identical text halves."""
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    from oracle import unet_fp32 as O
    from oracle.ddpm import DDPMSchedule
    net, sd = small
    pipe = VideoGenPipeline(unet=net)
    g = torch.Generator().manual_seed(19)
    pe = torch.randn(2, 77, 128, generator=g)                      # two prompts in one call
    lat = torch.randn(2, 4, 4, 8, 8, generator=g)
    out = pipe(prompt_embeds=pe, latents=lat, height=64, width=64, video_length=4, num_inference_steps=4,
               guidance_scale=1.0, generator=torch.Generator().manual_seed(5), output_type="latent").video.float().cpu()
    sch = DDPMSchedule()
    sch.set_timesteps(4)
    gen = torch.Generator().manual_seed(5)
    x = lat.clone()
    for t in sch.timesteps:
        eps = O.unet_forward(sd, x.half().float(), t, pe.half().float(), ocfg_small())
        x = sch.step(eps, t, x, torch.randn(lat.shape, generator=gen) if t > 0 else None)
    assert rel_l2(out, x) < 3e-2
    # a list of generators, one per latent (:499-504): each latent's noise comes from its own generator
    gl = lambda: [torch.Generator().manual_seed(31), torch.Generator().manual_seed(32)]
    o2 = pipe(prompt_embeds=pe, latents=lat, height=64, width=64, video_length=4, num_inference_steps=4, guidance_scale=1.0,
              generator=gl(), output_type="latent").video.float().cpu()
    o1 = pipe(prompt_embeds=pe[1:], latents=lat[1:], height=64, width=64, video_length=4, num_inference_steps=4,
              guidance_scale=1.0, generator=[torch.Generator().manual_seed(32)], output_type="latent").video.float().cpu()
    assert rel_l2(o2[1:], o1) < 5e-3                               # batch 2 vs batch 1: other tile choices, same noise
    with pytest.raises(ValueError):
        pipe(prompt_embeds=pe, latents=lat, height=64, width=64, video_length=4, num_inference_steps=4,
             guidance_scale=1.0, generator=[torch.Generator()], output_type="latent")


def test_prompts_batched_per_forward_match_single_prompt_runs(small):
    """bench.py --prompts-per-forward k ("batched B = 2k if memory-profitable"; the reference loops prompts one
    at a time, sample.py:78-91): k = 2 prompts with classifier-free guidance in ONE UNet forward per step (batch 4,
    [neg0 neg1 | pos0 pos1]) against the same two prompts denoised one at a time with the same per-prompt noise."""
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    net, _ = small
    pipe = VideoGenPipeline(unet=net)
    g = torch.Generator().manual_seed(77)
    pe, ne = torch.randn(2, 77, 128, generator=g), torch.randn(2, 77, 128, generator=g)
    lat = torch.randn(2, 4, 16, 8, 8, generator=g)
    kw = dict(height=64, width=64, video_length=16, num_inference_steps=6, guidance_scale=7.5, output_type="latent")
    both = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat,
                generator=[torch.Generator().manual_seed(41), torch.Generator().manual_seed(42)], **kw).video.float().cpu()
    for j in range(2):
        one = pipe(prompt_embeds=pe[j:j + 1], negative_prompt_embeds=ne[j:j + 1], latents=lat[j:j + 1],
                   generator=torch.Generator().manual_seed(41 + j), **kw).video.float().cpu()
        assert rel_l2(both[j:j + 1], one) < 5e-3, j
    assert rel_l2(both[0], both[1]) > 0.1                          # and they are two different videos


def test_pipeline_ddim_scheduler(small):
    """sample_method 'ddim' (base/pipelines/sample.py:44-49): the fused CFG + scheduler-step kernel driven by DDIM
    coefficients, eta = 0 (no noise drawn) and eta = 0.6 (host noise), against the oracle loop with oracle/ddim.py."""
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    from lavie_amd.scheduling_ddim import DDIMScheduler
    from oracle import unet_fp32 as O
    from oracle.ddim import DDIMSchedule
    net, sd = small
    pipe = VideoGenPipeline(unet=net, scheduler=DDIMScheduler())
    g = torch.Generator().manual_seed(19)
    pe, ne = torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
    lat = torch.randn(1, 4, 4, 8, 8, generator=g)
    ctx = torch.cat([ne, pe]).half().float()
    for eta in (0.0, 0.6):
        seen = []
        out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, height=64, width=64, video_length=4,
                   num_inference_steps=4, guidance_scale=7.5, eta=eta, generator=torch.Generator().manual_seed(5),
                   output_type="latent", callback=lambda i, t, x: seen.append(t)).video.float().cpu()
        assert seen == [751, 501, 251, 1]
        osch = DDIMSchedule()
        osch.set_timesteps(4)
        gen = torch.Generator().manual_seed(5)
        x = lat.clone()
        for t in osch.timesteps:
            eps = O.unet_forward(sd, torch.cat([x, x]), t, ctx, ocfg_small())
            guided = eps[0:1] + 7.5 * (eps[1:2] - eps[0:1])
            z = torch.randn(lat.shape, generator=gen) if eta > 0 else None
            x = osch.step(guided, t, x, eta=eta, noise=z)
        assert rel_l2(out, x) < 3e-2, eta


def test_pipeline_euler_scheduler(small):
    """sample_method 'eulerdiscrete' (base/pipelines/sample.py:50-55): fractional timesteps, init_noise_sigma-scaled
    latents, the sigma-scaled model input written by the fused step kernel — against the oracle loop (oracle/euler.py;
    PARITY UNPINNED against diffusers, see its header)."""
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    from lavie_amd.scheduling_euler_discrete import EulerDiscreteScheduler
    from oracle import unet_fp32 as O
    from oracle.euler import cfg_euler_loop
    net, sd = small
    pipe = VideoGenPipeline(unet=net, scheduler=EulerDiscreteScheduler())
    g = torch.Generator().manual_seed(23)
    pe, ne = torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
    lat = torch.randn(1, 4, 4, 8, 8, generator=g)
    seen = []
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, height=64, width=64, video_length=4,
               num_inference_steps=4, guidance_scale=7.5, output_type="latent",
               callback=lambda i, t, x: seen.append(t)).video.float().cpu()
    assert seen == pytest.approx([999.0, 666.0, 333.0, 0.0])
    fn = lambda x, t, c: O.unet_forward(sd, x, t, c, ocfg_small())
    ref = cfg_euler_loop(fn, lat, pe.half().float(), ne.half().float(), num_steps=4, guidance_scale=7.5)
    assert rel_l2(out, ref) < 3e-2


def test_pipeline_prompt_path_with_stock_text_encoder(small):
    """SURVEY §8 f4 (first half): a stock transformers CLIPTextModel on the device + a tokenizer object attached to the
    pipeline; `prompt=` / `negative_prompt=` go through `_encode_prompt` (pipeline_videogen.py:273-420: negative half first)
    and give the same latents as passing the encoder's outputs as prompt_embeds."""
    pytest.importorskip("transformers")
    from transformers import CLIPTextConfig, CLIPTextModel
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    net, _ = small
    torch.manual_seed(0)
    enc = CLIPTextModel(CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                       num_attention_heads=4, max_position_embeddings=77, bos_token_id=0, eos_token_id=2)).cuda().eval()

    class Tok:                       # stand-in for CLIPTokenizer (its vocabulary files are not in the image)
        model_max_length = 77

        def __call__(self, text, padding=None, max_length=77, truncation=True, return_tensors="pt"):
            text = [text] if isinstance(text, str) else text
            ids = torch.full((len(text), max_length), 2, dtype=torch.long)
            for i, s in enumerate(text):
                toks = [1 + (sum(map(ord, w)) % 900) for w in s.split()][: max_length - 1]
                ids[i, : len(toks)] = torch.tensor(toks, dtype=torch.long)
            from types import SimpleNamespace
            return SimpleNamespace(input_ids=ids)

    pipe = VideoGenPipeline(unet=net, text_encoder=enc, tokenizer=Tok())
    lat = torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(1))
    kw = dict(latents=lat, height=64, width=64, video_length=4, num_inference_steps=2, guidance_scale=7.5, output_type="latent")
    a = pipe(prompt="a corgi walking in the park", negative_prompt="blurry", generator=torch.Generator().manual_seed(3), **kw).video
    with torch.no_grad():
        pe = enc(Tok()("a corgi walking in the park").input_ids.cuda())[0]
        ne = enc(Tok()("blurry").input_ids.cuda())[0]
    b = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, generator=torch.Generator().manual_seed(3), **kw).video
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_full_width_two_prompts_per_forward_through_the_fused_kernels(full):
    """Two prompts' CFG pairs in one UNet batch of 4 ([neg0 neg1 | pos0 pos1]) at the base width, where the level-0 blocks run the
    row-resident fused kernels (the text cross-attention kernel streams one K / V image per batch entry and cuts its passes at
    the video boundaries): every entry must equal the same entry computed in a batch of 2, with the context cached and the shared
    CFG prefix on, as the guided loop runs it."""
    from lavie_amd import _lib
    import bench
    net, _ = full
    lib = _lib.load()
    g = torch.Generator().manual_seed(123)
    lat = torch.randn(2, 4, 16, 8, 8, generator=g).half().cuda()
    neg, pos = torch.randn(2, 77, 768, generator=g).half().cuda(), torch.randn(2, 77, 768, generator=g).half().cuda()

    def run(x, ctx):
        net.prepare(x.shape[0], 16, 8, 8, 77)
        try:
            net.set_cfg_shared_input(True)
            cc = net.cache_context(ctx.contiguous())
            bench.profile_begin(lib, 1 << 10, 64)
            y = net(x.contiguous(), 321, encoder_hidden_states=cc).sample.clone()
            rows = bench.profile_end(lib)
            assert rows[10]["launches"] == 5, rows[10]
            return y
        finally:
            net.cache_context(None)
            net.set_cfg_shared_input(False)
    both = run(torch.cat([lat, lat]), torch.cat([neg, pos]))
    for j in range(2):
        one = run(torch.cat([lat[j:j + 1]] * 2), torch.cat([neg[j:j + 1], pos[j:j + 1]]))
        assert rel_l2(both[j], one[0]) < 2e-3 and rel_l2(both[2 + j], one[1]) < 2e-3, j
    assert rel_l2(both[0], both[1]) > 1e-2 and rel_l2(both[0], both[2]) > 1e-2


def test_cfg_shared_input_matches_plain_forward(small):
    """lavie_unet_set_cfg_shared_input: with the two halves of the batch holding the same latents (classifier-free guidance,
    pipeline_videogen.py:666) the layers in front of the first text cross-attention run once; the output must equal the plain
    forward's to rounding, for batch 2 and batch 4 ([neg0 neg1 | pos0 pos1]), cached context or not, and the switch must
    switch off."""
    net, _ = small
    g = torch.Generator().manual_seed(5)
    for nb in (2, 4):
        lat = torch.randn(nb // 2, 4, 4, 8, 8, generator=g).half().cuda()
        x = torch.cat([lat, lat]).contiguous()
        ctx = torch.randn(nb, 77, 128, generator=g).half().cuda()
        plain = net(x, 400, encoder_hidden_states=ctx).sample.clone()
        try:
            net.set_cfg_shared_input(True)
            shared = net(x, 400, encoder_hidden_states=ctx).sample.clone()
            cc = net.cache_context(ctx)
            shared_c = net(x, 400, encoder_hidden_states=cc).sample.clone()
            net.cache_context(None)
        finally:
            net.set_cfg_shared_input(False)
        assert rel_l2(shared, plain) < 2e-3 and rel_l2(shared_c, plain) < 2e-3, (nb, rel_l2(shared, plain))
        assert rel_l2(plain[:nb // 2], plain[nb // 2:]) > 1e-2          # the halves do differ (different text)
        assert torch.equal(net(x, 400, encoder_hidden_states=ctx).sample, plain)      # off again: the plain path, bit for bit


def test_pipeline_cfg_shared_prefix_matches_reference_order(small):
    """VideoGenPipeline.cfg_shared_prefix (default on) against the same loop with both halves computed."""
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    net, _ = small
    pipe = VideoGenPipeline(unet=net)
    g = torch.Generator().manual_seed(9)
    pe, ne = torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
    lat = torch.randn(1, 4, 16, 8, 8, generator=g)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, height=64, width=64, video_length=16, num_inference_steps=6,
              guidance_scale=7.5, output_type="latent")
    a = pipe(generator=torch.Generator().manual_seed(3), **kw).video.float().cpu()
    try:
        pipe.cfg_shared_prefix = False
        b = pipe(generator=torch.Generator().manual_seed(3), **kw).video.float().cpu()
    finally:
        del pipe.cfg_shared_prefix
    assert rel_l2(a, b) < 5e-3


def test_context_cache_is_bit_identical_and_scoped(small):
    """lavie_unet_cache_context: text keys / values computed once per context give bit-identical outputs; another
    context tensor (other pointer) is computed as usual; dropping the cache restores the per-call path."""
    net, _ = small
    g = torch.Generator().manual_seed(41)
    x = torch.randn(2, 4, 4, 8, 8, generator=g).half().cuda()
    c1 = torch.randn(2, 77, 128, generator=g).half().cuda()
    c2 = torch.randn(2, 77, 128, generator=g).half().cuda()
    ref1 = net(x, 300, encoder_hidden_states=c1).sample.clone()
    ref2 = net(x, 300, encoder_hidden_states=c2).sample.clone()
    assert not torch.equal(ref1, ref2)
    cc = net.cache_context(c1)
    assert cc.data_ptr() == c1.data_ptr()
    assert torch.equal(net(x, 300, encoder_hidden_states=cc).sample, ref1)
    assert torch.equal(net(x, 300, encoder_hidden_states=c2).sample, ref2)          # other tensor: not served from the cache
    assert torch.equal(net(x, 300, encoder_hidden_states=cc).sample, ref1)
    net.cache_context(None)
    assert torch.equal(net(x, 300, encoder_hidden_states=c1).sample, ref1)
    # a batch-1 call with a view of the cached tensor's first row has the same pointer but another shape: recomputed
    net.cache_context(c1)
    y1 = net(x[:1], 300, encoder_hidden_states=c1[:1]).sample
    net.cache_context(None)
    assert torch.equal(y1, net(x[:1], 300, encoder_hidden_states=c1[:1]).sample)


def test_graph_replay_is_bit_identical(small):
    """lavie_unet_forward_graph: eager first call, capture on the second, replay afterwards — bit-identical to the eager
    forward while the tensor contents (latents, timestep) change under fixed addresses; a new address or a changed text
    cache starts over instead of replaying a stale graph."""
    net, _ = small
    g = torch.Generator().manual_seed(43)
    xs = [torch.randn(2, 4, 4, 8, 8, generator=g).half().cuda() for _ in range(5)]
    ctx = torch.randn(2, 77, 128, generator=g).half().cuda()
    ts = [900, 700, 500, 300, 100]
    ref = [net(x, t, encoder_hidden_states=ctx).sample.clone() for x, t in zip(xs, ts)]
    net.enable_graph(True)
    try:
        buf = torch.empty_like(xs[0])
        for i, (x, t) in enumerate(zip(xs, ts)):          # call 0 eager, 1 captures, 2.. replay
            buf.copy_(x)
            assert torch.equal(net(buf, t, encoder_hidden_states=ctx).sample, ref[i]), f"step {i}"
        # another input address: not replayed against the old one
        assert torch.equal(net(xs[2], ts[2], encoder_hidden_states=ctx).sample, ref[2])
        assert torch.equal(net(xs[2], ts[2], encoder_hidden_states=ctx).sample, ref[2])
        # the text cache changes what a forward enqueues: the graph is dropped, results stay the same
        cc = net.cache_context(ctx)
        for i in (0, 1, 3):
            buf.copy_(xs[i])
            assert torch.equal(net(buf, ts[i], encoder_hidden_states=cc).sample, ref[i])
        net.cache_context(None)
        buf.copy_(xs[4])
        assert torch.equal(net(buf, ts[4], encoder_hidden_states=ctx).sample, ref[4])
    finally:
        net.enable_graph(False)


def test_pipeline_with_graph_matches_eager(small):
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    net, _ = small
    g = torch.Generator().manual_seed(44)
    emb = torch.randn(1, 77, 128, generator=g)
    neg = torch.randn(1, 77, 128, generator=g)
    lat = torch.randn(1, 4, 4, 8, 8, generator=g)

    def run():
        pipe = VideoGenPipeline(unet=net)
        return pipe(prompt_embeds=emb, negative_prompt_embeds=neg, latents=lat.clone(), num_inference_steps=6,
                    guidance_scale=7.5, generator=torch.Generator().manual_seed(5), output_type="latent",
                    video_length=4, height=64, width=64).video.clone()

    eager = run()
    net.enable_graph(True)
    try:
        graphed = run()
    finally:
        net.enable_graph(False)
    assert torch.equal(eager, graphed)


def test_workspace_plan_covers_every_switch_combination(full):
    """prepare() sizes the workspace from dry runs, the forward may run under other switches (the guided loop turns the
    shared CFG prefix on AFTER prepare; lavie_debug_fused_mask and the text cache change which GEMMs, split-K slabs and statistics
    buffers exist).  The plan is the maximum over the switch combinations: after ONE prepare at the production shape, every mask /
    shared-prefix / cached-context combination must run (no 'workspace exhausted') and agree with the default to the engine's usual
    kernel-switch tolerance."""
    import bench
    from lavie_amd import _lib
    lib = _lib.load()
    DEF = _lib.FUSED_DEFAULT
    net, _ = full
    pe, ne, lat = bench.synth_inputs(0, "cpu")
    ctx = torch.cat([ne, pe]).half().cuda()
    x = torch.cat([lat, lat]).half().cuda()
    net.cache_context(None)
    net.set_cfg_shared_input(False)
    _lib.check(lib.lavie_debug_fused_mask(DEF), "lavie_debug_fused_mask")
    net._prepared = None
    net.prepare(2, bench.FRAMES, bench.LAT_H, bench.LAT_W, bench.CTX_LEN)             # once, under the default switches
    ref = net(x, 500, encoder_hidden_states=ctx).sample.clone()
    try:
        for mask in (0, 0x08, 0x10, 0x20, 0x38, DEF | 0x08, DEF & ~0x107, DEF & ~0x20, DEF & ~0x10, 0x1FF & ~0xC0):
            for shared in (False, True):
                for cached in (False, True):
                    _lib.check(lib.lavie_debug_fused_mask(mask), "lavie_debug_fused_mask")
                    cc = net.cache_context(ctx) if cached else ctx
                    if not cached:
                        net.cache_context(None)
                    net.set_cfg_shared_input(shared)
                    got = net(x, 500, encoder_hidden_states=cc).sample
                    assert rel_l2(got, ref) < 3e-3, (hex(mask), shared, cached, rel_l2(got, ref))
    finally:
        lib.lavie_debug_fused_mask(DEF)
        net.set_cfg_shared_input(False)
        net.cache_context(None)
