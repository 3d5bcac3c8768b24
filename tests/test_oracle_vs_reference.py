"""Pins oracle/ against the reference's own modules (build container only)."""
import pytest
import torch

import refbuild
import refimport
from lavie_amd import spec, weights
from lavie_amd.config import UNetConfig
from oracle import unet_fp32 as O

pytestmark = pytest.mark.reference

SMALL = UNetConfig(block_out_channels=(256, 512, 512, 512), cross_attention_dim=128)


def rel_l2(a, b):
    return ((a - b).norm() / b.norm()).item()


def ocfg(cfg):
    return O.UNetConfig(in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                        block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
                        heads=cfg.heads, cross_attention_dim=cfg.cross_attention_dim)


@pytest.fixture(scope="module")
def small():
    torch.manual_seed(0)
    net, sd = refbuild.reference_unet(SMALL, seed=3)
    return net, sd


def test_state_dict_keys_match_reference(small):
    net, _ = small
    ref = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert ref == spec.param_shapes(SMALL) == O.param_shapes(ocfg(SMALL))


def test_base_spec_counts():
    shapes = spec.param_shapes()
    assert len(shapes) == 830                                   # SURVEY §3.2 [probe]
    assert spec.param_count() == 909_124_116 + 15 * 16          # shared rotary freqs listed 16x in the state dict


@pytest.mark.parametrize("t", [980, 500, 0])
def test_whole_unet_small(small, t):
    net, sd = small
    g = torch.Generator().manual_seed(10 + t)
    x = torch.randn(2, 4, 16, 8, 8, generator=g)
    ctx = torch.randn(2, 77, SMALL.cross_attention_dim, generator=g)
    with torch.no_grad():
        ref = net(x, torch.tensor(t), encoder_hidden_states=ctx).sample
        got = O.unet_forward(sd, x, t, ctx, ocfg(SMALL))
    assert rel_l2(got, ref) < 1e-4


def test_resnet_block_direct_import():
    """T1: resnet.py imports standalone; 5-D GroupNorm domain and shortcut."""
    m = refimport.load()
    for cin, cout in ((64, 128), (64, 64)):
        torch.manual_seed(cin + cout)
        blk = m.resnet.ResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=256, groups=32, eps=1e-5).eval()
        sd = {"r." + k: v for k, v in blk.state_dict().items()}
        x = torch.randn(2, cin, 4, 8, 8) * torch.linspace(0.5, 2.0, 4).reshape(1, 1, 4, 1, 1)
        temb = torch.randn(2, 256)
        with torch.no_grad():
            ref = blk(x, temb)
            got = O.resnet_block(sd, "r.", x, temb, O.UNetConfig())
        assert rel_l2(got, ref) < 1e-5


def test_up_down_sample_direct_import():
    m = refimport.load()
    torch.manual_seed(1)
    up = m.resnet.Upsample3D(64, use_conv=True, out_channels=64).eval()
    dn = m.resnet.Downsample3D(64, use_conv=True, out_channels=64, padding=1, name="op").eval()
    x = torch.randn(1, 64, 4, 8, 8)
    with torch.no_grad():
        assert rel_l2(O.upsample({"u." + k: v for k, v in up.state_dict().items()}, "u.", x), up(x)) < 1e-5
        assert rel_l2(O.downsample({"d." + k: v for k, v in dn.state_dict().items()}, "d.", x), dn(x)) < 1e-5


@pytest.mark.parametrize("c,frames", [(320, 16), (640, 16), (1280, 16), (320, 61)])
def test_temporal_attention(c, frames):
    m = refimport.load()
    from rotary_embedding_torch import RotaryEmbedding
    torch.manual_seed(c + frames)
    att = m.attention.TemporalAttention(query_dim=c, heads=8, dim_head=c // 8, rotary_emb=RotaryEmbedding(32)).eval()
    torch.nn.init.normal_(att.to_out[0].weight, std=0.02)
    sd = {"a." + k: v for k, v in att.state_dict().items()}
    x = torch.randn(64, frames, c)
    with torch.no_grad():
        ref = att(x)
        got = O.temporal_attention(sd, "a.", x, O.UNetConfig())
    assert rel_l2(got, ref) < 1e-5


@pytest.mark.parametrize("n", [16, 61])
def test_rel_pos_buckets(n):
    m = refimport.load()
    q = torch.arange(n)
    rel = q.reshape(1, n) - q.reshape(n, 1)
    ref = m.attention.RelativePositionBias._relative_position_bucket(rel, num_buckets=32, max_distance=32)
    assert torch.equal(O.rel_pos_bucket_table(n, 32, 32), ref)
    if n == 16:   # SURVEY §8 a15 row q=0 / col k=0
        assert ref[0].tolist() == [0, 17, 18, 19, 20, 21, 22, 23, 24, 24, 25, 25, 26, 26, 27, 27]
        assert ref[:, 0].tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 8, 9, 9, 10, 10, 11, 11]


def test_transformer3d_block_order():
    m = refimport.load()
    from rotary_embedding_torch import RotaryEmbedding
    torch.manual_seed(5)
    tr = m.attention.Transformer3DModel(8, 40, in_channels=320, num_layers=1, cross_attention_dim=768,
                                        norm_num_groups=32, rotary_emb=RotaryEmbedding(32)).eval()
    torch.nn.init.normal_(tr.transformer_blocks[0].attn_temp.to_out[0].weight, std=0.02)
    sd = {"t." + k: v for k, v in tr.state_dict().items()}
    x = torch.randn(2, 320, 16, 8, 8) * torch.linspace(0.5, 2.0, 16).reshape(1, 1, 16, 1, 1)
    ctx = torch.randn(2, 77, 768)
    with torch.no_grad():
        ref = tr(x, encoder_hidden_states=ctx, use_image_num=0).sample
        got = O.transformer3d(sd, "t.", x, ctx, O.UNetConfig())
    assert rel_l2(got, ref) < 1e-5


def test_ddim_oracle_matches_vendored_reference_scheduler():
    """oracle/ddim.py vs the reference's own DDIMScheduler text (vsr/diffusion/scheduling_ddim.py), every 7th step."""
    from oracle.ddim import DDIMSchedule
    ref = refimport.load_vsr_ddim()(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, beta_schedule="linear",
                                    clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    ref.set_timesteps(50)
    sch = DDIMSchedule(timestep_spacing="vsr_linspace")
    sch.set_timesteps(50)
    assert sch.timesteps == [int(t) for t in ref.timesteps]
    g = torch.Generator().manual_seed(77)
    for t in sch.timesteps[1::7]:        # [0] is 1001 with this spacing: past the alpha table in the reference as well
        x, eps, z = (torch.randn(2, 4, 3, 8, 8, generator=g) for _ in range(3))
        for eta in (0.0, 1.0):
            want = ref.step(eps, t, x, eta=eta, variance_noise=z if eta > 0 else None).prev_sample
            assert rel_l2(sch.step(eps, t, x, eta=eta, noise=z), want) < 1e-6
    for kind in ("v_prediction", "sample"):          # the other branches of the vendored step (:356-363)
        pref = refimport.load_vsr_ddim()(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, beta_schedule="linear",
                                         clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type=kind)
        pref.set_timesteps(50)
        psch = DDIMSchedule(timestep_spacing="vsr_linspace", prediction_type=kind)
        psch.set_timesteps(50)
        for t in psch.timesteps[1::7]:
            x, m, z = (torch.randn(2, 4, 3, 8, 8, generator=g) for _ in range(3))
            for eta in (0.0, 1.0):
                want = pref.step(m, t, x, eta=eta, variance_noise=z if eta > 0 else None).prev_sample
                assert rel_l2(psch.step(m, t, x, eta=eta, noise=z), want) < 1e-6, (kind, t, eta)


# ------------------------------------------------------------------ frame-interpolation model
def test_interpolation_unet_small_matches_reference():
    """oracle (sparse-causal attn1, FF before temporal, plain temporal attention, 8 input channels) against the imported
    interpolation/models UNet at a small width, ragged frame count; also pins the state-dict contract."""
    m = refimport.load("interpolation")
    cfg = O.UNetConfig(in_channels=8, block_out_channels=(64, 128), attn_levels=(True, False), layers_per_block=1,
                       cross_attention_dim=64, sparse_causal_attn1=True, temporal_plain=True, ff_before_temporal=True)
    net = m.unet.UNet3DConditionModel(sample_size=8, in_channels=8, out_channels=4, block_out_channels=(64, 128),
                                      down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
                                      up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"), layers_per_block=1,
                                      cross_attention_dim=64, attention_head_dim=8, use_first_frame=True).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert shapes == O.param_shapes(cfg)
    sd = weights.synth_state_dict(shapes, 8)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    x, ctx = torch.randn(2, 8, 7, 8, 8, generator=g), torch.randn(2, 77, 64, generator=g)
    with torch.no_grad():
        ref = net(x, torch.tensor(500), encoder_hidden_states=ctx).sample
    assert rel_l2(O.unet_forward(sd, x, 500, ctx, cfg), ref) < 1e-5


def test_interpolation_ddim_loop_matches_reference_diffusion():
    """oracle/interp_ddim.py and the product's host-side SpacedDiffusion (generic loop) against the reference's
    interpolation/diffusion package, driven as interpolation/sample.py:160-168 does, with a small linear denoiser."""
    import numpy as np
    from lavie_amd.interpolation import create_diffusion
    from oracle import interp_ddim as D
    rd = refimport.load_interp_diffusion()
    for n in ("50", "25", "7"):
        r, s, p = rd.create_diffusion(n), D.SpacedSchedule(n), create_diffusion(n)
        assert r.timestep_map == s.timestep_map == p.timestep_map
        assert np.array_equal(r.alphas_cumprod, s.alphas_cumprod) and np.array_equal(r.alphas_cumprod, p.alphas_cumprod)
    g = torch.Generator().manual_seed(0)
    W = torch.randn(4, 8, generator=g) * 0.05
    toy = lambda x, t, c: (torch.einsum("oc,bcfhw->bofhw", W, x) * (1 + t.float().reshape(-1, 1, 1, 1, 1) / 1000)
                           + 0.1 * c.mean(dim=(1, 2)).reshape(-1, 1, 1, 1, 1))

    def fwd_cfg(x, t, encoder_hidden_states=None, class_labels=None, cfg_scale=4.0):
        return D.forward_with_cfg(toy, x, t, encoder_hidden_states, cfg_scale)

    z = torch.cat([torch.randn(1, 4, 3, 4, 4, generator=g)] * 2)
    xs = torch.cat([torch.randn(1, 4, 3, 4, 4, generator=g)] * 2)
    ctx = torch.randn(2, 7, 5, generator=g)
    kw = dict(clip_denoised=False, model_kwargs=dict(encoder_hidden_states=ctx, class_labels=None), mask=None, x_start=xs,
              use_concat=True, copy_no_mask=True)
    ref = rd.create_diffusion("10").ddim_sample_loop(fwd_cfg, z.shape, z, progress=False, device="cpu", **kw)
    assert rel_l2(D.ddim_sample_loop(toy, z, xs, ctx, D.SpacedSchedule("10"), 4.0), ref) < 1e-6
    assert rel_l2(create_diffusion("10").ddim_sample_loop(fwd_cfg, z.shape, z, **kw), ref) < 1e-6


def test_vsr_resnet_block_3dcnn_direct_import():
    """vsr/models/resnet.py imports only torch / einops: the reference's own ResnetBlock3DCNN (tier T1)."""
    import importlib.util
    from oracle import vsr_blocks as V
    spec = importlib.util.spec_from_file_location("ref_vsr_resnet", "/root/reference/vsr/models/resnet.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    g = torch.Generator().manual_seed(4)
    for c, kern, frames in ((64, (5, 1, 1), 7), (128, (3, 1, 1), 2)):
        blk = m.ResnetBlock3DCNN(in_channels=c, out_channels=c, kernel=kern, temb_channels=96).eval()
        sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in blk.state_dict().items()}, 5)
        blk.load_state_dict(sd)
        x, temb = torch.randn(2, c, frames, 4, 6, generator=g), torch.randn(2, 96, generator=g)
        with torch.no_grad():
            ref = blk(x, temb)
        assert rel_l2(V.resnet_block_3dcnn(sd, "", x, temb), ref) < 1e-5


@pytest.mark.parametrize("only_cross", [True, False])
def test_vsr_transformer3d_matches_reference(only_cross):
    """oracle/vsr_blocks.py against vsr/models/attention.py's Transformer3DModel (under the shim)."""
    from oracle import vsr_blocks as V
    m = refimport.load_vsr_blocks()                      # puts tests/refshim on sys.path
    from rotary_embedding_torch import RotaryEmbedding
    tr = m.attention.Transformer3DModel(8, 32, in_channels=256, num_layers=1, cross_attention_dim=128, norm_num_groups=32,
                                        use_linear_projection=True, only_cross_attention=only_cross,
                                        rotary_emb=RotaryEmbedding(32)).eval()
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in tr.state_dict().items()}, 7)
    tr.load_state_dict(sd)
    g = torch.Generator().manual_seed(2)
    x, ctx = torch.randn(2, 256, 5, 4, 4, generator=g), torch.randn(2, 77, 128, generator=g)
    with torch.no_grad():
        ref = tr(x, encoder_hidden_states=ctx).sample
    assert rel_l2(V.vsr_transformer3d(sd, "", x, ctx, 8, only_cross), ref) < 1e-5


def test_vsr_unet_small_matches_reference():
    """oracle.vsr_blocks.vsr_unet_forward against the imported vsr/models UNet3DVSRModel (temporal modules after every
    block, class-embedded noise level, 4 + 3 input channels, only_cross_attention levels) at a small width."""
    from oracle import vsr_blocks as V
    m = refimport.load_vsr_blocks()
    net = m.unet.UNet3DVSRModel(
        sample_size=8, in_channels=7, out_channels=4, block_out_channels=(64, 128),
        down_block_types=("DownBlock3D", "CrossAttnDownBlock3D"), up_block_types=("CrossAttnUpBlock3D", "UpBlock3D"),
        only_cross_attention=(True, False), layers_per_block=1, cross_attention_dim=64, attention_head_dim=4,
        use_linear_projection=True, num_class_embeds=1000, down_temporal_idx=(0, 1), mid_temporal=True,
        up_temporal_idx=(0, 1), video_condition=False, temporal_module_config=refimport.VSR_TEMPORAL_MODULE_CONFIG).eval()
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 9)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(1)
    x, lr = torch.randn(2, 4, 5, 8, 8, generator=g), torch.randn(2, 3, 5, 8, 8, generator=g)
    ctx, labels = torch.randn(2, 77, 64, generator=g), torch.tensor([20, 250])
    with torch.no_grad():
        ref = net(x, torch.tensor(500), lr, encoder_hidden_states=ctx, class_labels=labels).sample
    got = V.vsr_unet_forward(sd, x, lr, 500, ctx, labels, block_out_channels=(64, 128), attn_levels=(False, True),
                             only_cross_attention=(True, False), layers_per_block=1, heads=4)
    assert rel_l2(got, ref) < 1e-5


# ------------------------------------------------------------------ a5 / a17: pins against reference-HELD text
@pytest.mark.parametrize("dim", [320, 640])
def test_geglu_ff_matches_reference_held_feedforward(dim):
    """a17: oracle.geglu_ff against the FeedForward / GEGLU classes the reference itself holds
    (vsr/models/diffusers_attention.py:734-822, diffusers' text vendored by the reference; the class the base model takes
    from the absent diffusers package at attention.py:17,479 under the same name and state-dict keys)."""
    m = refimport.load_vsr_blocks()
    ff = m.diffusers_attention.FeedForward(dim, dropout=0.0, activation_fn="geglu").eval()
    assert sorted(ff.state_dict()) == ["net.0.proj.bias", "net.0.proj.weight", "net.2.bias", "net.2.weight"]
    assert tuple(ff.net[0].proj.weight.shape) == (8 * dim, dim) and tuple(ff.net[2].weight.shape) == (dim, 4 * dim)
    sd = weights.synth_state_dict({k: tuple(v.shape) for k, v in ff.state_dict().items()}, 21)
    ff.load_state_dict(sd)
    x = torch.randn(3, 50, dim, generator=torch.Generator().manual_seed(dim)) * 2.0
    with torch.no_grad():
        ref = ff(x)
    assert rel_l2(O.geglu_ff(sd, "", x), ref) < 1e-6


def test_timestep_sinusoid_matches_reference_held_embedding():
    """a5: the oracle's Timesteps(320, flip_sin_to_cos=True, freq_shift=0) against the sinusoidal embedding the reference
    holds in-tree (base/models/utils.py:74-94: [cos | sin], w_k = max_period^(-k/half)), every DDPM timestep the 50-step
    schedule visits plus fractional ones (Euler)."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("ref_base_utils", "/root/reference/base/models/utils.py")
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    t = torch.cat([torch.arange(980, -1, -20, dtype=torch.float32), torch.tensor([999.0, 0.5, 123.456, 1.0])])
    for dim in (320, 256):
        ref = mod.timestep_embedding(t, dim)
        got = O.timestep_sinusoid(t, dim)
        assert got.shape == ref.shape == (t.numel(), dim)
        assert torch.equal(got, ref) or (got - ref).abs().max().item() < 1e-6
