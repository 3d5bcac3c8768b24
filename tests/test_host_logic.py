"""CPU: host-side logic — parameter spec, weight recipe, scheduler arithmetic, library symbols, layout rules."""
import ctypes
import math
import os
import re

import pytest
import torch

from lavie_amd import _lib, spec, weights
from lavie_amd.config import BASE_CONFIG, UNetConfig
from lavie_amd.scheduling_ddpm import DDPMScheduler
from oracle import unet_fp32 as O
from oracle.ddpm import DDPMSchedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spec_matches_oracle_inventory():
    assert spec.param_shapes() == O.param_shapes()
    assert len(spec.param_shapes()) == 830
    small = UNetConfig(block_out_channels=(256, 512), attn_levels=(True, False), cross_attention_dim=128)
    assert spec.param_shapes(small) == O.param_shapes(O.UNetConfig(block_out_channels=(256, 512), attn_levels=(True, False),
                                                                   cross_attention_dim=128))


def test_weight_recipe_is_order_and_subset_independent():
    cfg = UNetConfig(block_out_channels=(64, 64), attn_levels=(False, False), cross_attention_dim=64)
    shapes = spec.param_shapes(cfg)
    full = weights.synth_state_dict(shapes, seed=5)
    part = weights.synth_state_dict(shapes, seed=5, only_prefix="mid_block.")
    assert part and all(torch.equal(full[k], v) for k, v in part.items())
    again = weights.synth_state_dict(dict(reversed(list(shapes.items()))), seed=5)
    assert all(torch.equal(full[k], again[k]) for k in full)
    assert not torch.equal(full["conv_in.weight"], weights.synth_state_dict(shapes, seed=6)["conv_in.weight"])


def test_temporal_out_projection_is_not_zero_in_synth_weights():
    shapes = {k: v for k, v in spec.param_shapes().items() if k.endswith("attn_temp.to_out.0.weight")}
    sd = weights.synth_state_dict(shapes, 0)
    assert all(v.abs().max() > 0 for v in sd.values())


def test_config_validation():
    BASE_CONFIG.validate()
    with pytest.raises(ValueError):
        UNetConfig(block_out_channels=(100, 200), attn_levels=(True, False)).validate()
    with pytest.raises(ValueError):
        UNetConfig(block_out_channels=(128, 128), attn_levels=(True, False)).validate()      # head dim 16 < rotary 32


def test_scheduler_matches_oracle_and_timesteps():
    sch, osch = DDPMScheduler(), DDPMSchedule()
    sch.set_timesteps(50)
    osch.set_timesteps(50)
    assert [int(t) for t in sch.timesteps] == osch.timesteps == list(range(980, -1, -20))
    for t in osch.timesteps:
        assert sch.coefficients(t) == pytest.approx(osch.coefficients(t), rel=1e-7)
    g1, g2 = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    x, eps = torch.randn(1, 4, 2, 4, 4), torch.randn(1, 4, 2, 4, 4)
    got = sch.step(eps, 500, x, generator=g1).prev_sample
    ref = osch.step(eps, 500, x, torch.randn(x.shape, generator=g2))
    assert torch.allclose(got, ref, atol=1e-6)
    assert torch.equal(sch.step(eps, 0, x).prev_sample, osch.step(eps, 0, x, None))          # no noise at t = 0


def test_ddpm_posterior_matches_independent_closed_form():
    """Cross-check against the textbook q(x_{t-1} | x_t, x_0) posterior for a respaced schedule — the form the
    reference's in-tree OpenAI sampler uses (interpolation/diffusion/gaussian_diffusion.py:232-252, 362-394):
    coef1 = beta_t sqrt(abar_prev) / (1 - abar_t), coef2 = (1 - abar_prev) sqrt(alpha_t) / (1 - abar_t)."""
    sch = DDPMScheduler()
    sch.set_timesteps(50)
    ab = sch.alphas_cumprod.double()
    for t in (980, 500, 20):
        a_t, a_p = ab[t].item(), ab[t - 20].item()
        beta = 1 - a_t / a_p
        coef1 = beta * math.sqrt(a_p) / (1 - a_t)
        coef2 = (1 - a_p) * math.sqrt(1 - beta) / (1 - a_t)
        var = beta * (1 - a_p) / (1 - a_t)
        k_x, k_e, c_x0, c_xt, sigma = sch.coefficients(t)
        assert (c_x0, c_xt, sigma) == pytest.approx((coef1, coef2, math.sqrt(var)), rel=1e-5)
        assert (k_x, k_e) == pytest.approx((1 / math.sqrt(a_t), math.sqrt(1 / a_t - 1)), rel=1e-5)


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads without a GPU and exports exactly what include/lavie_hip.h declares."""
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "lavie_hip.h")).read()
    declared = set(re.findall(r"\b(lavie_[a-z0-9_]+)\s*\(", header))
    declared -= {"lavie_unet_s"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.lavie_abi_version() == _lib.ABI_VERSION == 7


def test_relpos_buckets_host_function():
    from lavie_amd import ops
    for f in (1, 9, 16, 61):
        assert torch.equal(ops.relpos_buckets(f), O.rel_pos_bucket_table(f, 32, 32))
    buf = (ctypes.c_int * 4)()
    assert _lib.load().lavie_relpos_buckets(0, 32, 32, buf) != 0
    assert b"relpos_buckets" in _lib.load().lavie_last_error()


def test_engine_param_inventory_matches_spec():
    """lavie_unet_create needs no GPU: the engine's own name/numel list equals the Python spec."""
    from lavie_amd.unet import UNet3DConditionModel
    lib = _lib.load()
    net = UNet3DConditionModel(sample_size=8, block_out_channels=(256, 512), cross_attention_dim=128,
                               down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
                               up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"), init_weights=False)
    handle = ctypes.c_void_p()
    cfg = net._config_c()
    assert lib.lavie_unet_create(ctypes.byref(cfg), ctypes.byref(handle)) == 0
    try:
        n = lib.lavie_unet_num_params(handle)
        got = {}
        for i in range(n):
            name, numel = ctypes.c_char_p(), ctypes.c_longlong()
            assert lib.lavie_unet_param_info(handle, i, ctypes.byref(name), ctypes.byref(numel)) == 0
            got[name.value.decode()] = numel.value
        want = {k: math.prod(v) for k, v in spec.param_shapes(net.cfg).items()}
        assert got == want
        assert list(got) == [k for k, _ in spec.iter_params(net.cfg)]          # same enumeration order
        assert lib.lavie_unet_set_param(handle, b"no.such.key", ctypes.c_void_p(16), 1) != 0
    finally:
        lib.lavie_unet_destroy(handle)


def test_facade_surface_and_errors():
    from lavie_amd.unet import UNet3DConditionModel
    net = UNet3DConditionModel(sample_size=8, block_out_channels=(256, 512), cross_attention_dim=128,
                               down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
                               up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"), init_weights=False)
    assert net.config.in_channels == 4 and net.config.sample_size == 8
    assert set(net.state_dict()) == set(spec.param_shapes(net.cfg))
    with pytest.raises(RuntimeError, match="MI355X only"):       # no silent CPU path
        net(torch.zeros(2, 4, 4, 8, 8), 10, encoder_hidden_states=torch.zeros(2, 77, 128))
    with pytest.raises(NotImplementedError):
        UNet3DConditionModel(use_linear_projection=True, init_weights=False, block_out_channels=(256, 512),
                             down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
                             up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"))
    with pytest.raises(ValueError):
        UNet3DConditionModel(down_block_types=("Nope", "DownBlock3D"), up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"),
                             block_out_channels=(256, 512), init_weights=False)


def test_product_never_imports_oracle_or_reference():
    pat = re.compile(r"^\s*(from|import)\s+(oracle|refimport|refbuild|refshim)\b", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lavie_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f"{f} imports test infrastructure"
                assert "/root/reference" not in text or f.endswith((".py", ".hip", ".h", ".cpp")), f
    for f in ("bench.py",):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            assert "refimport" not in open(p).read()


def test_ddim_scheduler_matches_oracle_and_fused_form():
    """lavie_amd.scheduling_ddim: step == oracle step, and the five fused-kernel coefficients reproduce it."""
    from lavie_amd.scheduling_ddim import DDIMScheduler
    from oracle.ddim import DDIMSchedule
    sch, osch = DDIMScheduler(), DDIMSchedule()
    sch.set_timesteps(50)
    osch.set_timesteps(50)
    assert [int(t) for t in sch.timesteps] == osch.timesteps
    assert sch.init_noise_sigma == 1.0 and sch.order == 1
    g = torch.Generator().manual_seed(3)
    for t in (981, 501, 21, 1):
        for eta in (0.0, 0.7):
            x, eps, z = (torch.randn(1, 4, 2, 8, 8, generator=g) for _ in range(3))
            want = osch.step(eps, t, x, eta=eta, noise=z)
            got = sch.step(eps, t, x, eta=eta, variance_noise=z if eta > 0 else None).prev_sample
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)
            k_x, k_e, c0, ct, sigma = sch.coefficients(t, eta)
            x0 = k_x * x - k_e * eps
            fused = c0 * x0 + ct * x + sigma * z
            assert torch.allclose(fused, want, rtol=1e-4, atol=1e-5), (t, eta)
    # the other prediction types (vendored step :356-363) in step() and in the fused five-coefficient form, and the
    # frozen outputs of the vendored class itself (tests/golden/ddim_steps.pt)
    for kind in ("v_prediction", "sample"):
        psch, posch = DDIMScheduler(prediction_type=kind), DDIMSchedule(prediction_type=kind)
        psch.set_timesteps(50)
        posch.set_timesteps(50)
        for t in [int(v) for v in psch.timesteps]:           # every timestep can take a step
            eta = 0.3 if t % 40 == 1 else 0.0
            x, m, z = (torch.randn(1, 4, 2, 8, 8, generator=g) for _ in range(3))
            want = posch.step(m, t, x, eta=eta, noise=z)
            got = psch.step(m, t, x, eta=eta, variance_noise=z if eta > 0 else None).prev_sample
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)
            k_x, k_m, c0, ct, sigma = psch.coefficients(t, eta)
            fused = c0 * (k_x * x - k_m * m) + ct * x + sigma * z
            assert torch.allclose(fused, want, rtol=1e-4, atol=1e-5), (kind, t, eta)
    import golden_util as G
    for c in G.load("ddim_steps.pt")["prediction_cases"]:
        psch = DDIMScheduler(prediction_type=c["kind"])
        psch.set_timesteps(50)
        got = psch.step(c["model_output"], c["t"], c["x"], eta=c["eta"], variance_noise=c["noise"] if c["eta"] > 0 else None)
        rel = lambda a, b: ((a - b).norm() / b.norm()).item()
        # 5e-5: at t = 1 the `sample` branch divides by sqrt(1 - abar) = 1e-2 and the vendored class rounds in fp32
        assert rel(got.prev_sample, c["prev"]) < 5e-5 and rel(got.pred_original_sample, c["x0"]) < 5e-5
    for t in [int(v) for v in sch.timesteps]:                # stock spacing: all 50 timesteps lie inside the alpha table
        sch.coefficients(t, 0.0)
    vsr = DDIMScheduler(timestep_spacing="vsr_linspace")
    ovsr = DDIMSchedule(timestep_spacing="vsr_linspace")
    vsr.set_timesteps(50)
    ovsr.set_timesteps(50)
    assert [int(t) for t in vsr.timesteps] == ovsr.timesteps
    # ... whose first timestep (1001) is past the 1000-entry table, in the vendored text as in the mirror: a clear error
    with pytest.raises(ValueError, match="outside"):
        vsr.coefficients(int(vsr.timesteps[0]))
    with pytest.raises(ValueError, match="outside"):
        vsr.step(torch.zeros(1), int(vsr.timesteps[0]), torch.zeros(1))
    for t in [int(v) for v in vsr.timesteps[1:]]:
        vsr.coefficients(t)
    with pytest.raises(ValueError):
        DDIMScheduler(prediction_type="flow")
    with pytest.raises(ValueError):
        DDIMScheduler().coefficients(981)                # set_timesteps not called
    with pytest.raises(NotImplementedError):
        DDIMScheduler(clip_sample=True)


def test_euler_scheduler_matches_oracle_and_fused_form():
    """EulerDiscreteScheduler mirror vs the independently written oracle tables; step() == fused-kernel coefficients;
    scale_model_input / init_noise_sigma; the generic step loop == oracle loop on a toy denoiser."""
    import numpy as np
    from lavie_amd.scheduling_euler_discrete import EulerDiscreteScheduler
    from oracle.euler import EulerSchedule, cfg_euler_loop
    for n in (50, 7):
        s, o = EulerDiscreteScheduler(), EulerSchedule()
        s.set_timesteps(n)
        o.set_timesteps(n)
        assert np.allclose(s.timesteps.numpy(), o.timesteps, rtol=1e-6)
        assert np.allclose(s._sigmas, o.sigmas, rtol=1e-6)
        assert s.init_noise_sigma == pytest.approx(o.init_noise_sigma, rel=1e-6)
        x, e = torch.randn(16, dtype=torch.float64), torch.randn(16, dtype=torch.float64)
        for i in (0, n // 2, n - 1):
            t = s.timesteps[i]
            k_x, k_e, c0, ct, sigma = s.coefficients(t)
            assert sigma == 0.0
            fused = c0 * (k_x * x - k_e * e) + ct * x
            assert torch.allclose(fused, s.step(e, t, x).prev_sample, rtol=1e-10, atol=1e-10)
            assert torch.allclose(s.scale_model_input(x, t), x / (o.sigmas[i] ** 2 + 1) ** 0.5, rtol=1e-6)
    with pytest.raises(ValueError):
        s.coefficients(123.456)
    g = torch.Generator().manual_seed(1)
    W = torch.randn(4, 4, generator=g) * 0.2
    toy = lambda x, t, c: torch.einsum("oc,bcfhw->bofhw", W, x) * (1 + t / 1000) + c.mean(dim=(1, 2)).reshape(-1, 1, 1, 1, 1)
    lat, pe, ne = torch.randn(1, 4, 2, 4, 4, generator=g), torch.randn(1, 5, 3, generator=g), torch.randn(1, 5, 3, generator=g)
    ref = cfg_euler_loop(toy, lat, pe, ne, num_steps=6, guidance_scale=3.0)
    s = EulerDiscreteScheduler()
    s.set_timesteps(6)
    x = lat * s.init_noise_sigma
    ctx = torch.cat([ne, pe])
    for t in s.timesteps:
        xin = s.scale_model_input(torch.cat([x, x]), t)
        eps = toy(xin, float(t), ctx)
        x = s.step(eps[0:1] + 3.0 * (eps[1:2] - eps[0:1]), t, x).prev_sample
    assert torch.allclose(x, ref, rtol=1e-4, atol=1e-4)


def test_vsr_low_res_noise_schedule_and_add_noise():
    """DDPMScheduler(beta_schedule='squaredcos_cap_v2').add_noise (the VSR pipeline's low-res conditioning) against the
    independently written oracle table."""
    import numpy as np
    from lavie_amd.scheduling_ddpm import DDPMScheduler
    from oracle.vsr_loop import add_low_res_noise, cosine_alphas_cumprod
    sch = DDPMScheduler(beta_schedule="squaredcos_cap_v2")
    assert np.allclose(sch.alphas_cumprod.double().numpy(), cosine_alphas_cumprod(), rtol=2e-5)
    g = torch.Generator().manual_seed(0)
    img, noise = torch.randn(2, 3, 4, 5, 6, generator=g), torch.randn(2, 3, 4, 5, 6, generator=g)
    for level in (0, 20, 350):
        got = sch.add_noise(img, noise, torch.tensor([level, level]))
        assert torch.allclose(got, add_low_res_noise(img, noise, level), rtol=1e-4, atol=1e-5)


def test_ddim_from_config_takes_the_callers_scheduler_config(tmp_path):
    """vsr/sample.py:49-53 builds its DDIM scheduler from the checkpoint's scheduler_config.json (absent from the reference
    tree): `from_config` takes those fields from a dict or a file instead of implying the epsilon / SD-1.4 defaults."""
    import json
    from lavie_amd.scheduling_ddim import DDIMScheduler
    cfg = {"_class_name": "DDIMScheduler", "_diffusers_version": "0.8.0", "beta_start": 0.0001, "beta_end": 0.02,
           "beta_schedule": "scaled_linear", "clip_sample": False, "num_train_timesteps": 1000, "prediction_type": "v_prediction",
           "set_alpha_to_one": False, "steps_offset": 1, "trained_betas": None}
    path = tmp_path / "scheduler_config.json"
    path.write_text(json.dumps(cfg))
    for src in (cfg, str(path)):
        s = DDIMScheduler.from_config(src, beta_schedule="linear")
        assert s.config.prediction_type == "v_prediction" and s.config.beta_schedule == "linear" and s.config.steps_offset == 1
    ref = DDIMScheduler(beta_schedule="scaled_linear", prediction_type="v_prediction")
    got = DDIMScheduler.from_config(cfg)
    assert torch.equal(ref.alphas_cumprod, got.alphas_cumprod)
    with pytest.raises(TypeError):
        DDIMScheduler.from_config(cfg, no_such_field=1)


def test_ddim_from_config_refuses_keys_that_would_change_the_schedule():
    """Keys this class does not model are dropped only while they are inert (their diffusers defaults); a non-null
    trained_betas or rescale_betas_zero_snr=true would make a diffusers scheduler run other betas: an error here."""
    from lavie_amd.scheduling_ddim import DDIMScheduler
    base = {"beta_schedule": "linear", "trained_betas": None, "rescale_betas_zero_snr": False, "skip_prk_steps": True,
            "_class_name": "PNDMScheduler", "clip_sample_range": 1.0}
    DDIMScheduler.from_config(base)
    for key, val in (("trained_betas", [0.1, 0.2]), ("rescale_betas_zero_snr", True), ("clip_sample_range", 2.0)):
        with pytest.raises(NotImplementedError):
            DDIMScheduler.from_config(dict(base, **{key: val}))


class _StubUNet:
    """what VideoGenPipeline touches outside the denoise loop"""
    class config:
        in_channels, sample_size = 4, 64
    device = torch.device("cpu")

    def to(self, device):
        self.moved_to = device
        return self


def test_pipeline_object_walks_the_reference_callers_sequence(monkeypatch):
    """base/pipelines/sample.py:65-89 as a caller sees the pipeline object: the seven constructor keywords, `.to(device)`,
    `enable_xformers_memory_efficient_attention()` (line 72, unconditional), then `pipeline(prompt, image_tensor, video_length=,
    height=, width=, num_inference_steps=, guidance_scale=).video` with the second POSITIONAL argument the fork added (accepted and,
    as in the fork — its image branch sits inside a string literal, pipeline_videogen.py:350-358 — not used).  The denoiser itself is
    stubbed: this is the CPU half of the contract; tests/test_gpu_engine.py runs the loop."""
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    seen = {}

    def fake_denoise(self, latents, ctx, steps, scale, generator=None, callback=None, callback_steps=1, eta=0.0):
        seen.update(latents=tuple(latents.shape), ctx=tuple(ctx.shape), steps=steps, scale=scale)
        return latents

    monkeypatch.setattr(VideoGenPipeline, "denoise", fake_denoise)
    unet = _StubUNet()
    pipe = VideoGenPipeline(vae=None, text_encoder=None, tokenizer=None, scheduler=DDPMScheduler(beta_start=1e-4, beta_end=0.02,
                            beta_schedule="linear"), unet=unet, clip_model=object(), clip_processor=object()).to("cpu")
    assert unet.moved_to == "cpu"
    assert pipe.enable_xformers_memory_efficient_attention() is None          # sample.py:72
    for knob in (pipe.disable_xformers_memory_efficient_attention, pipe.enable_vae_slicing, pipe.disable_vae_slicing,
                 pipe.enable_vae_tiling, pipe.disable_vae_tiling, pipe.enable_attention_slicing, pipe.disable_attention_slicing):
        assert knob() is None
    assert pipe.set_attention_slice("auto") is None
    pe, ne = torch.zeros(1, 77, 768), torch.zeros(1, 77, 768)
    image_tensor = torch.zeros(1, 3, 224, 224)
    out = pipe(None, image_tensor, video_length=16, height=320, width=512, num_inference_steps=50, guidance_scale=7.5,
               prompt_embeds=pe, negative_prompt_embeds=ne, output_type="latent")
    assert tuple(out.video.shape) == (1, 4, 16, 40, 64)
    assert seen == {"latents": (1, 4, 16, 40, 64), "ctx": (2, 77, 768), "steps": 50, "scale": 7.5}
    with pytest.raises(ValueError):                                            # a prompt string without a tokenizer / text encoder
        pipe("A horse playing with a ball", image_tensor, video_length=16, height=320, width=512)


def test_from_sample_yaml_maps_the_reference_config_keys(tmp_path):
    """base/configs/sample.yaml:16-40: the keys sample.py reads become the scheduler (sample_method + beta_*) and the call
    keywords (video_length, image_size, num_sampling_steps, guidance_scale); the reference's own file is used when present."""
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    from lavie_amd.scheduling_ddim import DDIMScheduler
    from lavie_amd.scheduling_euler_discrete import EulerDiscreteScheduler
    text = ("text_prompt: ['A horse playing with a ball']\nvideo_length: 16\nimage_size: [320, 512]\nbeta_start: 0.0001\n"
            "beta_end: 0.02\nbeta_schedule: \"linear\"\nuse_fp16: true\nenable_xformers_memory_efficient_attention: false\n"
            "seed: null\nguidance_scale: 7.5\nsample_method: 'ddpm'\nnum_sampling_steps: 50\n")
    path = tmp_path / "sample.yaml"
    path.write_text(text)
    sources = [str(path)]
    ref = "/root/reference/base/configs/sample.yaml"
    if os.path.isfile(ref):
        sources.append(ref)
    for src in sources:
        pipe, kw, cfg = VideoGenPipeline.from_sample_yaml(src, unet=_StubUNet())
        assert isinstance(pipe.scheduler, DDPMScheduler)
        assert pipe.scheduler.config.beta_start == 1e-4 and pipe.scheduler.config.beta_end == 0.02
        assert kw == dict(video_length=16, height=320, width=512, num_inference_steps=50, guidance_scale=7.5)
        assert cfg["seed"] is None and cfg["text_prompt"] == ["A horse playing with a ball"]
    pipe, _, _ = VideoGenPipeline.from_sample_yaml({"sample_method": "ddim", "beta_schedule": "scaled_linear"}, unet=_StubUNet())
    assert isinstance(pipe.scheduler, DDIMScheduler) and pipe.scheduler.config.beta_schedule == "scaled_linear"
    pipe, _, _ = VideoGenPipeline.from_sample_yaml({"sample_method": "eulerdiscrete"}, unet=_StubUNet())
    assert isinstance(pipe.scheduler, EulerDiscreteScheduler)
    with pytest.raises(NotImplementedError):
        VideoGenPipeline.from_sample_yaml({"sample_method": "pndm"}, unet=_StubUNet())


def test_host_sanitizer_build_is_clean():
    """`make asan` compiles every native source HOST-ONLY with AddressSanitizer + UBSan against the stub HIP
    runtime (lavie_amd/csrc/hostcheck/) and walks the C ABI — parameter inventory, weight packing, workspace dry run, planners,
    launch geometry, argument checks — for the base, interpolation and VSR variants.  CPU only."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "lavie_amd", "csrc"), "-j", "8", "asan"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "hostcheck: ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_device_code_has_no_op_sel_modified_packed_fp32():
    """Round 4 (profiles/r04_packed_fp32_op_sel_fault.txt): the low half of `v_pk_fma_f32 ... op_sel:[0,1,0]` was measured
    reading 0 in lanes 48 - 63 at three waves per SIMD.  The cure is not to emit such forms (hand-written v_fma_f32 in the GEMM
    epilogue; three files built without packed FP32).  This walks the gfx950 code objects inside the built library and fails on any
    packed-FP32 instruction that carries an `op_sel:` modifier, so that a compiler or source change cannot bring one back unseen."""
    import shutil
    import struct
    import subprocess
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        objdump = shutil.which("llvm-objdump")
    if objdump is None:
        pytest.skip("no llvm-objdump")
    blob = open(_lib.LIB_PATH, "rb").read()
    packed, bad, objects = 0, [], 0
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):          # one uncompressed offload bundle per translation unit
        base = m.start()
        pos = base + 32
        for _ in range(struct.unpack_from("<Q", blob, base + 24)[0]):
            off, size, idlen = struct.unpack_from("<QQQ", blob, pos)
            ident = blob[pos + 24:pos + 24 + idlen].decode()
            pos += 24 + idlen
            if "gfx950" not in ident or size == 0:
                continue
            objects += 1
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(blob[base + off:base + off + size])
                f.flush()
                text = subprocess.run([objdump, "-d", f.name], capture_output=True, text=True, check=True).stdout
            kernel = "?"
            for line in text.splitlines():
                head = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if head:
                    kernel = head.group(1)
                elif re.search(r"\bv_pk_\w+_f32\b", line):
                    packed += 1
                    if "op_sel:" in line:
                        bad.append(f"{kernel}: {line.strip()[:100]}")
    assert objects >= 10 and packed > 1000, (objects, packed)            # the walk saw the kernels (attention / GEMMs use packed FP32 freely)
    assert not bad, "op_sel-modified packed FP32 in device code:\n" + "\n".join(bad[:10])
