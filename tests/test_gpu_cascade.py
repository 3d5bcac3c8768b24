"""-m gpu: the on-device cascade driver (BASELINE.json configs[4]) with reduced-width models of all three
stages and small stock VAEs: shapes, finiteness, determinism, and that the stage hand-overs are the ones the reference's
three scripts perform (frame resampling 16 -> 61, copied low-frame-rate conditioning, 8-frame VSR chunks)."""
import pytest
import torch

import golden_util as G

pytestmark = pytest.mark.gpu


def test_interpolation_condition_matches_reference_indexing():
    """interpolation/sample.py:74-79, 144-147 with an identity 'VAE': frame f of the condition comes from base frame
    linspace(0, 15, 61)[4 * ((f + 1) // 4)]."""
    import numpy as np
    from types import SimpleNamespace
    from lavie_amd.cascade import interpolation_condition

    class IdVae(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def encode(self, x):
            return SimpleNamespace(latent_dist=SimpleNamespace(sample=lambda g=None: x[:, :1].repeat(1, 4, 1, 1)))

    frames = torch.arange(16, dtype=torch.float32).reshape(1, 1, 16, 1, 1).repeat(1, 3, 1, 2, 2).cuda()
    cond = interpolation_condition(IdVae().cuda(), frames, 61)
    assert cond.shape == (1, 4, 61, 2, 2)
    idx = np.linspace(0, 15, 61, dtype=int)
    want = torch.tensor([idx[4 * ((f + 1) // 4)] for f in range(61)], dtype=torch.float32) * 0.18215
    assert torch.allclose(cond[0, 0, :, 0, 0].cpu(), want)


def test_cascade_reduced_models():
    from lavie_amd import spec
    from lavie_amd.autoencoder_kl import AutoencoderKL
    from lavie_amd.cascade import text_to_video_cascade
    from lavie_amd.config import UNetConfig
    from lavie_amd.interpolation import UNet3DConditionModel as InterpUNet
    from lavie_amd.interpolation import create_diffusion
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    from lavie_amd.scheduling_ddim import DDIMScheduler
    from lavie_amd.unet import UNet3DConditionModel
    from lavie_amd.vsr import UNet3DVSRModel, VideoUpscalePipeline

    def load(net, cfg, seed):
        net.load_state_dict({k: v.half() for k, v in G.synth16(spec.param_shapes(cfg), seed).items()})
        return net.to("cuda", torch.float16)

    two = dict(down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"), up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"))
    base = load(UNet3DConditionModel(init_weights=False, sample_size=8, block_out_channels=(256, 512), cross_attention_dim=128, **two),
                UNetConfig(block_out_channels=(256, 512), cross_attention_dim=128, attn_levels=(True, False)), 1)
    interp = load(InterpUNet(init_weights=False, sample_size=8, in_channels=8, block_out_channels=(256, 512), cross_attention_dim=128,
                             use_first_frame=True, **two),
                  UNetConfig(in_channels=8, block_out_channels=(256, 512), cross_attention_dim=128, attn_levels=(True, False),
                             sparse_causal_attn1=True, temporal_plain=True, ff_before_temporal=True), 2)
    vsr = load(UNet3DVSRModel(init_weights=False, sample_size=8, block_out_channels=(256, 512), cross_attention_dim=128,
                              layers_per_block=1, down_block_types=("DownBlock3D", "CrossAttnDownBlock3D"),
                              up_block_types=("CrossAttnUpBlock3D", "UpBlock3D"), only_cross_attention=(True, False),
                              down_temporal_idx=(0, 1), mid_temporal=True, up_temporal_idx=(0, 1)),
               UNetConfig(in_channels=7, block_out_channels=(256, 512), cross_attention_dim=128, attn_levels=(False, True),
                          layers_per_block=1, vsr_blocks=True, only_cross_attention=(True, False), vsr_temporal_modules=True,
                          num_class_embeds=1000), 3)
    torch.manual_seed(0)
    vae = AutoencoderKL(block_out_channels=(32, 64, 64, 64), layers_per_block=1, norm_num_groups=8).cuda().eval()
    vsr_vae = AutoencoderKL(block_out_channels=(32, 64, 64), layers_per_block=1, norm_num_groups=8, scaling_factor=0.08333).cuda().eval()
    g = torch.Generator().manual_seed(4)
    emb = lambda: torch.randn(1, 77, 128, generator=g)
    pe, ne, ipe, ine, vpe, vne = (emb() for _ in range(6))
    outs = []
    for _ in range(2):
        torch.manual_seed(11)          # the interpolation noise / VAE posterior samples are drawn on the device
        outs.append(text_to_video_cascade(
            VideoGenPipeline(unet=base), interp, create_diffusion("2"), VideoUpscalePipeline(unet=vsr, scheduler=DDIMScheduler()),
            vae, vsr_vae, pe, ne, vpe, vne, ipe, ine, height=64, width=64, base_steps=2, vsr_steps=2, noise_level=20,
            generator=torch.Generator().manual_seed(5)))
    b, i, u, frames = outs[0]
    assert b.shape == (1, 4, 16, 8, 8) and i.shape == (1, 4, 61, 8, 8) and u.shape == (1, 4, 61, 64, 64)
    assert frames.shape == (1, 3, 61, 256, 256)
    for t in (b, i, u, frames):
        assert torch.isfinite(t).all()
    assert float(frames.abs().max()) <= 1.0
    # the HIP stages are bit-reproducible (tests/test_gpu_engine.py); the stock MIOpen convolutions of the VAEs between them
    # are not guaranteed to be, so two runs agree to rounding, not to the bit
    assert torch.equal(outs[1][0], b)
    assert ((outs[1][2] - u).norm() / u.norm()).item() < 2e-2


@pytest.mark.parametrize("widths,size", [((128, 256, 512, 512), (2, 8, 16)), ((128, 256, 512), (1, 16, 24))])
def test_hip_vae_decoder_matches_stock_module(widths, size):
    """AutoencoderKL.decode on the engine's conv / GroupNorm / GEMM operators (lavie_amd/vae_hip.py) against the stock
    PyTorch module with the same weights in fp32 (both SD VAE layouts: factor 8 and the x4-upscaler's factor 4)."""
    from lavie_amd.autoencoder_kl import AutoencoderKL
    from lavie_amd.vae_hip import HipAutoencoderKL
    torch.manual_seed(3)
    vae = AutoencoderKL(block_out_channels=widths).cuda().eval()
    n, h, w = size
    z = torch.randn(n, 4, h, w, device="cuda")
    ref = vae.decode(z).sample
    got = HipAutoencoderKL(vae).decode(z).sample
    assert got.shape == ref.shape == (n, 3, h * 2 ** (len(widths) - 1), w * 2 ** (len(widths) - 1))
    assert ((got.float() - ref).norm() / ref.norm()).item() < 1e-2


def test_hip_vae_encoder_matches_stock_module():
    from lavie_amd.autoencoder_kl import AutoencoderKL
    from lavie_amd.vae_hip import HipAutoencoderKL
    torch.manual_seed(5)
    vae = AutoencoderKL().cuda().eval()
    x = torch.rand(2, 3, 64, 96, device="cuda") * 2 - 1
    ref = vae.encode(x).latent_dist
    got = HipAutoencoderKL(vae).encode(x).latent_dist
    assert got.mean.shape == ref.mean.shape == (2, 4, 8, 12)
    assert ((got.mean - ref.mean).norm() / ref.mean.norm()).item() < 1e-2
    assert ((got.logvar - ref.logvar).norm() / ref.logvar.norm()).item() < 1e-2
