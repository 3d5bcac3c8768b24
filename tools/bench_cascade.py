#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE GPU: the full cascade for one prompt — base T2V (16 x 320 x 512, 50 DDPM steps) ->
interpolation (61 frames, 50 DDIM steps) -> VSR (61 x 1280 x 2048, 8-frame chunks, DDIM) -> VAE decode — with random-init
fp16 weights of the full architectures and synthetic text embeddings.  Prints one JSON line with per-stage seconds.
(The 8-GPU prompt-DP number of configs[4] is this per-prompt time on every rank: prompts are independent.)
Usage: python tools/bench_cascade.py [--vsr-steps 50] [--no-final-decode]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lavie_amd import spec, weights  # noqa: E402
from lavie_amd.autoencoder_kl import AutoencoderKL  # noqa: E402
from lavie_amd.cascade import decode_frames, interpolation_condition  # noqa: E402
from lavie_amd.config import BASE_CONFIG, INTERPOLATION_CONFIG, VSR_CONFIG  # noqa: E402
from lavie_amd.interpolation import UNet3DConditionModel as InterpUNet  # noqa: E402
from lavie_amd.interpolation import create_diffusion  # noqa: E402
from lavie_amd.pipeline_videogen import VideoGenPipeline  # noqa: E402
from lavie_amd.scheduling_ddim import DDIMScheduler  # noqa: E402
from lavie_amd.scheduling_ddpm import DDPMScheduler  # noqa: E402
from lavie_amd.unet import UNet3DConditionModel  # noqa: E402
from lavie_amd.vsr import UNet3DVSRModel, VideoUpscalePipeline, upscale_in_chunks  # noqa: E402


def note(msg):
    print(f"[{time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def load(net, cfg, seed, dev):
    sd = weights.synth_state_dict(spec.param_shapes(cfg), seed)
    net.load_state_dict({k: v.half() for k, v in sd.items()})
    note(f"weights ready: {type(net).__module__}")
    return net.to(dev, torch.float16)


def timed(fn, what=""):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    note(f"{what}: {dt:.2f} s")
    return out, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vsr-steps", type=int, default=50)
    ap.add_argument("--no-final-decode", action="store_true")
    ap.add_argument("--stock-vae", action="store_true", help="decode with the stock PyTorch modules instead of lavie_amd.vae_hip")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    base = load(UNet3DConditionModel(init_weights=False, sample_size=64, cross_attention_dim=768), BASE_CONFIG, 0, dev)
    interp = load(InterpUNet(init_weights=False, sample_size=64, in_channels=8, cross_attention_dim=768, use_first_frame=True),
                  INTERPOLATION_CONFIG, 1, dev)
    vsr = load(UNet3DVSRModel(init_weights=False, sample_size=128, down_temporal_idx=(0, 1, 2, 3), mid_temporal=True,
                              up_temporal_idx=(0, 1, 2, 3)), VSR_CONFIG, 2, dev)
    vae = AutoencoderKL().to(dev, torch.float16).eval()
    vsr_vae = AutoencoderKL(block_out_channels=(128, 256, 512), sample_size=256, scaling_factor=0.08333).to(dev).eval()   # fp32 (:737)
    if not a.stock_vae:          # decoders on the engine's conv / GroupNorm / GEMM operators (encode stays stock)
        from lavie_amd.vae_hip import HipAutoencoderKL
        vae, vsr_vae = HipAutoencoderKL(vae), HipAutoencoderKL(vsr_vae)
    g = torch.Generator().manual_seed(0)
    e768 = lambda: torch.randn(1, 77, 768, generator=g).to(dev)
    e1024 = lambda: torch.randn(1, 77, 1024, generator=g).to(dev)
    pe, ne, ipe, ine, vpe, vne = e768(), e768(), e768(), e768(), e1024(), e1024()
    base_pipe = VideoGenPipeline(unet=base, scheduler=DDPMScheduler())
    vsr_pipe = VideoUpscalePipeline(unet=vsr, scheduler=DDIMScheduler())
    diffusion = create_diffusion("50")
    t = {}
    lat16, t["base_denoise_s"] = timed(what="base denoise", fn=lambda: base_pipe(prompt_embeds=pe, negative_prompt_embeds=ne, height=320, width=512,
                                                         video_length=16, num_inference_steps=50, guidance_scale=7.5,
                                                         generator=torch.Generator().manual_seed(1), output_type="latent").video)
    frames16, t["base_vae_decode_s"] = timed(what="base vae decode", fn=lambda: decode_frames(vae, lat16, 0.18215))
    copied, t["interp_vae_encode_s"] = timed(what="interp vae encode", fn=lambda: interpolation_condition(vae, frames16, 61))
    z = torch.randn(1, 4, 61, 40, 64, device=dev)
    z2, c2 = torch.cat([z] * 2), torch.cat([copied] * 2)
    ctx = torch.cat([ipe, ine], dim=0)
    lat61, t["interp_denoise_s"] = timed(what="interp denoise", fn=lambda: diffusion.ddim_sample_loop(
        interp.forward_with_cfg, z2.shape, z2, clip_denoised=False, model_kwargs=dict(encoder_hidden_states=ctx, class_labels=None),
        device=dev, mask=None, x_start=c2, use_concat=True, copy_no_mask=True).chunk(2, dim=0)[0])
    frames61, t["interp_vae_decode_s"] = timed(what="interp vae decode", fn=lambda: decode_frames(vae, lat61, 0.18215))
    gen = torch.Generator().manual_seed(2)
    chunks, t["vsr_denoise_s"] = [], 0.0
    for s0 in range(0, 61, 8):                        # upscale_in_chunks (vsr/sample.py:104-123), timed chunk by chunk
        out, dt = timed(what=f"vsr denoise frames {s0}-{min(61, s0 + 8)}", fn=lambda: vsr_pipe(
            image=frames61[:, :, s0:min(61, s0 + 8)], prompt_embeds=vpe, negative_prompt_embeds=vne,
            num_inference_steps=a.vsr_steps, guidance_scale=5.0, noise_level=50, generator=gen).images)
        chunks.append(out)
        t["vsr_denoise_s"] += dt
    up = torch.cat(chunks, dim=2)
    shape = None
    if not a.no_final_decode:
        final, t["vsr_vae_decode_s"] = timed(what="vsr vae decode", fn=lambda: decode_frames(vsr_vae, up.float(), None, chunk=1))
        shape = list(final.shape)
        assert torch.isfinite(final).all()
    assert torch.isfinite(up).all()
    total = sum(t.values())
    print(json.dumps({"metric": "full cascade, one prompt, one GPU (BASELINE.json configs[4] per-rank work)", "seconds_total": total,
                      "stages": {k: round(v, 3) for k, v in t.items()}, "vsr_steps": a.vsr_steps, "output_frames_shape": shape,
                      "dtype": "f16 (UNets, base VAE) / f32 (final VSR VAE decode, as the reference)", "data": "synthetic",
                      "vae_decode": "stock PyTorch" if a.stock_vae else "lavie_amd.vae_hip (engine operators)",
                      "hbm_allocated_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))


if __name__ == "__main__":
    main()
