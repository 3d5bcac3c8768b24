#!/usr/bin/env python3
"""BASELINE.json configs[3] ("temporal-attn stress"): the frame-interpolation UNet at 61 frames, 320x512 (latent 40x64),
DDIM 50 steps, guidance batch 2, random-init fp16 weights, synthetic conditioning.  Prints one JSON line:
interpolated-clip latents/s, ms per UNet forward and the per-class device time of one instrumented forward.
Not the driver's bench (bench.py measures the headline base-model metric); same measurement rules.
Usage: python tools/bench_interp.py [--steps 50] [--frames 61] [--loops 1]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lavie_amd import _lib, spec, weights  # noqa: E402
from lavie_amd.config import INTERPOLATION_CONFIG  # noqa: E402
from lavie_amd.interpolation import UNet3DConditionModel, create_diffusion  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--frames", type=int, default=61)
    ap.add_argument("--loops", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    lib = _lib.load()
    if os.environ.get("LAVIE_TEMPORAL_BUDGET"):          # tuning: LDS bytes one temporal-attention workgroup may stage
        lib.lavie_debug_temporal_budget(int(os.environ["LAVIE_TEMPORAL_BUDGET"]))
    sd = weights.synth_state_dict(spec.param_shapes(INTERPOLATION_CONFIG), 0)
    net = UNet3DConditionModel(init_weights=False, sample_size=64, in_channels=8, cross_attention_dim=768, use_first_frame=True)
    net.load_state_dict({k: v.half() for k, v in sd.items()})
    del sd
    net = net.to(dev, torch.float16)
    g = torch.Generator().manual_seed(0)
    F, H, W = a.frames, 40, 64
    z = torch.cat([torch.randn(1, 4, F, H, W, generator=g)] * 2).to(dev)
    xs = torch.cat([torch.randn(1, 4, F, H, W, generator=g)] * 2).to(dev)
    ctx = torch.randn(2, 77, 768, generator=g).to(dev)
    d = create_diffusion(str(a.steps))
    kw = dict(clip_denoised=False, model_kwargs=dict(encoder_hidden_states=ctx, class_labels=None), mask=None, x_start=xs,
              use_concat=True, copy_no_mask=True)
    d._ddim_loop_hip(net, z.shape, z, kw["model_kwargs"], 0.0, xs, True, max_steps=2)       # warm-up: 2 steps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.loops):
        out = d.ddim_sample_loop(net.forward_with_cfg, z.shape, z, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.loops
    assert torch.isfinite(out).all()
    # one instrumented forward (outside the timed region)
    x8 = torch.cat([z, xs], dim=1).half()
    bench.profile_begin(lib, 0xFF, 4096)
    net(x8, 500, encoder_hidden_states=ctx.half())
    classes = bench.profile_end(lib)
    print(json.dumps({
        "metric": "interpolated-clip latents/sec (61f x 320x512, 50 DDIM steps)", "value": 1.0 / dt, "unit": "clip-latents/s",
        "n_gpus": 1, "ms_per_clip": dt * 1e3, "ms_per_unet_forward": dt * 1e3 / a.steps, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"interpolation UNet, {F} frames, latent 40x64, guidance batch 2, DDIM {a.steps} steps"},
        "kernel_breakdown": [dict(name=c["name"], launches=c["launches"], ms=round(c["ms"], 3),
                                  tflops=round(c["flops"] / c["ms"] / 1e9, 1) if c["ms"] else 0,
                                  gbps=round(c["bytes"] / c["ms"] / 1e6, 1) if c["ms"] else 0) for c in classes if c["launches"]],
        "weight_bytes": int(lib.lavie_unet_weight_bytes(net.engine_handle())),
        "workspace_bytes": int(lib.lavie_unet_workspace_bytes(net.engine_handle())),
    }))


if __name__ == "__main__":
    main()
