#!/usr/bin/env python3
"""The VSR stage's UNet (UNet3DVSRModel, vsr/configs/unet_3d_config.json, 691 M parameters) at its working size: one 8-frame
chunk (vsr/sample.py chunking) of 320x512 latents, guidance batch 2, random-init fp16 weights, synthetic inputs.
Prints one JSON line: ms per UNet forward and the per-class device time of one instrumented forward.
Usage: python tools/bench_vsr.py [--frames 8] [--height 320] [--width 512] [--iters 3]"""
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
from lavie_amd.config import VSR_CONFIG  # noqa: E402
from lavie_amd.vsr import UNet3DVSRModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    lib = _lib.load()
    sd = weights.synth_state_dict(spec.param_shapes(VSR_CONFIG), 0)
    net = UNet3DVSRModel(init_weights=False, sample_size=128, down_temporal_idx=(0, 1, 2, 3), mid_temporal=True,
                         up_temporal_idx=(0, 1, 2, 3))
    net.load_state_dict({k: v.half() for k, v in sd.items()})
    del sd
    net = net.to(dev, torch.float16)
    g = torch.Generator().manual_seed(0)
    F, H, W = a.frames, a.height, a.width
    x = torch.randn(2, 4, F, H, W, generator=g).half().to(dev)
    low = torch.randn(2, 3, F, H, W, generator=g).half().to(dev)
    ctx = torch.randn(2, 77, 1024, generator=g).half().to(dev)
    labels = torch.tensor([20, 20])
    out = net(x, 500, low, encoder_hidden_states=ctx, class_labels=labels).sample       # warm-up (packs weights, sizes workspace)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.iters):
        out = net(x, 500 - i, low, encoder_hidden_states=ctx, class_labels=labels).sample
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    assert torch.isfinite(out).all()
    bench.profile_begin(lib, 0xFF, 8192)
    net(x, 400, low, encoder_hidden_states=ctx, class_labels=labels)
    classes = bench.profile_end(lib)
    print(json.dumps({
        "metric": "VSR UNet forward (8-frame chunk, 320x512 latents, guidance batch 2)", "ms_per_unet_forward": dt * 1e3,
        "dtype": "f16", "data": "synthetic", "config": {"workload": f"UNet3DVSRModel, {F} frames, {H}x{W}, batch 2"},
        "kernel_breakdown": [dict(name=c["name"], launches=c["launches"], ms=round(c["ms"], 3),
                                  tflops=round(c["flops"] / c["ms"] / 1e9, 1) if c["ms"] else 0,
                                  gbps=round(c["bytes"] / c["ms"] / 1e6, 1) if c["ms"] else 0) for c in classes if c["launches"]],
        "weight_bytes": int(lib.lavie_unet_weight_bytes(net.engine_handle())),
        "workspace_bytes": int(lib.lavie_unet_workspace_bytes(net.engine_handle())),
    }))


if __name__ == "__main__":
    main()
