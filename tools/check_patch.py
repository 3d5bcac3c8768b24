#!/usr/bin/env python3
"""Halo-patch conv kernel (mode 5) vs the automatic choice (mode 0) on the model's conv shapes: time and max difference."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops
from tools.bench_ops import rnd, timeit

lib = _lib.load()
cases = [(32, 40, 64, 320, 0, 320, 0), (32, 40, 64, 640, 320, 320, 0), (32, 40, 64, 640, 0, 320, 0), (32, 20, 32, 640, 0, 640, 0),
         (32, 20, 32, 1280, 640, 640, 0), (32, 20, 32, 320, 0, 640, 0), (32, 10, 16, 1280, 0, 1280, 2), (32, 10, 16, 1280, 1280, 1280, 2),
         (32, 5, 8, 1280, 0, 1280, 8), (32, 5, 8, 1280, 0, 1280, 4)]
if os.environ.get("PATCH_VSR"):      # the VSR stage's levels: 16 frames (batch 2 x 8) at 320x512 / 160x256 / 80x128 / 40x64
    cases = [(16, 320, 512, 256, 0, 256, 0), (16, 320, 512, 512, 0, 256, 0), (16, 160, 256, 512, 0, 512, 0), (16, 160, 256, 1024, 0, 512, 0),
             (16, 80, 128, 1024, 0, 1024, 0), (16, 40, 64, 1024, 0, 1024, 0)]
for ni, h, w, c1, c2, cout, s5 in cases:
    x1 = rnd(ni * h * w, c1)
    x2 = rnd(ni * h * w, c2) if c2 else None
    wp = ops.pack_conv3x3(rnd(cout, c1 + c2, 3, 3) / math.sqrt(9 * (c1 + c2)))
    bias = torch.randn(cout, device="cuda")
    row = f"conv {ni} {h}x{w} {c1}+{c2}->{cout} | "
    ref = None
    for rep in range(2):
        for m, sp in ((0, 0), (5, s5)):
            lib.lavie_debug_force_tile(m)
            lib.lavie_debug_force_splits(sp)
            out = ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2).float()
            if ref is None:
                ref = out
            us = timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2), iters=30)
            fl = 2.0 * ni * h * w * cout * 9 * (c1 + c2)
            row += f"mode {m} s{sp}: {us:7.1f} us {fl / us / 1e6:5.0f} TF/s diff {float((out - ref).abs().max()):.2g} | "
    print(row, flush=True)
lib.lavie_debug_force_tile(0)
lib.lavie_debug_force_splits(0)
