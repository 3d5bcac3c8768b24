#!/usr/bin/env python3
"""conv shapes with whatever library LAVIE_HIP_LIB points at (A/B of two builds on the same GPU)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops
from tools.bench_ops import timeit, rnd
lib = _lib.load()
try:
    lib.lavie_debug_force_splits(1)
except AttributeError:
    pass
for ni, h, w, c1, c2, cout in ((32, 40, 64, 320, 0, 320), (32, 40, 64, 640, 320, 320), (32, 20, 32, 640, 0, 640),
                               (32, 20, 32, 1280, 640, 640), (32, 10, 16, 1280, 0, 1280), (32, 5, 8, 1280, 0, 1280)):
    x1 = rnd(ni * h * w, c1)
    x2 = rnd(ni * h * w, c2) if c2 else None
    wp = ops.pack_conv3x3(rnd(cout, c1 + c2, 3, 3) / math.sqrt(9 * (c1 + c2)))
    bias = torch.randn(cout, device="cuda")
    us = timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2))
    print(f"{os.environ.get('LAVIE_HIP_LIB', 'default')[-24:]:>24} {ni} {h}x{w} {c1}+{c2}->{cout}: {us:8.1f} us {2.0 * ni * h * w * cout * 9 * (c1 + c2) / us / 1e6:6.0f} TF")
