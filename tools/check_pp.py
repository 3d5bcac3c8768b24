#!/usr/bin/env python3
"""Ping-pong kernel vs the 128-row kernel on full-size shapes: max abs difference and time (MI355X)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops
from tools.bench_ops import rnd, timeit

lib = _lib.load()
modes = [int(m, 0) for m in sys.argv[1:]] or [0, 3, 0x43]
for ni, h, w, c1, c2, cout in ((32, 40, 64, 640, 320, 320), (32, 20, 32, 640, 0, 640), (32, 10, 16, 1280, 0, 1280)):
    x1 = rnd(ni * h * w, c1)
    x2 = rnd(ni * h * w, c2) if c2 else None
    wp = ops.pack_conv3x3(rnd(cout, c1 + c2, 3, 3) / math.sqrt(9 * (c1 + c2)))
    bias = torch.randn(cout, device="cuda")
    outs = {}
    row = f"conv {ni} {h}x{w} {c1}+{c2}->{cout} | "
    for m in modes:
        lib.lavie_debug_force_tile(m)
        outs[m] = ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2).float()
        us = timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2), iters=30)
        fl = 2.0 * ni * h * w * cout * 9 * (c1 + c2)
        row += f"mode {m:#x}: {us:7.1f} us {fl / us / 1e6:5.0f} TF/s maxdiff {float((outs[m] - outs[modes[0]]).abs().max()):.3g} | "
    print(row)
for M, N, K in ((81920, 320, 1280), (20480, 640, 2560), (20480, 640, 640)):
    a, w_, r = rnd(M, K), rnd(N, K) / math.sqrt(K), rnd(M, N)
    bias = torch.randn(N, device="cuda")
    row = f"linear {M} {N} {K} | "
    outs = {}
    for m in modes:
        lib.lavie_debug_force_tile(m)
        outs[m] = ops.linear(a, w_, bias=bias, residual=r).float()
        us = timeit(lambda: ops.linear(a, w_, bias=bias, residual=r), iters=30)
        row += f"mode {m:#x}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF/s maxdiff {float((outs[m] - outs[modes[0]]).abs().max()):.3g} | "
    print(row)
lib.lavie_debug_force_tile(0)
