#!/usr/bin/env python3
"""Runs a handful of launches of one hot-operator shape (for rocprofv3 --pmc passes).
Usage: python tools/pmc_probe.py conv|linear|attn|temporal [mode]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops  # noqa: E402

kind = sys.argv[1]
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
_lib.load().lavie_debug_force_tile(mode)
dev = "cuda"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).half()
if kind == "conv":          # L0 resnet conv with concat input: 32 x 40x64, 640+320 -> 320
    x1, x2 = rnd(32 * 40 * 64, 640), rnd(32 * 40 * 64, 320)
    wp = ops.pack_conv3x3(rnd(320, 960, 3, 3) / 90)
    bias = torch.randn(320, device=dev)
    fn = lambda: ops.conv3x3(x1, wp, bias, 32, 40, 64, x2=x2)
elif kind == "linear":      # L0 attention out-projection + residual
    a, w, r = rnd(81920, 320), rnd(320, 320) / 18, rnd(81920, 320)
    bias = torch.randn(320, device=dev)
    out = torch.empty_like(r)
    fn = lambda: ops.linear(a, w, bias=bias, residual=r, out=out)
elif kind == "attn":
    qkv = rnd(32 * 2560, 960)
    fn = lambda: ops.attention(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], nb=32, lq=2560, lk=2560, heads=8)
else:
    qkv = rnd(2 * 16 * 2560, 960)
    bias = torch.randn(8, 16, 16, device=dev)
    cos, sin = ops.rotary_tables(16, 32)
    fn = lambda: ops.temporal_attention(qkv, 2, 16, 2560, 8, bias, cos, sin)
for _ in range(5):
    fn()
torch.cuda.synchronize()
