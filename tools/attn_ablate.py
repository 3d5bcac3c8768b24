#!/usr/bin/env python3
"""Ablations of the attention kernel at the top-level self-attention shape (head dim 40, 2560 tokens, 32 frames): which of
the softmax VALU work, the MFMAs and the staging / barrier chain the time goes to (results of the ablated modes are wrong).
Usage: python tools/attn_ablate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops  # noqa: E402
from tools.bench_attn_sc import timeit  # noqa: E402

lib = _lib.load()
C, D, nb, heads = 320, 2560, 32, 8
qkv = (torch.randn(nb * D, 3 * C, device="cuda") * 0.5).half()
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
for name, mode in (("full", 0), ("no-softmax-VALU", 0x12), ("no-MFMA", 0x22), ("staging+barriers only", 0x32), ("QT=1", 1), ("QT=4", 4), ("VALU row sums", 0x40)):
    lib.lavie_debug_attention_qt(mode)
    print(f"{name:24s} {timeit(lambda: ops.attention(q, k, v, nb, D, D, heads), iters=10):8.1f} us", flush=True)
