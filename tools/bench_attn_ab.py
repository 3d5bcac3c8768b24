#!/usr/bin/env python3
"""Attention core: LDS-DMA staging (default) vs the register-staged kernels (lavie_debug_attention_qt(0x50)) on the
model's shapes, in one process (MI355X); max abs difference between the two (both fp32-accumulating, same order: 0)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops
from tools.bench_ops import rnd, timeit

lib = _lib.load()
cases = [(32, 2560, 320, 2560, 1, "L0 self"), (32, 640, 640, 640, 1, "L1 self"), (32, 160, 1280, 160, 1, "L2 self"),
         (32, 40, 1280, 40, 1, "mid self"), (32, 2560, 320, 77, 16, "L0 text"), (32, 640, 640, 77, 16, "L1 text"),
         (32, 160, 1280, 77, 16, "L2 text")]
for nb, l, c, lk, div, name in cases:
    if lk == l:
        qkv = rnd(nb * l, 3 * c)
        fn = lambda: ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], nb=nb, lq=l, lk=l, heads=8)
    else:
        q, kv = rnd(nb * l, c), rnd(nb // div * lk, 2 * c)
        fn = lambda: ops.attention(q, kv[:, :c], kv[:, c:], nb=nb, lq=l, lk=lk, heads=8, kv_batch_div=div)
    row = f"{name:9s} nb={nb} Lq={l} Lk={lk} dh={c // 8:3d} | "
    outs = []
    for mode in (0x50, 0, 0x50, 0):
        lib.lavie_debug_attention_qt(mode)
        outs.append(fn().float().clone())
        us = timeit(fn, iters=30)
        row += f"{'reg' if mode else 'dma'} {us:7.1f} us {4.0 * nb * l * lk * c / us / 1e6:5.0f} TF/s | "
    row += f"maxdiff {float((outs[0] - outs[1]).abs().max()):.3g}"
    print(row, flush=True)
lib.lavie_debug_attention_qt(0)
# sparse-causal (interpolation model): F = 61 frames at L0
nb, l, c = 61, 2560, 320
qkv = rnd(nb * l, 3 * c)
for mode in (0x50, 0):
    lib.lavie_debug_attention_qt(mode)
    fn = lambda: ops.sparse_causal_attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], nb=nb, frames=61, d=l, heads=8)
    us = timeit(fn, iters=10)
    print(f"sparse-causal L0 F=61 {'reg' if mode else 'dma'}: {us:8.1f} us", flush=True)
lib.lavie_debug_attention_qt(0)
