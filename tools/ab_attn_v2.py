#!/usr/bin/env python3
"""Attention core, round-4 softmax (scale folded into Q, running maximum subtracted by the matrix pipe, cross-lane maximum only in
the rescale branch) against the round-3 one (lavie_debug_attention_qt(0x60)) on the model's shapes, one process, interleaved
(MI355X); the relative L2 difference between the two and of each against an fp32 torch reference on a slice."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops
from tools.bench_ops import rnd, timeit

lib = _lib.load()
cases = [(32, 2560, 320, 2560, 1, "L0 self"), (16, 2560, 320, 2560, 1, "L0 self/2"), (32, 640, 640, 640, 1, "L1 self"),
         (32, 160, 1280, 160, 1, "L2 self"), (32, 2560, 320, 77, 16, "L0 text"), (32, 640, 640, 77, 16, "L1 text")]
for nb, l, c, lk, div, name in cases:
    if lk == l:
        qkv = rnd(nb * l, 3 * c)
        fn = lambda: ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], nb=nb, lq=l, lk=l, heads=8)
        q_, k_, v_ = qkv[:l, :c], qkv[:l, c:2 * c], qkv[:l, 2 * c:]
    else:
        q, kv = rnd(nb * l, c), rnd(nb // div * lk, 2 * c)
        fn = lambda: ops.attention(q, kv[:, :c], kv[:, c:], nb=nb, lq=l, lk=lk, heads=8, kv_batch_div=div)
        q_, k_, v_ = q[:l], kv[:lk, :c], kv[:lk, c:]
    dh = c // 8
    ref = torch.softmax((q_.float().view(l, 8, dh).transpose(0, 1) @ k_.float().view(-1, 8, dh).permute(1, 2, 0)) * dh ** -0.5, -1) \
        @ v_.float().view(-1, 8, dh).transpose(0, 1)
    ref = ref.transpose(0, 1).reshape(l, c)
    row = f"{name:10s} nb={nb} Lq={l} Lk={lk} dh={dh:3d} | "
    outs = {}
    for mode in (0x60, 0x70, 0, 0x60, 0x70, 0):       # 0x60: round-3 softmax; 0x70: round-4 softmax, 4 waves; 0: + 8 waves at level 0
        lib.lavie_debug_attention_qt(mode)
        outs[mode] = fn().float().clone()
        us = timeit(fn, iters=30)
        row += f"{ {0x60: 'r3', 0x70: 'r4/4w', 0: 'r4'}[mode]} {us:7.1f} us {4.0 * nb * l * lk * c / us / 1e6:5.0f} TF/s | "
    rl = lambda a, b: float((a - b).norm() / b.norm())
    row += f"r4 vs r3 {rl(outs[0], outs[0x60]):.2e}  vs fp32: r3 {rl(outs[0x60][:l], ref):.2e} r4 {rl(outs[0][:l], ref):.2e}"
    print(row, flush=True)
lib.lavie_debug_attention_qt(0)
nb, l, c = 61, 2560, 320      # sparse-causal (interpolation model): F = 61 frames at L0
qkv = rnd(nb * l, 3 * c)
for mode in (0x60, 0x70, 0, 0x60, 0x70, 0):
    lib.lavie_debug_attention_qt(mode)
    fn = lambda: ops.sparse_causal_attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], nb=nb, frames=61, d=l, heads=8)
    us = timeit(fn, iters=10)
    print(f"sparse-causal L0 F=61 { {0x60: 'r3', 0x70: 'r4/4w', 0: 'r4'}[mode]}: {us:8.1f} us", flush=True)
lib.lavie_debug_attention_qt(0)
