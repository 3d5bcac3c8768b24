#!/usr/bin/env python3
"""Isolates what the sparse-causal attention costs: plain attention at Lk = D and 2D, small and large batch, vs the
sparse-causal kernel on the same fused-QKV layout.  Usage: python tools/bench_attn_sc.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import ops  # noqa: E402


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    heads = 8
    scale = float(os.environ.get("ATTN_INPUT_SCALE", "0.5"))
    for C, D in ((320, 2560), (640, 640)):
        dh = C // heads
        for nb, frames in ((32, 16), (122, 61)):
            qkv = (torch.randn(nb * D, 3 * C, device="cuda") * scale).half()
            q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
            kv2 = (torch.randn(nb * 2 * D, 2 * C, device="cuda") * 0.5).half()
            t1 = timeit(lambda: ops.attention(q, k, v, nb, D, D, heads))
            t2 = timeit(lambda: ops.attention(q, kv2[:, :C], kv2[:, C:], nb, D, 2 * D, heads))
            t3 = timeit(lambda: ops.sparse_causal_attention(q, k, v, nb, frames, D, heads))
            fl = 4.0 * nb * heads * D * D * dh
            print(f"scale={scale} C={C} D={D} NB={nb}: plain Lk=D {t1:8.1f} us {fl / t1 / 1e6:6.0f} TF/s | plain Lk=2D {t2:8.1f} us "
                  f"{2 * fl / t2 / 1e6:6.0f} TF/s | sparse-causal {t3:8.1f} us {2 * fl / t3 / 1e6:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
