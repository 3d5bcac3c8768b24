"""Temporal-attention kernel alone at the three base-UNet levels (run through gpurun): the streaming kernel (default) against
the tile kernel at its LDS budgets.  GB/s = algorithmic bytes (q | k | v read once, o written once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops

dev = "cuda"


def timeit(fn, iters=30):
    for _ in range(5):
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
    lib = _lib.load()
    print("B F D C | streaming | tile kernel at 65K / 33K / 17K (F <= 16) or 135K / 70K / 36K (F > 16)   (us, GB/s)")
    for f, d, c in ((16, 2560, 320), (16, 640, 640), (16, 160, 1280), (16, 40, 1280), (8, 2560, 320),
                    (61, 2560, 320), (61, 640, 640), (61, 160, 1280),
                    (8, 163840, 256), (8, 40960, 512), (8, 10240, 1024)):       # the VSR stage: 8-frame chunks at 320x512 / 160x256 / 80x128
        nb = 2 if f * d * c < 2e8 else 1
        qkv = torch.randn(nb * f * d, 3 * c, device=dev, dtype=torch.float16)
        bias = torch.randn(8, f, f, device=dev)
        cos, sin = ops.rotary_tables(f, 32)
        row = f"{nb} {f:>2} {d:>6} {c:>5} | "
        for budget in ((0, 66560, 33000, 17000) if f <= 16 else (0, 135000, 70000, 36000)):
            lib.lavie_debug_temporal_budget(budget)
            us = timeit(lambda: ops.temporal_attention(qkv, nb, f, d, 8, bias, cos, sin))
            row += f"{us:7.1f} {4.0 * nb * f * d * c * 2 / us / 1e3:6.0f} | "
        print(row)
    lib.lavie_debug_temporal_budget(0)


if __name__ == "__main__":
    main()
