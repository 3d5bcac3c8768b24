#!/usr/bin/env python3
"""Per-shape timing of the hot operators at the benchmark's real shapes (MI355X).  A/B over GEMM kernel modes
(lavie_debug_force_tile: 0 automatic, 4 never the ping-pong kernel, ...).
Usage: python tools/bench_ops.py [linear|conv|attn|all]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops  # noqa: E402

dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).half()


def bench_linear():
    shapes = []
    for M, C in ((81920, 320), (20480, 640), (5120, 1280), (1280, 1280)):
        shapes += [(M, C, C, "out+res"), (M, 3 * C, C, "qkv"), (M, C, 4 * C, "ff2+res"), (M, 8 * C, C, "geglu")]
    print(f"{'M':>6} {'N':>6} {'K':>5} {'kind':>8} | " + " | ".join(f"mode{m}: us   TF/s   GB/s" for m in (0, 3)))
    for M, N, K, kind in shapes:
        a, w = rnd(M, K), rnd(N, K) / math.sqrt(K)
        bias = torch.randn(N, device=dev)
        res = rnd(M, N) if "res" in kind else None
        geglu = kind == "geglu"
        if geglu:
            w, bias = ops.pack_geglu(w, bias.half())
        out = torch.empty(M, N // 2 if geglu else N, dtype=torch.float16, device=dev)
        row = f"{M:>6} {N:>6} {K:>5} {kind:>8} | "
        for mode in (0, 3):
            _lib.load().lavie_debug_force_tile(mode)
            us = timeit(lambda: ops.linear(a, w, bias=None if kind == "qkv" else bias, residual=res, geglu=geglu, out=out))
            fl = 2.0 * M * N * K
            by = 2.0 * (M * K + N * K + out.numel() + (res.numel() if res is not None else 0))
            row += f"{us:9.1f} {fl / us / 1e6:6.0f} {by / us / 1e3:6.0f} | "
        print(row)
    _lib.load().lavie_debug_force_tile(0)


def bench_conv():
    cases = [(32, 40, 64, 320, 0, 320), (32, 40, 64, 640, 0, 320), (32, 40, 64, 320, 320, 320), (32, 40, 64, 640, 320, 320),
             (32, 20, 32, 640, 0, 640), (32, 20, 32, 640, 640, 640), (32, 20, 32, 1280, 640, 640), (32, 20, 32, 320, 0, 640),
             (32, 10, 16, 1280, 0, 1280), (32, 10, 16, 1280, 1280, 1280), (32, 10, 16, 640, 0, 1280),
             (32, 5, 8, 1280, 0, 1280), (32, 5, 8, 1280, 1280, 1280)]
    print(f"{'NI':>3} {'H':>3} {'W':>3} {'C1':>5} {'C2':>5} {'Cout':>5} | " + " | ".join(f"mode{m}: us   TF/s" for m in (0, 3)))
    for ni, h, w, c1, c2, cout in cases:
        x1 = rnd(ni * h * w, c1)
        x2 = rnd(ni * h * w, c2) if c2 else None
        wt = rnd(cout, c1 + c2, 3, 3) / math.sqrt(9 * (c1 + c2))
        wp = ops.pack_conv3x3(wt)
        bias = torch.randn(cout, device=dev)
        row = f"{ni:>3} {h:>3} {w:>3} {c1:>5} {c2:>5} {cout:>5} | "
        for mode in (0, 3):
            _lib.load().lavie_debug_force_tile(mode)
            us = timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2))
            fl = 2.0 * ni * h * w * cout * 9 * (c1 + c2)
            row += f"{us:9.1f} {fl / us / 1e6:6.0f} | "
        print(row)
    _lib.load().lavie_debug_force_tile(0)


def bench_attn_qt():
    qkv = rnd(32 * 2560, 960)
    for qt in (0, 0x40, 0, 0x40, 0x12, 0x22, 0x32):
        _lib.load().lavie_debug_attention_qt(qt)
        us = timeit(lambda: ops.attention(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], nb=32, lq=2560, lk=2560, heads=8))
        print(f"L0 self-attention QT/ABL={qt:#x}: {us:8.1f} us {4.0 * 32 * 2560 * 2560 * 320 / us / 1e6:6.0f} TF/s")
    _lib.load().lavie_debug_attention_qt(0)


def bench_attn():
    bench_attn_qt()
    print("attention: nb heads L dh | us TF/s")
    for nb, l, c, lk, div in ((32, 2560, 320, 2560, 1), (32, 640, 640, 640, 1), (32, 160, 1280, 160, 1), (32, 2560, 320, 77, 16)):
        if lk == l:
            qkv = rnd(nb * l, 3 * c)
            fn = lambda: ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], nb=nb, lq=l, lk=l, heads=8)
        else:
            q, kv = rnd(nb * l, c), rnd(nb // div * lk, 2 * c)
            fn = lambda: ops.attention(q, kv[:, :c], kv[:, c:], nb=nb, lq=l, lk=lk, heads=8, kv_batch_div=div)
        us = timeit(fn)
        print(f"{nb:>3} 8 {l:>5} x{lk:>5} dh={c // 8:>3} | {us:9.1f} {4.0 * nb * l * lk * c / us / 1e6:6.0f}")
    print("temporal: B F D C | us GB/s streaming kernel / tile kernel at LDS budget 65K / 33K / 17K")
    for d, c in ((2560, 320), (640, 640), (160, 1280)):
        qkv = rnd(2 * 16 * d, 3 * c)
        bias = torch.randn(8, 16, 16, device=dev)
        cos, sin = ops.rotary_tables(16, 32)
        row = f"  2 16 {d:>5} {c:>5} | "
        for budget in (0, 66560, 33000, 17000):
            _lib.load().lavie_debug_temporal_budget(budget)
            us = timeit(lambda: ops.temporal_attention(qkv, 2, 16, d, 8, bias, cos, sin))
            row += f"{us:8.1f} us {4.0 * 2 * 16 * d * c * 2 / us / 1e3:6.0f} GB/s | "
        print(row)
    _lib.load().lavie_debug_temporal_budget(0)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("linear", "all"):
        bench_linear()
    if what in ("conv", "all"):
        bench_conv()
    if what in ("attn", "all"):
        bench_attn()


def bench_ablate():
    """Diagnostic: which pipeline paces the conv K loop (results are wrong in ablated modes)."""
    for ni, h, w, c1, c2, cout in ((32, 40, 64, 640, 320, 320), (32, 10, 16, 1280, 0, 1280), (32, 5, 8, 1280, 0, 1280)):
        x1 = rnd(ni * h * w, c1)
        x2 = rnd(ni * h * w, c2) if c2 else None
        wp = ops.pack_conv3x3(rnd(cout, c1 + c2, 3, 3) / math.sqrt(9 * (c1 + c2)))
        bias = torch.randn(cout, device=dev)
        row = f"{ni} {h}x{w} {c1}+{c2}->{cout} | "
        for mode, name in ((1, "full"), (0x11, "noMFMA"), (0x21, "noLoads"), (0x31, "loadsOnly")):
            _lib.load().lavie_debug_force_tile(mode)
            us = timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2))
            row += f"{name} {us:8.1f} us | "
        print(row)
    _lib.load().lavie_debug_force_tile(0)


def bench_ablate_pp():
    """Which phase paces the 160x320 ping-pong kernel (results are wrong in ablated modes)."""
    for ni, h, w, c1, c2, cout in ((32, 40, 64, 640, 320, 320), (32, 20, 32, 640, 0, 640), (32, 40, 64, 320, 0, 320)):
        x1 = rnd(ni * h * w, c1)
        x2 = rnd(ni * h * w, c2) if c2 else None
        wp = ops.pack_conv3x3(rnd(cout, c1 + c2, 3, 3) / math.sqrt(9 * (c1 + c2)))
        bias = torch.randn(cout, device=dev)
        row = f"pp {ni} {h}x{w} {c1}+{c2}->{cout} | "
        for rnd_ in range(2):
            for mode, name in ((0x03, "full"), (0x43, "noSetprio"), (0x13, "noMFMA"), (0x23, "noDMA")):
                _lib.load().lavie_debug_force_tile(mode)
                us = timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2), iters=30)
                row += f"{name} {us:8.1f} us | "
        print(row)
    _lib.load().lavie_debug_force_tile(0)


def bench_pp_splits():
    """Split-K sweep of the 160x320 ping-pong kernel on the shapes whose grid does not fill the chip, against the
    automatic choice of the 128-row kernel."""
    lib = _lib.load()
    print("conv: shape | auto(128-row) | pp at S=1,2,3,4,6,8")
    for ni, h, w, c1, c2, cout in ((32, 10, 16, 1280, 0, 1280), (32, 10, 16, 1280, 1280, 1280), (32, 10, 16, 640, 0, 1280),
                                   (32, 5, 8, 1280, 0, 1280), (32, 5, 8, 1280, 1280, 1280), (32, 20, 32, 1280, 640, 640),
                                   (8, 20, 32, 1280, 0, 1280)):
        x1 = rnd(ni * h * w, c1)
        x2 = rnd(ni * h * w, c2) if c2 else None
        wp = ops.pack_conv3x3(rnd(cout, c1 + c2, 3, 3) / math.sqrt(9 * (c1 + c2)))
        bias = torch.randn(cout, device=dev)
        lib.lavie_debug_force_tile(0); lib.lavie_debug_force_splits(0)
        row = f"{ni} {h}x{w} {c1}+{c2}->{cout} | {timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2)):7.1f} | "
        lib.lavie_debug_force_tile(3)
        for s_ in (1, 2, 3, 4, 6, 8):
            lib.lavie_debug_force_splits(s_)
            row += f"{timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2)):7.1f} "
        print(row)
    print("linear: M N K | auto(128-row) | pp at S=1,2,3,4,6,8")
    for M, N, K in ((5120, 1280, 1280), (5120, 1280, 5120), (5120, 3840, 1280), (20480, 640, 640), (20480, 1920, 640),
                    (1280, 1280, 5120), (1280, 3840, 1280), (81920, 960, 320)):
        a, w, r = rnd(M, K), rnd(N, K) / math.sqrt(K), rnd(M, N)
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        lib.lavie_debug_force_tile(0); lib.lavie_debug_force_splits(0)
        row = f"{M} {N} {K} | {timeit(lambda: ops.linear(a, w, bias=bias, residual=r, out=out)):7.1f} | "
        lib.lavie_debug_force_tile(3)
        for s_ in (1, 2, 3, 4, 6, 8):
            lib.lavie_debug_force_splits(s_)
            row += f"{timeit(lambda: ops.linear(a, w, bias=bias, residual=r, out=out)):7.1f} "
        print(row)
    lib.lavie_debug_force_tile(0); lib.lavie_debug_force_splits(0)


def bench_ablate_gemm():
    for M, N, K in ((20480, 640, 2560), (81920, 320, 1280), (5120, 1280, 5120)):
        a, w = rnd(M, K), rnd(N, K) / math.sqrt(K)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        for tile, tname in ((1, "128x160 2-stage"),):
            row = f"{M}x{N}x{K} {tname:24s} | "
            for abl, name in ((0, "full"), (1, "noMFMA"), (2, "noLoads"), (3, "loadsOnly")):
                _lib.load().lavie_debug_force_tile(abl * 16 + tile)
                us = timeit(lambda: ops.linear(a, w, out=out))
                row += f"{name} {us:8.1f} us | "
            print(row)
    _lib.load().lavie_debug_force_tile(0)


def bench_splits():
    """Split-K sweep on the under-filled conv / linear grids."""
    lib = _lib.load()
    print("conv: shape | us at S=1,2,3,4,5,6,8")
    for ni, h, w, c1, c2, cout in ((32, 20, 32, 640, 0, 640), (32, 20, 32, 1280, 640, 640), (32, 10, 16, 1280, 0, 1280),
                                   (32, 10, 16, 1280, 1280, 1280), (32, 5, 8, 1280, 0, 1280), (32, 5, 8, 1280, 1280, 1280),
                                   (32, 40, 64, 320, 0, 320)):
        x1 = rnd(ni * h * w, c1)
        x2 = rnd(ni * h * w, c2) if c2 else None
        wp = ops.pack_conv3x3(rnd(cout, c1 + c2, 3, 3) / math.sqrt(9 * (c1 + c2)))
        bias = torch.randn(cout, device=dev)
        row = f"{ni} {h}x{w} {c1}+{c2}->{cout} | "
        for s in (1, 2, 3, 4, 5, 6, 8):
            lib.lavie_debug_force_splits(s)
            row += f"{timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2)):7.1f} "
        lib.lavie_debug_force_splits(0)
        row += f"| auto {timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2)):7.1f}"
        print(row)
    print("linear: M N K | us at S=1,2,3,4,6,8")
    for M, N, K in ((20480, 640, 640), (20480, 640, 2560), (5120, 1280, 1280), (5120, 1280, 5120), (5120, 3840, 1280),
                    (1280, 1280, 1280), (1280, 1280, 5120), (1280, 3840, 1280)):
        a, w, r = rnd(M, K), rnd(N, K) / math.sqrt(K), rnd(M, N)
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        row = f"{M} {N} {K} | "
        for s in (1, 2, 3, 4, 6, 8):
            lib.lavie_debug_force_splits(s)
            row += f"{timeit(lambda: ops.linear(a, w, bias=bias, residual=r, out=out)):7.1f} "
        lib.lavie_debug_force_splits(0)
        row += f"| auto {timeit(lambda: ops.linear(a, w, bias=bias, residual=r, out=out)):7.1f}"
        print(row)


def bench_order():
    lib = _lib.load()
    lib.lavie_debug_force_splits(1)
    print("conv K order: shape | slab>tap us | tap>slab us")
    for ni, h, w, c1, c2, cout in ((32, 40, 64, 320, 0, 320), (32, 40, 64, 640, 320, 320), (32, 20, 32, 640, 0, 640),
                                   (32, 20, 32, 1280, 640, 640), (32, 10, 16, 1280, 0, 1280), (32, 5, 8, 1280, 0, 1280)):
        x1 = rnd(ni * h * w, c1)
        x2 = rnd(ni * h * w, c2) if c2 else None
        wt = rnd(cout, c1 + c2, 3, 3) / math.sqrt(9 * (c1 + c2))
        bias = torch.randn(cout, device=dev)
        row = f"{ni} {h}x{w} {c1}+{c2}->{cout} | "
        outs = []
        for tm in (0, 1):
            lib.lavie_debug_conv_tap_major(tm)
            wp = ops.pack_conv3x3(wt)
            row += f"{timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2)):8.1f} | "
            outs.append(ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2).float())
        row += f"max diff {float((outs[0] - outs[1]).abs().max()):.3g}"
        print(row)
    lib.lavie_debug_conv_tap_major(0)
    lib.lavie_debug_force_splits(0)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "order":
    bench_order()

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "splits":
    bench_splits()

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "pp_splits":
    bench_pp_splits()

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ablate_pp":
    bench_ablate_pp()

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ablate":
    bench_ablate()
    bench_ablate_gemm()


def bench_tconv():
    """(T,1,1) temporal convs of the VSR stage at its full size: 8-frame chunk at 320x512 (vsr/sample.py chunking), widths of
    the VSR UNet levels."""
    print(f"{'B':>2} {'F':>3} {'D':>7} {'C':>5} {'taps':>4} | us   TF/s   GB/s(algorithmic: x + y once)")
    for F_, D, C, taps in ((8, 320 * 512, 256, 5), (8, 320 * 512, 256, 3), (8, 160 * 256, 512, 5), (8, 80 * 128, 512, 5),
                           (8, 40 * 64, 1024, 5)):
        x = rnd(F_ * D, C)
        w = rnd(C, C, taps, 1, 1) / math.sqrt(C * taps)
        wp = ops.pack_temporal_conv(w)
        bias = torch.randn(C, device=dev)
        fl = 2.0 * F_ * D * C * C * taps
        by = 2.0 * 2 * F_ * D * C
        row = f"{1:>2} {F_:>3} {D:>7} {C:>5} {taps:>4} | "
        for mode in (0, 3):                     # automatic choice (the halo-patch kernel's temporal mode) / ping-pong kernel forced
            _lib.load().lavie_debug_force_tile(mode)
            us = timeit(lambda: ops.temporal_conv(x, wp, bias, 1, F_, D, taps), iters=10)
            row += f"mode {mode}: {us:9.1f} {fl / us / 1e6:6.0f} {by / us / 1e3:6.0f} | "
        _lib.load().lavie_debug_force_tile(0)
        print(row)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "tconv":
    bench_tconv()
