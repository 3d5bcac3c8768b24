#!/usr/bin/env python3
"""Phase shares of the persistent ping-pong GEMM from its in-kernel stamp build (mode 0x37, MI355X): cycles per step by
segment, separately for ordinary K-tile steps and for the steps that carry a deferred epilogue."""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops
from tools.bench_ops import rnd

lib = _lib.load()
names = ["dma", "epilogue", "R0", "bar", "M0", "bar", "R1", "bar", "M1+wait", "bar"]
for M, N, K, res in ((81920, 960, 320, False), (81920, 320, 320, True), (20480, 640, 640, True), (20480, 1920, 640, False)):
    a, w = rnd(M, K), rnd(N, K) / math.sqrt(K)
    r = rnd(M, N) if res else None
    bias = torch.randn(N, device="cuda")
    lib.lavie_debug_force_tile(0x37)
    for _ in range(3):
        ops.linear(a, w, bias=bias, residual=r)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 256)()
    _lib.check(lib.lavie_debug_ppx_stamps(ctypes.cast(buf, ctypes.c_void_p)))
    lib.lavie_debug_force_tile(0)
    print(f"GEMM {M}x{N}x{K} res={res}: cycles per step (wave: " + " ".join(f"{n:>8s}" for n in names) + " | total)   [ordinary steps / epilogue steps]")
    for wv in (0, 1, 4, 5):
        nk, ne = buf[wv * 32 + 10] or 1, buf[wv * 32 + 11] or 1
        o = [buf[wv * 32 + i] / nk for i in range(10)]
        e = [buf[wv * 32 + 16 + i] / ne for i in range(10)]
        print(f"  wave {wv} ord ({nk:3d}): " + " ".join(f"{v:8.0f}" for v in o) + f" | {sum(o):7.0f}")
        print(f"  wave {wv} epi ({ne:3d}): " + " ".join(f"{v:8.0f}" for v in e) + f" | {sum(e):7.0f}   loop total {buf[wv * 32 + 12]}")
