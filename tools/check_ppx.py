#!/usr/bin/env python3
"""Persistent ping-pong GEMM (igemm_ppx.hip, mode 7) vs the automatic choice (mode 8 = never ppx) and the one-tile-per-
workgroup ping-pong kernel (mode 3) on the model's plain-GEMM shapes: max abs difference (must be 0) and time (MI355X).
Each shape runs on buffers rotated through > 256 MiB so that operands come from HBM as they do inside the UNet."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops
from tools.bench_ops import rnd

lib = _lib.load()
modes = [int(m, 0) for m in sys.argv[1:]] or [8, 3, 7]


def timeit_rot(fns, iters=24, warm=4):
    n = len(fns)
    for i in range(warm):
        fns[i % n]()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fns[i % n]()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


shapes = []
for M, C in [tuple(int(v) for v in t.split("x")) for t in os.environ.get("PPX_SHAPES", "81920x320,20480x640,5120x1280,1280x1280").split(",")]:
    shapes += [(M, C, C, "out+res"), (M, C, C, "plain"), (M, 3 * C, C, "qkv"), (M, C, 4 * C, "ff2+res"), (M, 8 * C, C, "geglu")]
print(f"{'M':>6} {'N':>6} {'K':>5} {'kind':>8} | " + " | ".join(f"mode{m:#x}:  us  TF/s  GB/s diff" for m in modes))
for M, N, K, kind in shapes:
    geglu = kind == "geglu"
    nout = N // 2 if geglu else N
    per = 2 * (M * K + nout * M + (M * N if "res" in kind else 0))
    nset = max(2, min(8, int(300e6 // per) + 1))
    sets = []
    w = rnd(N, K) / math.sqrt(K)
    bias = torch.randn(N, device="cuda")
    if geglu:
        w, bias = ops.pack_geglu(w, bias.half())
    for _ in range(nset):
        sets.append((rnd(M, K), rnd(M, N) if "res" in kind else None, torch.empty(M, nout, dtype=torch.float16, device="cuda")))
    row = f"{M:>6} {N:>6} {K:>5} {kind:>8} | "
    ref = None
    for m in modes:
        lib.lavie_debug_force_tile(m)
        fns = [(lambda s=s: ops.linear(s[0], w, bias=None if kind in ("qkv",) else bias, residual=s[1], geglu=geglu, out=s[2])) for s in sets]
        us = timeit_rot(fns)
        fns[0]()
        out = sets[0][2].float().clone()
        if ref is None:
            ref = out
        fl = 2.0 * M * N * K
        by = 2.0 * (M * K + N * K + M * nout + (M * N if "res" in kind else 0))
        row += f"{us:8.1f} {fl / us / 1e6:5.0f} {by / us / 1e3:5.0f} {float((out - ref).abs().max()):.2g} | "
    print(row, flush=True)
lib.lavie_debug_force_tile(0)
