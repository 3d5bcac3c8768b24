"""Development aid: the plain GEMM shapes of levels 1 - 3 under every kernel-choice mode (lavie_debug_force_tile / _force_splits), isolated
launches with bias + residual, best of three timing rounds.  Which kernel should the planner pick where the grid under-fills the chip?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavie_amd import _lib, ops
from tools.bench_ops import rnd, timeit

lib = _lib.load()
shapes = [(5120, 1280, 1280, "L2 proj"), (5120, 3840, 1280, "L2 qkv"), (5120, 1280, 5120, "L2 ff2"), (1280, 1280, 1280, "mid proj"),
          (1280, 3840, 1280, "mid qkv"), (20480, 640, 640, "L1 proj"), (20480, 1920, 640, "L1 qkv"), (20480, 640, 2560, "L1 ff2")]
modes = [("auto", 0, 0), ("128-row widest", 1, 0), ("pp", 3, 0), ("pp split 2", 3, 2), ("ppx", 7, 0), ("auto split 2", 0, 2), ("auto split 3", 0, 3),
         ("no pp", 4, 0)]
for M, N, K, name in shapes:
    a, w, r = rnd(M, K), rnd(N, K) / K ** 0.5, rnd(M, N)
    b = torch.randn(N, device="cuda")
    out = torch.empty_like(r)
    row = f"{name:9s} {M:6d} x {N:5d} x {K:5d} ({2.0 * M * N * K / 1e9:6.1f} GFLOP) |"
    ref = None
    for label, mode, splits in modes:
        lib.lavie_debug_force_tile(mode)
        lib.lavie_debug_force_splits(splits)
        try:
            fn = lambda: ops.linear(a, w, bias=b, residual=r, out=out)
            fn()
            us = min(timeit(fn, iters=20) for _ in range(3))
            if ref is None:
                ref = out.clone()
            ok = "" if torch.equal(out, ref) or splits else " (!= auto)"
            row += f" {label} {us:6.1f}{ok} |"
        except RuntimeError as e:
            row += f" {label} n/a |"
    print(row, flush=True)
lib.lavie_debug_force_tile(0)
lib.lavie_debug_force_splits(0)
