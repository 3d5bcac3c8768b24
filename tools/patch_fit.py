#!/usr/bin/env python3
"""Where the halo-patch conv kernel's time goes (round 4): launches of the SAME tile grid with K loops of different length
(input channels 320 .. 1280 -> 45 .. 180 K-tiles) at the three levels it runs at, timed in isolation, and the least-squares line
t = a + b * K-tiles per workgroup.  b = microseconds per K-tile in the loop (6.55 MFLOP per K-tile and CU -> the loop's own share
of the MFMA peak); a = everything outside the loop (tables, first patch, epilogue, dispatch) per workgroup round."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops
from tools.bench_ops import rnd, timeit

lib = _lib.load()
lib.lavie_debug_force_tile(5)
levels = [("L0 32x40x64 -> 320 (512 tiles: 2 per CU)", 32, 40, 64, 320, 2, (320, 640, 960, 1280)),
          ("L1 32x20x32 -> 640 (256 tiles: 1 per CU)", 32, 20, 32, 640, 1, (320, 640, 1280, 1920)),
          ("L2 32x10x16 -> 1280 (128 tiles x 2 splits)", 32, 10, 16, 1280, 1, (1280, 2560))]
for name, ni, h, w, cout, per_cu, cins in levels:
    pts = []
    for cin in cins:
        x = rnd(ni * h * w, cin)
        wp = ops.pack_conv3x3(rnd(cout, cin, 3, 3) / math.sqrt(9 * cin))
        bias = torch.randn(cout, device="cuda")
        fn = lambda: ops.conv3x3(x, wp, bias, ni, h, w)
        us = min(timeit(fn, iters=30) for _ in range(3))
        nk = 9 * cin // 64
        splits = 2 if "splits" in name else 1
        pts.append((nk * per_cu / splits, us, 2.0 * ni * h * w * cout * 9 * cin / us / 1e6))
    n = len(pts)
    sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts); sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    a = (sy - b * sx) / n
    print(f"{name}: " + "  ".join(f"{int(k)} K-tiles/CU {us:6.1f} us ({tf:4.0f} TF/s)" for k, us, tf in pts), flush=True)
    print(f"    fit: {b:.3f} us per K-tile = {6.5536 / b / 9.766:.3f} of the 2.5 PFLOP/s peak inside the loop; {a:5.1f} us per launch outside it"
          f" ({a / per_cu:.1f} us per workgroup round)", flush=True)
lib.lavie_debug_force_tile(0)
