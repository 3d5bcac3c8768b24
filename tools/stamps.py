#!/usr/bin/env python3
"""Phase shares of the halo-patch conv kernel from its in-kernel stamp build (MI355X)."""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops
from tools.bench_ops import rnd

lib = _lib.load()
names = {1: "R(t,0)", 2: "bar", 3: "M(t,0)", 4: "bar", 6: "R(t,1)", 7: "bar", 8: "M(t,1)+addr", 9: "dma wait", 10: "bar"}
modes = [int(m, 0) for m in sys.argv[1:]] or [0x75]
for mode in modes:
  for ni, h, w, c1, cout in ((32, 20, 32, 640, 640),):
    x1 = rnd(ni * h * w, c1)
    wp = ops.pack_conv3x3(rnd(cout, c1, 3, 3) / math.sqrt(9 * c1))
    bias = torch.randn(cout, device="cuda")
    lib.lavie_debug_force_tile(mode)
    for _ in range(3):
        ops.conv3x3(x1, wp, bias, ni, h, w)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    _lib.check(lib.lavie_debug_patch_stamps(ctypes.cast(buf, ctypes.c_void_p)))
    lib.lavie_debug_force_tile(0)
    print(f"mode {mode:#x} conv {ni} {h}x{w} {c1}->{cout}: cycles per K-tile by segment (wave: " + " ".join(f"{names[i]:>11s}" for i in sorted(names)) + " | total)")
    for wv in range(8):
        nk = buf[wv * 16 + 11] or 1
        vals = [buf[wv * 16 + i] / nk for i in sorted(names)]
        print(f"  wave {wv}: " + " ".join(f"{v:11.0f}" for v in vals) + f" | {sum(vals):7.0f}")
