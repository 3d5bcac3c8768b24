#!/usr/bin/env python3
"""Does a conv hold its burst rate when it runs back to back for seconds?  (DVFS check: MI355X_MICROARCH 'DVFS give-back')"""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavie_amd import _lib, ops
from tools.bench_ops import rnd, timeit

ni, h, w, c1, c2, cout = 32, 20, 32, 640, 0, 640
x1 = rnd(ni * h * w, c1)
wp = ops.pack_conv3x3(rnd(cout, c1 + c2, 3, 3) / math.sqrt(9 * (c1 + c2)))
bias = torch.randn(cout, device="cuda")
fl = 2.0 * ni * h * w * cout * 9 * (c1 + c2)
for it in (20, 200, 2000, 10000, 20, 20000):
    us = timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w), iters=it, warm=3)
    print(f"iters {it:6d}: {us:7.1f} us {fl / us / 1e6:6.0f} TF/s", flush=True)
    time.sleep(0.5 if it != 20 else 0.0)
