"""Halo-patch kernel (lavie_debug_force_tile 5) against the 160x320 ping-pong kernel (3) on the plain 3x3 conv shapes of levels 0-2
of the bench (single source and channel-concatenated inputs), interleaved rounds."""
import math
import sys

import torch

sys.path.insert(0, ".")
from lavie_amd import _lib, ops  # noqa: E402


def timeit(fn, n=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


def main():
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    for ni, h, w, c1, c2, cout in ((32, 40, 64, 320, 0, 320), (32, 40, 64, 640, 320, 320), (32, 20, 32, 640, 0, 640), (32, 20, 32, 1280, 640, 640),
                                   (32, 20, 32, 320, 0, 640), (32, 10, 16, 1280, 0, 1280), (32, 10, 16, 1280, 1280, 1280), (32, 10, 16, 640, 0, 1280)):
        x1 = torch.randn(ni * h * w, c1, generator=g).half().cuda()
        x2 = torch.randn(ni * h * w, c2, generator=g).half().cuda() if c2 else None
        wp = ops.pack_conv3x3((torch.randn(cout, c1 + c2, 3, 3, generator=g) / math.sqrt(9 * (c1 + c2))).half().cuda())
        bias = torch.randn(cout, generator=g).cuda()
        line = f"conv {ni}x{h}x{w} {c1}+{c2}->{cout}:"
        for r in range(2):
            for mode in (5, 3, 0):
                lib.lavie_debug_force_tile(mode)
                t = timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2))
                line += f"  [{mode}] {t:7.1f}"
        lib.lavie_debug_force_tile(0)
        print(line + "  us", flush=True)


if __name__ == "__main__":
    main()
