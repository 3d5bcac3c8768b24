"""Where does the fused-shortcut conv2 of an up-block resnet spend its time?  Level-0 shape (32 frames of 40x64, 320 channels,
shortcut over 640 + 320 concatenated channels): 9-tap part alone on the ping-pong kernel / on the patch kernel, the fused conv,
and the 1x1 shortcut alone as a plain GEMM on a materialised [M, 960] tensor."""
import math
import sys

import torch

sys.path.insert(0, ".")
from lavie_amd import _lib, ops  # noqa: E402


def timeit(fn, n=10):
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
    for ni, h, w, c, c1, c2 in ((32, 40, 64, 320, 640, 320), (32, 20, 32, 640, 1280, 640)):
        M = ni * h * w
        x = torch.randn(M, c, generator=g).half().cuda()
        s1 = torch.randn(M, c1, generator=g).half().cuda()
        s2 = torch.randn(M, c2, generator=g).half().cuda()
        cat = torch.cat([s1, s2], 1).contiguous()
        w3 = (torch.randn(c, c, 3, 3, generator=g) / math.sqrt(9 * c)).half().cuda()
        wsc = (torch.randn(c, c1 + c2, 1, 1, generator=g) / math.sqrt(c1 + c2)).half().cuda()
        wp9 = ops.pack_conv3x3(w3)
        wpf = ops.pack_conv3x3(w3, wsc)
        wl = wsc.reshape(c, c1 + c2).contiguous()
        bias = torch.randn(c, generator=g).cuda()
        rows = {}
        lib.lavie_debug_force_tile(3)
        rows["9-tap part, ping-pong kernel"] = timeit(lambda: ops.conv3x3(x, wp9, bias, ni, h, w))
        lib.lavie_debug_force_tile(0)
        rows["9-tap part, automatic (patch kernel)"] = timeit(lambda: ops.conv3x3(x, wp9, bias, ni, h, w))
        rows["fused conv + shortcut (automatic)"] = timeit(lambda: ops.conv3x3(x, wpf, bias, ni, h, w, sc1=s1, sc2=s2))
        rows["shortcut alone, plain GEMM on [M, C1 + C2]"] = timeit(lambda: ops.linear(cat, wl))
        print(f"{ni}x{h}x{w}, {c} channels, shortcut over {c1}+{c2}:")
        for k, v in rows.items():
            print(f"   {k:48s} {v:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
