"""A/B of the halo-patch conv kernel's two K-loop builds (lavie_debug_force_tile 5 = software-pipelined, shipped; 0xC5 = ping-pong groups)
on the bench's conv shapes, then the whole UNet forward with the kernel choice left automatic (0 vs 0xC0).  Interleaved rounds."""
import math
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from lavie_amd import _lib, ops, spec, weights  # noqa: E402
from lavie_amd.unet import UNet3DConditionModel  # noqa: E402


def timeit(fn, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
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
    for ni, h, w, c1, c2, cout in ((32, 40, 64, 320, 0, 320), (32, 40, 64, 320, 320, 320), (32, 20, 32, 640, 0, 640), (32, 20, 32, 640, 640, 640),
                                   (32, 10, 16, 1280, 0, 1280), (32, 10, 16, 1280, 1280, 1280)):
        x1 = torch.randn(ni * h * w, c1, generator=g).half().cuda()
        x2 = torch.randn(ni * h * w, c2, generator=g).half().cuda() if c2 else None
        wp = ops.pack_conv3x3((torch.randn(cout, c1 + c2, 3, 3, generator=g) / math.sqrt(9 * (c1 + c2))).half().cuda())
        bias = torch.randn(cout, generator=g).cuda()
        flop = 2.0 * ni * h * w * cout * 9 * (c1 + c2)
        line = f"conv {ni}x{h}x{w} {c1}+{c2}->{cout}:"
        outs = {}
        for r in range(2):
            for mode in (5, 0xC5):
                lib.lavie_debug_force_tile(mode)
                t = timeit(lambda: ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2))
                outs[mode] = ops.conv3x3(x1, wp, bias, ni, h, w, x2=x2)
                line += f"  [{mode:#x}] {t:7.1f} us ({flop / t / 1e6:6.0f} TF/s)"
        lib.lavie_debug_force_tile(0)
        d = (outs[5].float() - outs[0xC5].float()).norm() / outs[5].float().norm()
        print(line + f"   rel diff {d.item():.1e}", flush=True)

    dev = torch.device("cuda", 0)
    sd = weights.synth_state_dict(spec.param_shapes(), 0)
    net = UNet3DConditionModel(sample_size=64, cross_attention_dim=bench.CTX_DIM, init_weights=False)
    net.load_state_dict({k: v.half() for k, v in sd.items()})
    net = net.to(dev, torch.float16)
    net.prepare(2, bench.FRAMES, bench.LAT_H, bench.LAT_W, bench.CTX_LEN)
    pe, ne, lat = bench.synth_inputs(0, dev)
    ctx = net.cache_context(torch.cat([ne, pe]).half().contiguous())
    x2 = torch.cat([lat, lat]).half().contiguous()
    for r in range(4):
        line = f"forward round {r}:"
        for mode in (0, 0xC0):
            lib.lavie_debug_force_tile(mode)
            t = timeit(lambda: net(x2, 500, encoder_hidden_states=ctx))
            line += f"  [{mode:#x}] {t / 1e3:7.3f} ms"
        print(line, flush=True)
    lib.lavie_debug_force_tile(0)


if __name__ == "__main__":
    main()
