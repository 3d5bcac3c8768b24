"""A/B of the row-resident fused sub-block kernels inside the full UNet forward (bench shape), one process, interleaved
rounds (guide rule 24): lavie_debug_fused_mask 0 = one GEMM per launch, 1 = fused feed-forward, 2 = fused temporal sub-block,
3 = both, 7 = those and the fused text cross-attention sub-block (needs the cached context, as the pipeline runs), 23 = those and the parity form of the
upsample convs (bit 4; bit 3 = conv_shortcut as its own GEMM, measured slower, off)."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from lavie_amd import _lib, spec, weights  # noqa: E402
from lavie_amd.unet import UNet3DConditionModel  # noqa: E402


def main():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    sd = weights.synth_state_dict(spec.param_shapes(), 0)
    net = UNet3DConditionModel(sample_size=64, cross_attention_dim=bench.CTX_DIM, init_weights=False)
    net.load_state_dict({k: v.half() for k, v in sd.items()})
    net = net.to(dev, torch.float16)
    net.prepare(2, bench.FRAMES, bench.LAT_H, bench.LAT_W, bench.CTX_LEN)
    pe, ne, lat = bench.synth_inputs(0, dev)
    ctx = net.cache_context(torch.cat([ne, pe]).half().contiguous())
    x2 = torch.cat([lat, lat]).half().contiguous()
    outs = {}
    for mask in (0, 7, 23):
        lib.lavie_debug_fused_mask(mask)
        outs[mask] = net(x2, 500, encoder_hidden_states=ctx).sample.float()
    for mask in (7, 23):
        d = (outs[mask] - outs[0]).norm() / outs[0].norm()
        print(f"mask {mask} vs 0: rel-L2 {d.item():.2e}", flush=True)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for r in range(4):
        line = f"round {r}:"
        for mask in (0, 7, 23):
            lib.lavie_debug_fused_mask(mask)
            net(x2, 500, encoder_hidden_states=ctx)
            s, e = ev(), ev()
            s.record()
            for _ in range(10):
                net(x2, 500, encoder_hidden_states=ctx)
            e.record()
            torch.cuda.synchronize()
            line += f"  mask {mask}: {s.elapsed_time(e) / 10:7.3f} ms"
        print(line, flush=True)
    lib.lavie_debug_fused_mask(_lib.FUSED_DEFAULT)


if __name__ == "__main__":
    main()
