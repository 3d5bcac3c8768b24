"""A/B of the producer-side GroupNorm statistics (lavie_debug_fused_mask bit 5) inside the full UNet forward at the bench shape
(cached context, shared CFG prefix as the guided loop runs), one process, interleaved rounds; also the difference of the outputs."""
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
    net.set_cfg_shared_input(True)
    masks = (_lib.FUSED_DEFAULT & ~32, _lib.FUSED_DEFAULT)
    outs = {}
    for mask in masks:
        lib.lavie_debug_fused_mask(mask)
        outs[mask] = net(x2, 500, encoder_hidden_states=ctx).sample.float()
    d = (outs[masks[1]] - outs[masks[0]]).norm() / outs[masks[0]].norm()
    print(f"producer statistics vs statistics pass: rel-L2 {d.item():.2e}", flush=True)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for r in range(4):
        line = f"round {r}:"
        for mask in masks:
            lib.lavie_debug_fused_mask(mask)
            net(x2, 500, encoder_hidden_states=ctx)
            s, e = ev(), ev()
            s.record()
            for _ in range(10):
                net(x2, 500, encoder_hidden_states=ctx)
            e.record()
            torch.cuda.synchronize()
            line += f"  {'stats pass' if mask == masks[0] else 'from producers'}: {s.elapsed_time(e) / 10:7.3f} ms"
        print(line, flush=True)
    for mask in masks:       # per-class device time, one instrumented forward each
        lib.lavie_debug_fused_mask(mask)
        net(x2, 500, encoder_hidden_states=ctx)
        bench.profile_begin(lib, 0x7FF, 4096)
        net(x2, 500, encoder_hidden_states=ctx)
        rows = bench.profile_end(lib)
        print(("stats pass    " if mask == masks[0] else "from producers"), " ".join(f"{r['name'].split('_')[0]}={r['ms']:.3f}({r['launches']})" for r in rows if r["launches"]), flush=True)
    lib.lavie_debug_fused_mask(_lib.FUSED_DEFAULT)
    net.set_cfg_shared_input(False)


if __name__ == "__main__":
    main()
