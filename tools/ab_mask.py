"""A/B of lavie_debug_fused_mask settings inside the full UNet forward at the bench shape (cached context, shared CFG prefix, as the
guided loop runs it), one process, interleaved rounds (guide rule 24): per-forward wall time, output difference against the first
mask, and the per-class device time of one instrumented forward each.
Usage: python tools/ab_mask.py 0xB7 0x37 [...]      (bits: include/lavie_hip.h)"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from lavie_amd import _lib, spec, weights  # noqa: E402
from lavie_amd.unet import UNet3DConditionModel  # noqa: E402


def main():
    masks = [int(a, 0) for a in sys.argv[1:]] or [_lib.FUSED_DEFAULT, _lib.FUSED_DEFAULT & ~0x80]
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
    outs = {}
    for mask in masks:
        lib.lavie_debug_fused_mask(mask)
        outs[mask] = net(x2, 500, encoder_hidden_states=ctx).sample.float()
    for mask in masks[1:]:
        d = (outs[mask] - outs[masks[0]]).norm() / outs[masks[0]].norm()
        print(f"mask {mask:#x} vs {masks[0]:#x}: rel-L2 {d.item():.2e}", flush=True)
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
            line += f"  {mask:#x}: {s.elapsed_time(e) / 10:7.3f} ms"
        print(line, flush=True)
    for mask in masks:
        lib.lavie_debug_fused_mask(mask)
        net(x2, 500, encoder_hidden_states=ctx)
        bench.profile_begin(lib, 0x7FF, 4096)
        net(x2, 500, encoder_hidden_states=ctx)
        rows = bench.profile_end(lib)
        print(f"{mask:#x}", " ".join(f"{r['name'].split('_')[0]}={r['ms']:.3f}({r['launches']})" for r in rows if r["launches"]), flush=True)
    lib.lavie_debug_fused_mask(_lib.FUSED_DEFAULT)
    net.set_cfg_shared_input(False)


if __name__ == "__main__":
    main()
