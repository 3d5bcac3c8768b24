"""A/B of lavie_debug_force_tile modes inside the full UNet forward at the bench shape (cached context, shared CFG prefix), one
process, interleaved rounds.  Usage: python tools/ab_tile.py 0 9 9 0
(give the modes in A B B A order: the second forward batch of a pair runs ~0.1 ms faster than the first on this pool, which an A B order
books to B — profiles/r04_ab_persistent_kernel_at_one_tile_per_cu_rejected.txt)"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from lavie_amd import _lib, spec, weights  # noqa: E402
from lavie_amd.unet import UNet3DConditionModel  # noqa: E402


def main():
    modes = [int(a, 0) for a in sys.argv[1:]] or [0, 9]
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    sd = weights.synth_state_dict(spec.param_shapes(), 0)
    net = UNet3DConditionModel(sample_size=64, cross_attention_dim=bench.CTX_DIM, init_weights=False)
    net.load_state_dict({k: v.half() for k, v in sd.items()})
    net = net.to(dev, torch.float16)
    pe, ne, lat = bench.synth_inputs(0, dev)
    x2 = torch.cat([lat, lat]).half().contiguous()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    ctx = None
    for r in range(4):
        line = f"round {r}:"
        for mode in modes:
            lib.lavie_debug_force_tile(mode)
            net.prepare(2, bench.FRAMES, bench.LAT_H, bench.LAT_W, bench.CTX_LEN)      # the plan follows the kernel choice
            ctx = net.cache_context(torch.cat([ne, pe]).half().contiguous())
            net.set_cfg_shared_input(True)
            net(x2, 500, encoder_hidden_states=ctx)
            s, e = ev(), ev()
            s.record()
            for _ in range(10):
                net(x2, 500, encoder_hidden_states=ctx)
            e.record()
            torch.cuda.synchronize()
            line += f"  mode {mode:#x}: {s.elapsed_time(e) / 10:7.3f} ms"
            if r == 3:
                bench.profile_begin(lib, 0x7FF, 4096)
                net(x2, 500, encoder_hidden_states=ctx)
                rows = bench.profile_end(lib)
                print(f"mode {mode:#x}", " ".join(f"{q['name'].split('_')[0]}={q['ms']:.3f}({q['launches']})" for q in rows if q["launches"]), flush=True)
            net.set_cfg_shared_input(False)
            net.cache_context(None)
        print(line, flush=True)
    lib.lavie_debug_force_tile(0)


if __name__ == "__main__":
    main()
