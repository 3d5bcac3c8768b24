"""A/B of lavie_unet_set_cfg_shared_input inside the full UNet forward (bench shape, cached context), interleaved rounds."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from lavie_amd import spec, weights  # noqa: E402
from lavie_amd.unet import UNet3DConditionModel  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    sd = weights.synth_state_dict(spec.param_shapes(), 0)
    net = UNet3DConditionModel(sample_size=64, cross_attention_dim=bench.CTX_DIM, init_weights=False)
    net.load_state_dict({k: v.half() for k, v in sd.items()})
    net = net.to(dev, torch.float16)
    net.prepare(2, bench.FRAMES, bench.LAT_H, bench.LAT_W, bench.CTX_LEN)
    pe, ne, lat = bench.synth_inputs(0, dev)
    ctx = net.cache_context(torch.cat([ne, pe]).half().contiguous())
    x2 = torch.cat([lat, lat]).half().contiguous()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    outs = {}
    for r in range(4):
        line = f"round {r}:"
        for on in (False, True):
            net.set_cfg_shared_input(on)
            outs[on] = net(x2, 500, encoder_hidden_states=ctx).sample.float()
            s, e = ev(), ev()
            s.record()
            for _ in range(10):
                net(x2, 500, encoder_hidden_states=ctx)
            e.record()
            torch.cuda.synchronize()
            line += f"  shared={on}: {s.elapsed_time(e) / 10:7.3f} ms"
        print(line, flush=True)
    net.set_cfg_shared_input(False)
    print(f"rel-L2 shared vs plain: {((outs[True] - outs[False]).norm() / outs[False].norm()).item():.2e}")


if __name__ == "__main__":
    main()
