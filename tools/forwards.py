"""N UNet forwards at the bench shape (CFG batch 2, cached context) and nothing else: the program to put under
`rocprofv3 --kernel-trace --stats` for a clean per-forward kernel table (bench.py's trace mixes in the batched-prompt leg)."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from lavie_amd import spec, weights  # noqa: E402
from lavie_amd.unet import UNet3DConditionModel  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    sd = weights.synth_state_dict(spec.param_shapes(), 0)
    net = UNet3DConditionModel(sample_size=64, cross_attention_dim=bench.CTX_DIM, init_weights=False)
    net.load_state_dict({k: v.half() for k, v in sd.items()})
    net = net.to(dev, torch.float16)
    net.prepare(2, bench.FRAMES, bench.LAT_H, bench.LAT_W, bench.CTX_LEN)
    pe, ne, lat = bench.synth_inputs(0, dev)
    ctx = net.cache_context(torch.cat([ne, pe]).half().contiguous())
    x2 = torch.cat([lat, lat]).half().contiguous()
    for _ in range(n):
        net(x2, 500, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
    print(f"{n} forwards done")


if __name__ == "__main__":
    main()
