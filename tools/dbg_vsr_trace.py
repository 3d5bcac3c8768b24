"""Development aid: one VSR forward with identical batch halves under LAVIE_DEBUG_TRACE_HALVES=1 (engine.cpp prints, step by step, whether the halves still agree)."""
import sys, os
os.environ["LAVIE_DEBUG_TRACE_HALVES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lavie_amd import _lib, spec, weights
from lavie_amd.config import VSR_CONFIG
from lavie_amd.vsr import UNet3DVSRModel

H, W, F_ = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sd = weights.synth_state_dict(spec.param_shapes(VSR_CONFIG), 0)
net = UNet3DVSRModel(init_weights=False, sample_size=128, down_temporal_idx=(0, 1, 2, 3), mid_temporal=True, up_temporal_idx=(0, 1, 2, 3))
net.load_state_dict({k: v.half() for k, v in sd.items()})
net = net.to("cuda", torch.float16)
gg = torch.Generator().manual_seed(8)
x1 = torch.randn(1, 4, F_, H, W, generator=gg).half()
l1 = torch.randn(1, 3, F_, H, W, generator=gg).half()
c1 = torch.randn(1, 77, 1024, generator=gg).half()
x, low, ctx = torch.cat([x1, x1]).cuda(), torch.cat([l1, l1]).cuda(), torch.cat([c1, c1]).cuda()
if len(sys.argv) > 4 and sys.argv[4] == "nofold":
    net(x[:, :, :1], 500, low[:, :, :1], encoder_hidden_states=ctx, class_labels=torch.tensor([20, 20]))
    _lib.load().lavie_unet_set_ln_fold(net.engine_handle(), 0)
    print("LayerNorm folding off", file=sys.stderr, flush=True)
if len(sys.argv) > 4 and sys.argv[4].startswith("tile"):
    _lib.load().lavie_debug_force_tile(int(sys.argv[4][4:], 0))
y = net(x, 500, low, encoder_hidden_states=ctx, class_labels=torch.tensor([20, 20])).sample
torch.cuda.synchronize()
print("halves equal:", bool(torch.equal(y[0], y[1])))
