import math, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from gpu_util import f32, h16, q16, rel_l2
from lavie_amd import _lib, ops
from oracle import unet_fp32 as O

lib = _lib.load()
C, heads, Fr = 320, 8, 16
cfg = O.UNetConfig()
for (B, D) in ((2, 12),):
    g = torch.Generator().manual_seed(B * 1000 + D)
    sd = {"to_q.weight": q16(torch.randn(C, C, generator=g) / math.sqrt(C)), "to_k.weight": q16(torch.randn(C, C, generator=g) / math.sqrt(C)),
          "to_v.weight": q16(torch.randn(C, C, generator=g) / math.sqrt(C)), "to_out.0.weight": q16(torch.randn(C, C, generator=g) / math.sqrt(C)),
          "to_out.0.bias": torch.randn(C, generator=g) * 0.2,
          "time_rel_pos_bias.relative_attention_bias.weight": q16(torch.randn(cfg.rel_buckets, heads, generator=g))}
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    x = q16(torch.randn(B * Fr * D, C, generator=g) * 1.5)
    xr = x.reshape(B, Fr, D, C).permute(0, 2, 1, 3).reshape(B * D, Fr, C)
    delta = O.temporal_attention(sd, "", F.layer_norm(xr, (C,), gamma, beta, 1e-5), cfg)
    delta = delta.reshape(B, D, Fr, C).permute(0, 2, 1, 3).reshape(B * Fr * D, C)
    inv = 10000.0 ** (-torch.arange(0, 32, 2, dtype=torch.float32) / 32)
    ang = torch.arange(Fr, dtype=torch.float32).reshape(Fr, 1) * inv.reshape(1, -1)
    relbias = O.rel_pos_bias(sd, "", Fr, cfg).contiguous()
    img = ops.pack_temporal_block(h16(sd["to_q.weight"]), h16(sd["to_k.weight"]), h16(sd["to_v.weight"]), h16(sd["to_out.0.weight"]))
    xd = h16(x)
    args = (img, f32(gamma), f32(beta), f32(sd["to_out.0.bias"]), f32(relbias), f32(ang.cos()), f32(ang.sin()), B, Fr, D, heads, 32, 40 ** -0.5)
    for v, nm in ((0, "shipped"), (2, "plain reads everywhere"), (3, "plain reads in to_out"), (4, "plain reads in q/k/v"), (5, "PF 4")):
        lib.lavie_debug_rowfuse_variant(v)
        got = ops.temporal_block(xd, *args).float().cpu()
        e = rel_l2(got - x, delta)
        # per-row error map: which frames / pixels are wrong
        rowerr = ((got - x - delta).norm(dim=1) / delta.norm(dim=1)).reshape(B, Fr, D)
        bad = (rowerr > 0.02)
        print(f"B={B} D={D} variant {v} ({nm}): delta rel-L2 {e:.4f}; bad rows {int(bad.sum())} / {bad.numel()}; "
              f"bad per frame {bad.sum(dim=(0, 2)).tolist()}", flush=True)
    lib.lavie_debug_rowfuse_variant(0)
