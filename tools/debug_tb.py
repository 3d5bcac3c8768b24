"""Bisects the fused temporal sub-block kernel against the oracle with structured weights (development aid)."""
import math
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from gpu_util import f32, h16, q16, rel_l2  # noqa: E402
from lavie_amd import ops  # noqa: E402
from oracle import unet_fp32 as O  # noqa: E402


def run(sd, gamma, beta, x, B, D, relbias, ang, label):
    C, heads, Fr = 320, 8, 16
    cfg = O.UNetConfig()
    xr = x.reshape(B, Fr, D, C).permute(0, 2, 1, 3).reshape(B * D, Fr, C)
    ln = F.layer_norm(xr, (C,), gamma, beta, 1e-5)
    q = O.split_heads(F.linear(ln, sd["to_q.weight"]), heads)
    k = O.split_heads(F.linear(ln, sd["to_k.weight"]), heads)
    v = O.split_heads(F.linear(ln, sd["to_v.weight"]), heads)
    # oracle core with the GIVEN bias / angle tables (so that structured tables can be tested)
    scale = (C // heads) ** -0.5
    c, s = ang.cos(), ang.sin()

    def rot(t):
        out = t.clone()
        ev, od = t[..., 0:32:2], t[..., 1:32:2]
        out[..., 0:32:2] = ev * c - od * s
        out[..., 1:32:2] = od * c + ev * s
        return out
    sc = rot(q * scale) @ rot(k).transpose(-1, -2) + relbias
    o = torch.softmax(sc, dim=-1) @ v
    delta = F.linear(O.merge_heads(o), sd["to_out.0.weight"], sd["to_out.0.bias"])
    back = lambda t: t.reshape(B, D, Fr, C).permute(0, 2, 1, 3).reshape(B * Fr * D, C)
    delta = back(delta)
    img = ops.pack_temporal_block(h16(sd["to_q.weight"]), h16(sd["to_k.weight"]), h16(sd["to_v.weight"]), h16(sd["to_out.0.weight"]))
    got = ops.temporal_block(h16(x), img, f32(gamma), f32(beta), f32(sd["to_out.0.bias"]), f32(relbias.contiguous()), f32(c), f32(s),
                             B, Fr, D, heads, 32, scale)
    gd = got.float().cpu() - x
    print(f"{label:60s} delta rel-L2 {rel_l2(gd, delta):.4f}   |delta| {delta.norm():.2f}", flush=True)
    return gd, delta


def main():
    C, heads, Fr, B, D = 320, 8, 16, 1, 24
    cfg = O.UNetConfig()
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: q16(torch.randn(*s, generator=g))
    base = {"to_q.weight": rnd(C, C) / math.sqrt(C), "to_k.weight": rnd(C, C) / math.sqrt(C), "to_v.weight": rnd(C, C) / math.sqrt(C),
            "to_out.0.weight": rnd(C, C) / math.sqrt(C), "to_out.0.bias": torch.randn(C, generator=g) * 0.2}
    base = {k: q16(v) for k, v in base.items()}
    emb = q16(torch.randn(cfg.rel_buckets, heads, generator=g))
    relbias = O.rel_pos_bias({"time_rel_pos_bias.relative_attention_bias.weight": emb}, "", Fr, cfg)
    gamma, beta = torch.ones(C), torch.zeros(C)
    x = q16(torch.randn(B * Fr * D, C, generator=g))
    inv = 10000.0 ** (-torch.arange(0, 32, 2, dtype=torch.float32) / 32)
    ang = torch.arange(Fr, dtype=torch.float32).reshape(Fr, 1) * inv.reshape(1, -1)
    zero_ang = torch.zeros_like(ang)
    zb = torch.zeros_like(relbias)

    sd = dict(base)
    sd["to_q.weight"] = torch.zeros(C, C)
    sd["to_k.weight"] = torch.zeros(C, C)
    run(sd, gamma, beta, x, B, D, zb, zero_ang, "uniform attention (q = k = 0, no bias): V and to_out paths")
    for h in range(8):
        for lo, hi, nm in ((0, 32, "ch 0-31"), (32, 40, "ch 32-39")):
            sd2 = dict(sd)
            w = torch.zeros(C, C)
            w[:, h * 40 + lo:h * 40 + hi] = base["to_out.0.weight"][:, h * 40 + lo:h * 40 + hi]
            sd2["to_out.0.weight"] = w
            sd2["to_out.0.bias"] = torch.zeros(C)
            run(sd2, gamma, beta, x, B, D, zb, zero_ang, f"  uniform attention, to_out restricted to head {h} {nm}")
    run(sd, gamma, beta, x, B, D, relbias, zero_ang, "q = k = 0, with bias: softmax(bias)")
    run(base, gamma, beta, x, B, D, zb, zero_ang, "full q k v, no bias, no rotary")
    for h in range(8):
        sd2 = dict(base)
        w = torch.zeros(C, C)
        w[:, h * 40:h * 40 + 40] = base["to_out.0.weight"][:, h * 40:h * 40 + 40]
        sd2["to_out.0.weight"] = w
        run(sd2, gamma, beta, x, B, D, zb, zero_ang, f"  full q k v, no bias, no rotary, to_out restricted to head {h}")
    run(base, gamma, beta, x, B, D, zb, ang, "full q k v, no bias, rotary")
    run(base, gamma, beta, x, B, D, relbias, ang, "everything")


if __name__ == "__main__":
    main()
