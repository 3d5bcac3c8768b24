
import math, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from gpu_util import f32, h16, q16
from lavie_amd import _lib, ops
from oracle import unet_fp32 as O
C, heads, Fr, B, D = 320, 8, 16, 1, 24
cfg = O.UNetConfig()
g = torch.Generator().manual_seed(0)
rnd = lambda *s: q16(torch.randn(*s, generator=g))
sd = {"to_q.weight": q16(rnd(C, C) / math.sqrt(C)), "to_k.weight": q16(rnd(C, C) / math.sqrt(C)),
      "to_v.weight": q16(rnd(C, C) / math.sqrt(C)), "to_out.0.weight": q16(rnd(C, C) / math.sqrt(C)), "to_out.0.bias": torch.zeros(C)}
emb = q16(torch.randn(cfg.rel_buckets, heads, generator=g))
relbias = O.rel_pos_bias({"time_rel_pos_bias.relative_attention_bias.weight": emb}, "", Fr, cfg).contiguous()
gamma, beta = torch.ones(C), torch.zeros(C)
x = q16(torch.randn(B * Fr * D, C, generator=g))
inv = 10000.0 ** (-torch.arange(0, 32, 2, dtype=torch.float32) / 32)
ang = torch.arange(Fr, dtype=torch.float32).reshape(Fr, 1) * inv.reshape(1, -1)
dbg = torch.zeros(520 * 64, dtype=torch.float32, device="cuda")
lib = _lib.load()
lib.lavie_debug_temporal_block_dump(dbg.data_ptr())
img = ops.pack_temporal_block(h16(sd["to_q.weight"]), h16(sd["to_k.weight"]), h16(sd["to_v.weight"]), h16(sd["to_out.0.weight"]))
ops.temporal_block(h16(x), img, f32(gamma), f32(beta), f32(sd["to_out.0.bias"]), f32(relbias), f32(ang.cos()), f32(ang.sin()), B, Fr, D, heads, 32, 40 ** -0.5)
torch.cuda.synchronize()
lib.lavie_debug_temporal_block_dump(None)
d = dbg.cpu().reshape(520, 64)
lane = torch.arange(64); qq, col = lane // 16, lane % 16
xr = x.reshape(B, Fr, D, C)[0, :, 0, :]
# R before the to_out product of pair 0 must be x
worst = 0
for t in range(20):
    for r in range(4):
        exp = torch.tensor([xr[int(col[l]), 16 * t + 4 * int(qq[l]) + r].item() for l in range(64)])
        worst = max(worst, (d[440 + t * 4 + r] - exp).abs().max().item())
print("R before the to_out product of pair 0 vs x: max|diff|", worst)
# rebuild O[frame][80 channels of the pair] from the dumped B operands
Opair = torch.zeros(16, 80)
for l in range(64):
    f, q = int(col[l]), int(qq[l])
    for j in range(8):
        ch = 16 * (j >> 2) + 4 * q + (j & 3)
        Opair[f, ch] = d[420 + j, l]            # head 0 channels 0..31
        Opair[f, 40 + ch] = d[428 + j, l]       # head 1
    for j in range(4):
        row = 4 * q + j
        if row < 8: Opair[f, 32 + row] = d[436 + j, l]
        else: Opair[f, 40 + 32 + row - 8] = d[436 + j, l]
part = xr + Opair @ sd["to_out.0.weight"][:, :80].t()
per_tile = []
for t in range(20):
    w_ = 0
    for r in range(4):
        exp = torch.tensor([part[int(col[l]), 16 * t + 4 * int(qq[l]) + r].item() for l in range(64)])
        w_ = max(w_, (d[100 + 0 * 80 + t * 4 + r] - exp).abs().max().item())
    per_tile.append(round(w_, 3))
print("R after pair 0 vs x + Wo[:, :80] O(dumped operands): per tile", per_tile)
# which weights did the wrong tiles see?  contributions of the three k-ranges per candidate output tile
W = sd["to_out.0.weight"]
Oh0, Oh1 = Opair[:, 0:32], Opair[:, 40:72]
Osh = torch.cat([Opair[:, 32:40], Opair[:, 72:80]], dim=1)
def contrib(tc):
    rows = slice(16 * tc, 16 * tc + 16)
    c0 = Oh0 @ W[rows, 0:32].t()
    c1 = Oh1 @ W[rows, 40:72].t()
    cs = Osh @ torch.cat([W[rows, 32:40], W[rows, 72:80]], dim=1).t()
    return c0, c1, cs                              # [16 frames, 16 channels]
for t in (5, 7, 9, 13, 15, 17, 4, 3):
    got = torch.zeros(16, 16)
    for l in range(64):
        for r in range(4):
            got[int(col[l]), 4 * int(qq[l]) + r] = d[100 + t * 4 + r, l] - xr[int(col[l]), 16 * t + 4 * int(qq[l]) + r]
    c0, c1, cs = contrib(t)
    best = None
    for tc in range(20):
        a0, a1, a_s = contrib(tc)
        for m0 in (0, 1):
            for m1 in (0, 1):
                for ms in (0, 1):
                    for own in (0, 1):       # own tile's other parts present
                        cand = m0 * a0 + m1 * a1 + ms * a_s
                        if own: cand = cand + (1 - m0) * c0 + (1 - m1) * c1 + (1 - ms) * cs
                        e = (got - cand).abs().max().item()
                        if best is None or e < best[0]: best = (e, tc, m0, m1, ms, own)
    print(f"tile {t}: expected-fit error {(got - c0 - c1 - cs).abs().max():.3f}; best fit: err {best[0]:.3f} with tile {best[1]} parts(h0,h1,shared)=({best[2]},{best[3]},{best[4]}) own-rest={best[5]}")
