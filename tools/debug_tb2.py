"""Dumps the register tiles of workgroup 0 / wave 0 / head pair 0 of the fused temporal kernel and compares them with the
values the D-layout maps say they should hold (development aid)."""
import math
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from gpu_util import f32, h16, q16  # noqa: E402
from lavie_amd import _lib, ops  # noqa: E402
from oracle import unet_fp32 as O  # noqa: E402


def main():
    C, heads, Fr, B, D = 320, 8, 16, 1, 24
    cfg = O.UNetConfig()
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: q16(torch.randn(*s, generator=g))
    sd = {"to_q.weight": q16(rnd(C, C) / math.sqrt(C)), "to_k.weight": q16(rnd(C, C) / math.sqrt(C)),
          "to_v.weight": q16(rnd(C, C) / math.sqrt(C)), "to_out.0.weight": q16(rnd(C, C) / math.sqrt(C)),
          "to_out.0.bias": torch.zeros(C)}
    emb = q16(torch.randn(cfg.rel_buckets, heads, generator=g))
    relbias = O.rel_pos_bias({"time_rel_pos_bias.relative_attention_bias.weight": emb}, "", Fr, cfg).contiguous()
    gamma, beta = torch.ones(C), torch.zeros(C)
    x = q16(torch.randn(B * Fr * D, C, generator=g))
    inv = 10000.0 ** (-torch.arange(0, 32, 2, dtype=torch.float32) / 32)
    ang = torch.arange(Fr, dtype=torch.float32).reshape(Fr, 1) * inv.reshape(1, -1)
    c, s = ang.cos(), ang.sin()
    scale = 40 ** -0.5
    # expected per-head tensors for pixel 0 (unit 0 = b 0, pixel 0): rows f * D
    xr = x.reshape(B, Fr, D, C)[0, :, 0, :]                       # [16, C]
    ln = F.layer_norm(xr, (C,), gamma, beta, 1e-5).half().float()
    q = (ln @ sd["to_q.weight"].t()) * scale
    k = ln @ sd["to_k.weight"].t()
    v = ln @ sd["to_v.weight"].t()

    def rot(t):                                                    # [16, 40] of one head
        out = t.clone()
        ev, od = t[:, 0:32:2], t[:, 1:32:2]
        out[:, 0:32:2] = ev * c - od * s
        out[:, 1:32:2] = od * c + ev * s
        return out
    PAIR = 1
    qh = [rot(q[:, h * 40:(h + 1) * 40]) for h in range(2 * PAIR, 2 * PAIR + 2)]
    kh = [rot(k[:, h * 40:(h + 1) * 40]) for h in range(2 * PAIR, 2 * PAIR + 2)]
    vh = [v[:, h * 40:(h + 1) * 40] for h in range(2 * PAIR, 2 * PAIR + 2)]
    dbg = torch.zeros(420 * 64, dtype=torch.float32, device="cuda")
    lib = _lib.load()
    lib.lavie_debug_temporal_block_dump(dbg.data_ptr())
    img = ops.pack_temporal_block(h16(sd["to_q.weight"]), h16(sd["to_k.weight"]), h16(sd["to_v.weight"]), h16(sd["to_out.0.weight"]))
    got = ops.temporal_block(h16(x), img, f32(gamma), f32(beta), f32(sd["to_out.0.bias"]), f32(relbias), f32(c), f32(s), B, Fr, D, heads, 32, scale)
    torch.cuda.synchronize()
    lib.lavie_debug_temporal_block_dump(None)
    got2 = ops.temporal_block(h16(x), img, f32(gamma), f32(beta), f32(sd["to_out.0.bias"]), f32(relbias), f32(c), f32(s), B, Fr, D, heads, 32, scale)
    sd["time_rel_pos_bias.relative_attention_bias.weight"] = emb
    xr_all = x.reshape(B, Fr, D, C).permute(0, 2, 1, 3).reshape(B * D, Fr, C)
    dl = O.temporal_attention(sd, "", F.layer_norm(xr_all, (C,), gamma, beta, 1e-5), cfg).reshape(B, D, Fr, C).permute(0, 2, 1, 3).reshape(B * Fr * D, C)
    from gpu_util import rel_l2
    # residual registers after each head pair: x + sum over the pairs so far of Wo[:, pair channels] O_pair  (bias is zero here)
    lane_ = torch.arange(64); qq_, col_ = lane_ // 16, lane_ % 16
    sc_all = torch.stack([rot(q[:, h * 40:(h + 1) * 40]) @ rot(k[:, h * 40:(h + 1) * 40]).t() + relbias[h] for h in range(8)])
    o_all = torch.cat([torch.softmax(sc_all[h], dim=-1) @ v[:, h * 40:(h + 1) * 40] for h in range(8)], dim=1)      # [16, 320]
    dr = dbg.cpu().reshape(420, 64)
    for hp in range(4):
        part = xr + o_all[:, :80 * (hp + 1)] @ sd["to_out.0.weight"][:, :80 * (hp + 1)].t()      # [16 frames, 320]
        per_tile = []
        for t in range(20):
            w_ = 0.0
            for r in range(4):
                exp = torch.tensor([part[int(col_[l]), 16 * t + 4 * int(qq_[l]) + r].item() for l in range(64)])
                w_ = max(w_, (dr[100 + hp * 80 + t * 4 + r] - exp).abs().max().item())
            per_tile.append(round(w_, 3))
        print(f"residual registers after head pair {hp}: max|diff| per output tile {per_tile}", flush=True)
    print("final output with the dump on :", rel_l2(got.float().cpu() - x, dl))
    print("final output with the dump off:", rel_l2(got2.float().cpu() - x, dl), flush=True)
    dall = dbg.cpu().reshape(420, 64)
    d = dall[:100]
    lane = torch.arange(64)
    qq, col = lane // 16, lane % 16

    def tile_channels(j, r):     # D layout [channel][frame]: channel index within the head pair's 80 = (head, d) for tile j, row 4 q + r
        row = 4 * qq + r
        if j == 0: return 0, row
        if j == 1: return 0, 16 + row
        if j == 3: return 1, row
        if j == 4: return 1, 16 + row
        return (row >= 8).long(), 32 + (row % 8)

    def report(name, got, exp):
        err = (got - exp).abs().max().item()
        print(f"{name:28s} max|diff| {err:9.4f}   max|exp| {exp.abs().max().item():8.3f}", flush=True)

    for which, nm, src in ((0, "q", qh), (1, "k", kh)):
        for j in range(5):
            for r in range(4):
                hsel, dch = tile_channels(j, r)
                exp = torch.stack([src[int(hh)][int(cc), int(dd)] if not torch.is_tensor(hh) else None for hh, cc, dd in []]) if False else None
                hs = hsel if torch.is_tensor(hsel) else torch.full((64,), hsel)
                dd = dch if torch.is_tensor(dch) else torch.full((64,), dch)
                exp = torch.tensor([src[int(hs[l])][int(col[l]), int(dd[l])].item() for l in range(64)])
                report(f"{nm} tile {j} reg {r}", d[which * 20 + j * 4 + r], exp)
    # v tiles: D layout [frame][channel]: lane (col = channel-in-tile, q): rows 4 q + r = frames
    for j in range(5):
        for r in range(4):
            exp = []
            for l in range(64):
                n = int(col[l]); fr = 4 * int(qq[l]) + r
                if j == 0: hsel, dch = 0, n
                elif j == 1: hsel, dch = 0, 16 + n
                elif j == 3: hsel, dch = 1, n
                elif j == 4: hsel, dch = 1, 16 + n
                else: hsel, dch = (1 if n >= 8 else 0), 32 + n % 8
                exp.append(vh[hsel][fr, dch].item())
            report(f"v tile {j} reg {r}", d[2 * 20 + j * 4 + r], torch.tensor(exp))
    for E in range(2):
        S = qh[E] @ kh[E].t() + relbias[2 * PAIR + E]                 # [query, key]
        P = torch.softmax(S, dim=-1)
        Oh = P @ vh[E]                                     # [query, 40]
        for r in range(4):
            key = 4 * qq + r
            report(f"head {E} S^T reg {r}", d[(3 + E) * 20 + r], torch.tensor([S[int(col[l]), int(key[l])].item() for l in range(64)]))
            report(f"head {E} P^T reg {r}", d[(3 + E) * 20 + 4 + r], torch.tensor([P[int(col[l]), int(key[l])].item() for l in range(64)]))
            ch = 4 * qq + r
            report(f"head {E} O tile a reg {r}", d[(3 + E) * 20 + 8 + r], torch.tensor([Oh[int(col[l]), int(ch[l])].item() for l in range(64)]))
            report(f"head {E} O tile b reg {r}", d[(3 + E) * 20 + 12 + r], torch.tensor([Oh[int(col[l]), 16 + int(ch[l])].item() for l in range(64)]))
            own = (ch < 8) if E == 0 else (ch >= 8)
            exp = torch.tensor([Oh[int(col[l]), 32 + int(ch[l]) % 8].item() if own[l] else float("nan") for l in range(64)])
            got = d[(3 + E) * 20 + 16 + r]
            m = ~torch.isnan(exp)
            report(f"head {E} O shared (own rows) reg {r}", got[m], exp[m])


if __name__ == "__main__":
    main()
