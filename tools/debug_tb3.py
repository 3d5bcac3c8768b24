import math, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from gpu_util import f32, h16, q16
from lavie_amd import ops
from oracle import unet_fp32 as O

C, heads, Fr, B, D = 320, 8, 16, 1, 24
cfg = O.UNetConfig()
g = torch.Generator().manual_seed(0)
rnd = lambda *s: q16(torch.randn(*s, generator=g))
wv = q16(rnd(C, C) / math.sqrt(C)); wo = q16(rnd(C, C) / math.sqrt(C))
gamma, beta = torch.ones(C), torch.zeros(C)
x = q16(torch.randn(B * Fr * D, C, generator=g))
zero_ang = torch.zeros(Fr, 16)
zb = torch.zeros(heads, Fr, Fr)
torch.set_printoptions(precision=3, linewidth=200, sci_mode=False)
for (h, lo, hi) in ((0, 0, 32), (0, 32, 40), (1, 0, 32), (1, 32, 40), (2, 0, 32), (7, 32, 40)):
    w = torch.zeros(C, C); w[:, h * 40 + lo:h * 40 + hi] = wo[:, h * 40 + lo:h * 40 + hi]
    xr = x.reshape(B, Fr, D, C).permute(0, 2, 1, 3).reshape(B * D, Fr, C)
    ln = F.layer_norm(xr, (C,), gamma, beta, 1e-5)
    v = ln @ wv.t()
    o = v.mean(dim=1, keepdim=True).expand(-1, Fr, -1)          # uniform attention
    delta = (o @ w.t()).reshape(B, D, Fr, C).permute(0, 2, 1, 3).reshape(B * Fr * D, C)
    img = ops.pack_temporal_block(h16(torch.zeros(C, C)), h16(torch.zeros(C, C)), h16(wv), h16(w))
    got = ops.temporal_block(h16(x), img, f32(gamma), f32(beta), f32(torch.zeros(C)), f32(zb), f32(zero_ang.cos()), f32(zero_ang.sin()),
                             B, Fr, D, heads, 32, 40 ** -0.5).float().cpu() - x
    print(f"--- to_out restricted to head {h} channels {lo}..{hi}: pixel 0, frame 0")
    print("expected", delta[0, :40])
    print("got     ", got[0, :40])
    r = (got[0] / delta[0])
    print("ratio   ", r[:40])
    print("frame 5 ratio", (got[5 * D] / delta[5 * D])[:24])
    print("pixel 3 frame 0 ratio", (got[3] / delta[3])[:24], flush=True)
