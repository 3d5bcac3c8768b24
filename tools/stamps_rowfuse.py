"""Stamp build of the fused feed-forward kernel (variant 7): where a wave of workgroup 0 spends its cycles (shares, not times)."""
import math
import sys

import torch

sys.path.insert(0, ".")
from lavie_amd import _lib, ops  # noqa: E402

M, C = 81920, 320
g = torch.Generator().manual_seed(0)
x = torch.randn(M, C, generator=g).half().cuda()
w1 = (torch.randn(8 * C, C, generator=g) / math.sqrt(C)).half().cuda()
b1 = (torch.randn(8 * C, generator=g) * 0.1).half().cuda()
w2 = (torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)).half().cuda()
b2 = torch.randn(C, generator=g).cuda()
gamma, beta = torch.ones(C).cuda(), torch.zeros(C).cuda()
img, b1img = ops.pack_geglu_mlp(w1, b1, w2)
out = torch.empty_like(x)
lib = _lib.load()
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
lib.lavie_debug_rowfuse_stamps(buf.data_ptr())
lib.lavie_debug_rowfuse_variant(7)
for _ in range(3):
    ops.geglu_mlp(x, img, b1img, gamma, beta, b2, out=out)
torch.cuda.synchronize()
s = buf.cpu().reshape(8, 8).double()
names = ["DMA issue (5 pieces per sync)", "first product runs", "GEGLU", "second product runs", "pass prologue (x, LN)", "store + tail", "wait for own DMA pieces (vmcnt)", "barrier"]
tot = s.sum(dim=1)
print("cycles per wave (workgroup 0, three passes):", [int(v) for v in tot.tolist()])
for i, n in enumerate(names):
    print(f"{n:36s} " + " ".join(f"{100 * s[w, i] / tot[w]:5.1f}%" for w in range(8)))
lib.lavie_debug_rowfuse_variant(0)
lib.lavie_debug_rowfuse_stamps(None)
