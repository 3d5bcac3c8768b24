"""fp32 CPU restatement of the VSR stage's temporal building blocks (TEST INFRASTRUCTURE, see oracle/__init__.py) —
SURVEY.md §8 f2, first pieces: the (T,1,1) temporal convolution and `ResnetBlock3DCNN`.

PINNED: `vsr/models/resnet.py` imports only torch / einops and is imported directly in the build container
(tests/test_oracle_vs_reference.py, tier T1); outputs frozen in tests/golden/vsr_resnet3dcnn.pt."""
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def temporal_conv(x, w, b):
    """nn.Conv3d(kernel (T,1,1), stride 1, padding (T//2,0,0)) (vsr/models/resnet.py:258-259, 274) written as T shifted
    matrix products: x [b, c, f, h, w], w [cout, cin, T, 1, 1]."""
    taps = w.shape[2]
    f = x.shape[2]
    xp = F.pad(x, (0, 0, 0, 0, taps // 2, taps // 2))
    y = sum(torch.einsum("oc,bcfhw->bofhw", w[:, :, t, 0, 0], xp[:, :, t:t + f]) for t in range(taps))
    return y + b.reshape(1, -1, 1, 1, 1)


def resnet_block_3dcnn(sd: SD, p: str, x, temb, groups: int = 32, eps: float = 1e-6):
    """ResnetBlock3DCNN.forward (vsr/models/resnet.py:283-315), in == out channels (no shortcut), time_embedding_norm
    'default', output_scale_factor 1: GroupNorm over (C/G, F, H, W) + SiLU -> temporal conv (T) + temb -> GroupNorm +
    SiLU -> temporal conv (3) -> residual."""
    h = F.silu(F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps))
    h = temporal_conv(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"])
    t = F.linear(F.silu(temb), sd[p + "time_emb_proj.weight"], sd[p + "time_emb_proj.bias"])
    h = h + t[:, :, None, None, None]
    h = F.silu(F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps))
    h = temporal_conv(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"])
    return x + h
