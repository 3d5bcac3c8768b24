"""FeedForward / GEGLU restated from the published diffusers 0.16 behaviour.

Textual spec available in-tree: /root/reference/vsr/models/diffusers_attention.py:734-822
(`net.0.proj: Linear(dim, 2*inner)`, `h, gate = chunk(2)`, `h * gelu(gate)` (exact erf
GELU), `net.1: Dropout`, `net.2: Linear(inner, dim)`).  Parity unpinned against the
real package (absent from this image)."""
import torch
import torch.nn.functional as F
from torch import nn


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu"):
        super().__init__()
        assert activation_fn == "geglu"
        inner = int(dim * mult)
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out or dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class AdaLayerNorm(nn.Module):  # never instantiated on the base path (num_embeds_ada_norm=None)
    def __init__(self, *a, **k):
        raise NotImplementedError
