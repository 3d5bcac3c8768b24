"""RotaryEmbedding stand-in (lucidrains/rotary-embedding-torch semantics; the reference
leaves the package unpinned: environment.yml:26).  Parity unpinned.

RotaryEmbedding(dim): freqs_k = theta^(-2k/dim), k < dim/2, stored as an nn.Parameter
named `freqs` (it shows up in the UNet state dict).  rotate_queries_or_keys(t) rotates
the first `dim` channels of the last axis with positions 0..n-1 taken from axis -2,
pairing channels (2k, 2k+1)."""
import torch
from torch import nn


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, theta=10000):
        super().__init__()
        self.rot_dim = dim
        self.freqs = nn.Parameter(1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim)),
                                  requires_grad=False)

    def rotate_queries_or_keys(self, t, seq_dim=-2):
        n = t.shape[seq_dim]
        pos = torch.arange(n, device=t.device).type_as(self.freqs)
        ang = torch.einsum("n,f->nf", pos, self.freqs).repeat_interleave(2, dim=-1)  # [n, dim]
        rot, rest = t[..., : self.rot_dim], t[..., self.rot_dim:]
        pairs = rot.reshape(*rot.shape[:-1], -1, 2)
        half = torch.stack((-pairs[..., 1], pairs[..., 0]), dim=-1).reshape(rot.shape)
        rot = rot * ang.cos().to(rot.dtype) + half * ang.sin().to(rot.dtype)
        return torch.cat((rot, rest), dim=-1)
