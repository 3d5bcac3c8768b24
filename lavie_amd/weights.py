"""Deterministic synthetic ("random-init") weights and the checkpoint loader.

There are no pretrained weights in the build environment, so benchmarks
and parity fixtures use a seeded recipe that is independent of device, thread count and tensor
creation order: tensor i (names sorted) is drawn from `torch.Generator().manual_seed(seed*1000003+i)`
on the CPU.  Distributions follow the reference constructor's defaults (PyTorch Linear/Conv
U(-1/sqrt(fan_in), 1/sqrt(fan_in)), Embedding N(0,1)) with three deliberate deviations that make
parity tests meaningful: `attn_temp.to_out.0.weight` ~ N(0, 0.02^2) instead
of the zero init at attention.py:475; norm scales 1 + 0.1 N(0,1); norm/linear biases 0.05 N(0,1).
"""
import math
from typing import Dict, Mapping, Tuple

import torch

Shape = Tuple[int, ...]


def rotary_freqs(rot_dim: int, theta: float = 10000.0) -> torch.Tensor:
    """`RotaryEmbedding(rot_dim).freqs` (theta^(-2k/rot_dim)); stored in the state dict under
    every `...attn_temp.rotary_emb.freqs` key."""
    return theta ** (-torch.arange(0, rot_dim, 2, dtype=torch.float32) / rot_dim)


def synth_tensor(name: str, shape: Shape, gen: torch.Generator) -> torch.Tensor:
    if name.endswith("rotary_emb.freqs"):
        return rotary_freqs(shape[0] * 2)
    if name.endswith("relative_attention_bias.weight"):
        return torch.randn(shape, generator=gen)
    if name.endswith("attn_temp.to_out.0.weight"):
        return torch.randn(shape, generator=gen) * 0.02
    if len(shape) == 1:
        r = torch.randn(shape, generator=gen)
        is_scale = name.endswith(".weight")          # 1-D weights are norm scales
        return 1.0 + 0.1 * r if is_scale else 0.05 * r
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    bound = 1.0 / math.sqrt(fan_in)
    return (torch.rand(shape, generator=gen) * 2.0 - 1.0) * bound


def synth_state_dict(shapes: Mapping[str, Shape], seed: int = 0, dtype: torch.dtype = torch.float32,
                     only_prefix: str = "") -> Dict[str, torch.Tensor]:
    """Seeded weights for `shapes` (name -> shape).  `only_prefix` restricts which tensors are
    materialised while keeping every tensor's value identical to the full dict's."""
    out: Dict[str, torch.Tensor] = {}
    for i, name in enumerate(sorted(shapes)):
        if not name.startswith(only_prefix):
            continue
        gen = torch.Generator().manual_seed(seed * 1000003 + i)
        out[name] = synth_tensor(name, tuple(shapes[name]), gen).to(dtype)
    return out


def load_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """`find_model` (base/download.py:10-18): torch.load to CPU, unwrap an optional "ema" entry.
    Loaded with weights_only=True (nothing in the file is executed)."""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(blob, dict) and "ema" in blob:
        blob = blob["ema"]
    return blob
