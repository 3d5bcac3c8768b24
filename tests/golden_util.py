import os

import torch

from lavie_amd import weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=True)


def synth16(shapes, seed, prefix=""):
    """Seeded weights rounded to fp16 and returned as fp32 (what the reference ran with)."""
    sd = weights.synth_state_dict({k: tuple(v) for k, v in shapes.items()}, seed)
    return {prefix + k: v.to(torch.float16).to(torch.float32) for k, v in sd.items()}
