"""Symbol-only stand-in: vsr/models/diffusers_attention.py imports `Attention` at module level; the VSR configuration
(TemporalModule3D with attention_block_types ["", ""]) never instantiates it."""


class Attention:
    def __init__(self, *a, **k):
        raise NotImplementedError
