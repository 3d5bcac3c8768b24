import torch


class ModelMixin(torch.nn.Module):
    """Plumbing: `.dtype`/`.device` accessors used by the reference (unet.py:433)."""

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device
