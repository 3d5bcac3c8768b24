"""Symbol-only stand-in: vsr/models/temporal_module.py imports torchvision at module level but only calls
torchvision.ops.deform_conv2d inside the deformable-attention branch, which the VSR configuration
(attention_block_types ["", ""]) never builds."""


class _Ops:
    @staticmethod
    def deform_conv2d(*a, **k):
        raise NotImplementedError("torchvision is not installed in this image")


ops = _Ops()
