"""lavie_amd — MI355X-native LaVie base T2V denoising path.

Hand-written gfx950 HIP kernels behind a C-ABI library (`lavie_amd/csrc`, `include/lavie_hip.h`)
plus the host-side mirror of the reference's `UNet3DConditionModel.forward` /
`VideoGenPipeline` / DDPM-scheduler surface.  There is no CPU fallback: importing the compute
entry points without the built library raises.
"""
from .config import UNetConfig, BASE_CONFIG  # noqa: F401

__version__ = "0.1.0"
