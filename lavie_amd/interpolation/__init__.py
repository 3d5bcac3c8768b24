"""Host-side mirror of the reference's frame-interpolation stage (`interpolation/`): the 16 -> 61 frame
UNet variant on the same gfx950 engine."""
from .diffusion import SpacedDiffusion, create_diffusion  # noqa: F401
from .unet import UNet3DConditionModel  # noqa: F401
