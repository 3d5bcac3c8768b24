"""Host-side mirror of the reference's video super-resolution stage (`vsr/`) — in progress: the
transformer-block variant of `UNet3DVSRModel` and the temporal convolution blocks run on the gfx950 engine; the UNet's
temporal modules, class-embedded noise level and the upscale pipeline are not built yet."""
from .unet import UNet3DVSRModel  # noqa: F401
