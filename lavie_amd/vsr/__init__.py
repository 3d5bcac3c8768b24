"""Host-side mirror of the reference's video super-resolution stage (`vsr/`): `UNet3DVSRModel` on the
gfx950 engine and the latent upscaling loop around it (text encoder / VAE are attachable stock objects)."""
from .pipeline import VideoUpscalePipeline, upscale_in_chunks  # noqa: F401
from .unet import UNet3DVSRModel  # noqa: F401
