import logging as _logging

WEIGHTS_NAME = "diffusion_pytorch_model.bin"


class BaseOutput:
    """Dataclass base; the reference only reads `.sample`."""


class _Logging:
    @staticmethod
    def get_logger(name):
        return _logging.getLogger(name)


logging = _Logging()


def deprecate(*a, **k):
    pass


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.randn_tensor contract: a CPU generator draws on the CPU, the result is moved."""
    import torch
    device = torch.device(device) if device is not None else torch.device("cpu")
    gen_dev = generator.device.type if generator is not None else device.type
    if gen_dev == "cpu" and device.type != "cpu":
        return torch.randn(shape, generator=generator, dtype=dtype).to(device)
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)
