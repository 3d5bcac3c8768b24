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
