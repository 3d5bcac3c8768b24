"""ConfigMixin / register_to_config stand-ins: plumbing only, no arithmetic."""
import functools
import inspect
from types import SimpleNamespace


class _Config(SimpleNamespace):
    def get(self, k, default=None):
        return getattr(self, k, default)

    def __getitem__(self, k):
        return getattr(self, k)


def register_to_config(init):
    sig = inspect.signature(init)

    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        self._internal_cfg = _Config(**cfg)
        init(self, *args, **kwargs)

    return wrapped


class ConfigMixin:
    @property
    def config(self):
        return self._internal_cfg

    @classmethod
    def from_config(cls, config, **kw):
        names = set(inspect.signature(cls.__init__).parameters) - {"self"}
        return cls(**{k: v for k, v in dict(config).items() if k in names})
