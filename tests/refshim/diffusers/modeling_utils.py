from .models.modeling_utils import ModelMixin  # noqa: F401
