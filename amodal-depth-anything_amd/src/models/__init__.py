"""Model registry with the reference's surface: ``get_model(name, **kwargs)``
(reference src/models/__init__.py:24-31).  Only the hot-path family is registered -- the reference's
other entries (DepthFM, ADDeepLab, ...) are out of scope and, unlike the reference,
importing this package pulls in no diffusers / torchdiffeq / timm."""
from .amodalsynthdrive.dav2 import AmodalDAv2

model_name_class_dict = {"AmodalDAv2": AmodalDAv2}


def get_model(model_name, **kwargs):
    if model_name not in model_name_class_dict:
        raise NotImplementedError
    return model_name_class_dict[model_name](**kwargs)
