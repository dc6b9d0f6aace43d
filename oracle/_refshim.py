"""TEST INFRASTRUCTURE ONLY -- import shim for the *real* reference at /root/reference.

Used only in the build container (where /root/reference exists) by
``oracle/make_golden.py`` and by ``tests/test_oracle_vs_reference.py`` to pin the oracle
restatement (``oracle/dav2_oracle.py``) against the reference itself.  Nothing here is
importable on the GPU box (the reference does not travel) and nothing in the product
package may import this module.

Why a shim: ``src/models/__init__.py:1-9`` drags in DepthFM/torchdiffeq,
``depth_anything_v2/dpt.py:1,5`` imports cv2 / torchvision at module import and
``dav2.py:11`` imports timm.  None of them is touched by ``forward``; empty stand-in
modules in ``sys.modules`` plus bare namespace packages for ``src`` / ``src.models`` let the
two model files import unmodified.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("ADA_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "src", "models", "amodalsynthdrive", "dav2.py"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    mod.__path__ = []  # behave like a package so "import a.b" works
    sys.modules[name] = mod
    return mod


def _namespace(name, path):
    mod = types.ModuleType(name)
    mod.__path__ = [path]
    sys.modules[name] = mod
    return mod


_LOADED = {}


def load_reference():
    """Returns (AmodalDAv2, RawDepthAnythingV2) classes imported from the reference tree."""
    if _LOADED:
        return _LOADED["amodal"], _LOADED["raw"]
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)

    class _Anything:  # placeholder for never-called symbols (Compose, timm.create_model ...)
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            raise RuntimeError("stubbed third-party symbol was called")

    _stub("cv2", INTER_CUBIC=2, INTER_AREA=3, INTER_NEAREST=0)
    _stub("timm")
    tv = _stub("torchvision")
    tvt = _stub("torchvision.transforms", Compose=_Anything)
    tv.transforms = tvt

    # Our own product package also exposes a top-level ``src``; keep the reference's copy under
    # a private alias so both can coexist in one interpreter.
    saved = {k: v for k, v in sys.modules.items() if k == "src" or k.startswith("src.")}
    for k in saved:
        del sys.modules[k]
    try:
        _namespace("src", os.path.join(REFERENCE_ROOT, "src"))
        _namespace("src.models", os.path.join(REFERENCE_ROOT, "src", "models"))
        _namespace("src.models.amodalsynthdrive", os.path.join(REFERENCE_ROOT, "src", "models", "amodalsynthdrive"))
        # dav2.py:14 imports deeplab symbols it never uses -> give it a stand-in module.
        _stub(
            "src.models.amodalsynthdrive.deeplab",
            resize=_Anything, Conv2DModule=_Anything, ASPPModule=_Anything, ASPPHead=_Anything,
            UpSample=_Anything, DepthPredictionHead=_Anything, mViT=_Anything,
        )
        dav2 = importlib.import_module("src.models.amodalsynthdrive.dav2")
        raw = importlib.import_module("src.models.amodalsynthdrive.depth_anything_v2_raw.dpt")
        _LOADED["amodal"] = dav2.AmodalDAv2
        _LOADED["raw"] = raw.DepthAnythingV2
    finally:
        ref_mods = {k: v for k, v in sys.modules.items() if k == "src" or k.startswith("src.")}
        for k, v in ref_mods.items():
            del sys.modules[k]
            sys.modules["_adaref_" + k] = v
        sys.modules.update(saved)
    return _LOADED["amodal"], _LOADED["raw"]
