"""Import path of the reference kept (swiglu_ffn.py): the implementation lives in ffn.py."""
from .ffn import SwiGLUFFN, SwiGLUFFNFused  # noqa: F401
