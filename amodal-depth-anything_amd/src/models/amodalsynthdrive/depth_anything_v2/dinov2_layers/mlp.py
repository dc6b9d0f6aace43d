"""Import path of the reference kept (mlp.py): the implementation lives in ffn.py."""
from .ffn import Mlp  # noqa: F401
