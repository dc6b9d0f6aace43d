from .attention import Attention, MemEffAttention
from .block import Block, NestedTensorBlock
from .layer_scale import LayerScale
from .mlp import Mlp
from .patch_embed import PatchEmbed
from .swiglu_ffn import SwiGLUFFN, SwiGLUFFNFused
