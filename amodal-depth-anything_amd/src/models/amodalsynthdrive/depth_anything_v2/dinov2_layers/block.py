"""Transformer block (reference DA2/dinov2_layers/block.py:36-107,245-247), inference branch only:
    x = x + ls1(attn(norm1(x)));  x = x + ls2(mlp(norm2(x)))
Stochastic depth and the nested-tensor (list) path are training-only and not provided."""
from torch import nn

from .attention import Attention
from .layer_scale import LayerScale
from .mlp import Mlp


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, proj_bias=True, ffn_bias=True, drop=0.0,
                 attn_drop=0.0, init_values=None, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 attn_class=Attention, ffn_layer=Mlp):
        super().__init__()
        if drop_path > 0.0:
            raise NotImplementedError("stochastic depth is a training feature; this build is inference-only")
        self.norm1 = norm_layer(dim)
        self.attn = attn_class(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_bias=proj_bias)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = ffn_layer(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop, bias=ffn_bias)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.sample_drop_ratio = drop_path

    def forward(self, x):
        from hip_ext import functional as HF
        x = x + self.ls1(self.attn(HF.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)))
        return x + self.ls2(self.mlp(HF.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)))


class NestedTensorBlock(Block):
    def forward(self, x_or_x_list):
        if isinstance(x_or_x_list, (list, tuple)):
            raise NotImplementedError("nested-tensor inputs need xformers in the reference too (block.py:249-250)")
        return super().forward(x_or_x_list)
