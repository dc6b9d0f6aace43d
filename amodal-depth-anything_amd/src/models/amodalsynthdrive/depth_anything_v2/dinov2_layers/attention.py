"""Attention / MemEffAttention (reference DA2/dinov2_layers/attention.py:29-81).

Both reference classes compute softmax(q k^T / sqrt(d)) v; MemEffAttention only swaps in xformers'
fused kernel when it is installed.  Here both run the same fused HIP attention kernel (the
N x N score matrix is never materialised); head_dim must be 64, which holds for every DINOv2 size."""
from torch import nn


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, proj_bias=True, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)  # parameter containers only
        self.proj = nn.Linear(dim, dim, bias=proj_bias)

    def forward(self, x, attn_bias=None):
        assert attn_bias is None, "nested-tensor attention bias is a training-only feature"
        from hip_ext import functional as HF
        return HF.self_attention(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, self.num_heads)


class MemEffAttention(Attention):
    pass
