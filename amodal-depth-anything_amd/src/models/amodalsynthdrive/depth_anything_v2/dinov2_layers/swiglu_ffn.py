"""SwiGLU FFN of ViT-giant2 (reference DA2/dinov2_layers/swiglu_ffn.py:13-63):
w3(silu(x1) * x2) with [x1, x2] = w12(x); hidden = (int(h * 2 / 3) + 7) // 8 * 8 in the fused flavour.
The gate is computed in the w12 GEMM epilogue (ADA_EP_SWIGLU), so the 2*hidden tensor never exists."""
from torch import nn


class SwiGLUFFN(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=None, drop=0.0, bias=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.w12 = nn.Linear(in_features, 2 * hidden_features, bias=bias)
        self.w3 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x):
        from hip_ext import functional as HF
        h = HF.swiglu_linear(x, self.w12.weight, self.w12.bias)
        return HF.linear(h, self.w3.weight, self.w3.bias)


class SwiGLUFFNFused(SwiGLUFFN):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=None, drop=0.0, bias=True):
        hidden_features = hidden_features or in_features
        hidden_features = (int(hidden_features * 2 / 3) + 7) // 8 * 8
        super().__init__(in_features, hidden_features, out_features or in_features, bias=bias)
