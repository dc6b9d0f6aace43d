"""Mlp (reference DA2/dinov2_layers/mlp.py:16-41): fc1 -> exact-erf GELU -> fc2 (dropouts are p=0).
fc1 + bias + GELU is one launch (ADA_EP_GELU epilogue); fc2 + bias another."""
from torch import nn


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, bias=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)  # parameter containers only
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x):
        from hip_ext import functional as HF
        h = HF.linear(x, self.fc1.weight, self.fc1.bias, gelu=True, out_operand=True)
        return HF.linear(h, self.fc2.weight, self.fc2.bias)
