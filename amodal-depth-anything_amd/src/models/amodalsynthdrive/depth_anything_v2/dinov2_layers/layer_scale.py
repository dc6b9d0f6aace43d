"""LayerScale (reference DA2/dinov2_layers/layer_scale.py:16-28): per-channel gain ``gamma``.
Inside a Block the multiply is fused into the proj / fc2 GEMM epilogue (ADA_EP_GAMMA)."""
import torch
from torch import nn


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5, inplace=False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(torch.full((dim,), float(init_values)))

    def forward(self, x):
        from hip_ext import functional as HF
        return HF.scale_channels(x, self.gamma)
