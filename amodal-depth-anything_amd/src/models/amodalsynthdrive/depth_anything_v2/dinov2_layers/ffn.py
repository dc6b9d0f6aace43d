"""Feed-forward halves of the transformer block, both flavours of the reference in one place:

* ``Mlp``  -- fc1 -> exact-erf GELU -> fc2 (DA2/dinov2_layers/mlp.py:16-41; the dropouts are p = 0 at inference)
* ``SwiGLUFFN`` / ``SwiGLUFFNFused`` -- w3(silu(x1) * x2) with [x1 | x2] = w12(x) (swiglu_ffn.py:13-63); the fused flavour
  (ViT-giant2) shrinks the hidden width to ``(int(h * 2 / 3) + 7) // 8 * 8``

The modules only own the parameters (same names and shapes as the reference, so its checkpoints load unchanged); the
arithmetic is two HIP launches each: the first GEMM carries the activation in its epilogue (``ADA_EP_GELU`` resp.
``ADA_EP_SWIGLU`` -- the 2*hidden tensor of the SwiGLU never exists), the second one the bias.
"""
from torch import nn


def _widths(in_features, hidden_features, out_features):
    return hidden_features or in_features, out_features or in_features


class _TwoLayerFFN(nn.Module):
    """Parameter container + the two launches; subclasses say how the first layer is named, how wide it is and which epilogue it gets."""
    first = "fc1"
    second = "fc2"
    first_width_factor = 1      # rows of the first weight matrix per hidden unit
    gated = False

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=None, drop=0.0, bias=True):
        super().__init__()
        hidden, out = _widths(in_features, hidden_features, out_features)
        hidden = self._hidden_width(hidden)
        setattr(self, self.first, nn.Linear(in_features, self.first_width_factor * hidden, bias=bias))
        setattr(self, self.second, nn.Linear(hidden, out, bias=bias))

    @staticmethod
    def _hidden_width(hidden):
        return hidden

    def forward(self, x):
        from hip_ext import functional as HF
        l1, l2 = getattr(self, self.first), getattr(self, self.second)
        if self.gated:
            h = HF.swiglu_linear(x, l1.weight, l1.bias)
        else:
            h = HF.linear(x, l1.weight, l1.bias, gelu=True, out_operand=True)
        return HF.linear(h, l2.weight, l2.bias)


class Mlp(_TwoLayerFFN):
    pass


class SwiGLUFFN(_TwoLayerFFN):
    first, second = "w12", "w3"
    first_width_factor = 2
    gated = True


class SwiGLUFFNFused(SwiGLUFFN):
    @staticmethod
    def _hidden_width(hidden):
        return (int(hidden * 2 / 3) + 7) // 8 * 8
