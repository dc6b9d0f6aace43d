"""DPT decoder blocks (reference DA2/util/blocks.py): scratch convs, ResidualConvUnit, FeatureFusionBlock.
The nn.Conv2d members are parameter containers; compute goes through the implicit-GEMM HIP kernel."""
from torch import nn


def _make_scratch(in_shape, out_shape, groups=1, expand=False):
    if groups != 1 or expand:
        raise NotImplementedError("grouped / expanding scratch layers are not used by Depth-Anything-V2")
    scratch = nn.Module()
    for i, cin in enumerate(in_shape[:4]):
        setattr(scratch, f"layer{i + 1}_rn", nn.Conv2d(cin, out_shape, kernel_size=3, stride=1, padding=1, bias=False))
    return scratch


class ResidualConvUnit(nn.Module):
    """conv2(relu(conv1(relu(x)))) + x -- the ReLU is not in place, so the skip adds the pre-activation x.
    bn=True (reference blocks.py:49-51,70-76; no shipped configuration) puts a BatchNorm2d behind each conv: inference uses the running
    statistics, i.e. a per-channel affine map that is folded into the conv's weight and bias before they are packed."""

    def __init__(self, features, activation, bn):
        super().__init__()
        self.bn, self.groups, self.activation = bn, 1, activation
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        if bn:
            self.bn1 = nn.BatchNorm2d(features)
            self.bn2 = nn.BatchNorm2d(features)

    def folded(self):
        """(w1, b1, w2, b2) with the BatchNorms (if any) folded in."""
        from hip_ext.functional import fold_batchnorm
        w1, b1, w2, b2 = self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias
        if self.bn:
            if self.training:
                raise NotImplementedError("BatchNorm fusion blocks are inference-only here (running statistics); call .eval()")
            w1, b1 = fold_batchnorm(w1, b1, self.bn1.weight, self.bn1.bias, self.bn1.running_mean, self.bn1.running_var, self.bn1.eps)
            w2, b2 = fold_batchnorm(w2, b2, self.bn2.weight, self.bn2.bias, self.bn2.running_mean, self.bn2.running_var, self.bn2.eps)
        return w1, b1, w2, b2

    def forward(self, x):
        from hip_ext import functional as HF
        return HF.residual_conv_unit(x, *self.folded())


class FeatureFusionBlock(nn.Module):
    def __init__(self, features, activation, deconv=False, bn=False, expand=False, align_corners=True, size=None):
        super().__init__()
        if deconv or expand or not align_corners:
            raise NotImplementedError("only the Depth-Anything-V2 configuration (align_corners=True, no expand) is built")
        self.deconv, self.align_corners, self.groups, self.expand, self.size = deconv, align_corners, 1, expand, size
        self.out_conv = nn.Conv2d(features, features, kernel_size=1, stride=1, padding=0, bias=True)
        self.resConfUnit1 = ResidualConvUnit(features, activation, bn)
        self.resConfUnit2 = ResidualConvUnit(features, activation, bn)

    def forward(self, *xs, size=None):
        from hip_ext import functional as HF
        out = xs[0]
        if len(xs) == 2:
            out = out + self.resConfUnit1(xs[1])
        out = self.resConfUnit2(out)
        if size is None and self.size is None:
            size = (out.shape[-2] * 2, out.shape[-1] * 2)
        elif size is None:
            size = self.size
        out = HF.interpolate_bilinear_ac(out, tuple(size))
        return HF.conv2d(out, self.out_conv.weight, self.out_conv.bias)
