"""Unguided Depth-Anything-V2 -- the base-depth model of infer.py (reference RAW/dpt.py:153-184).
``DepthAnythingV2(encoder='vitg', ...).forward(x) -> [B,H,W]`` with x already ImageNet-normalised by the
caller (reference infer.py:19); head ends in ReLU and forward applies F.relu + squeeze(1) again
(RAW/dpt.py:182-184).  Shares every layer class with the guided package: the reference keeps a verbatim
second copy of the tree, which differs only in the guidance embed, input_projection and the tail."""
import torch.nn as nn

from ..depth_anything_v2.dinov2 import DINOv2 as _DINOv2
from ..depth_anything_v2.dpt import INTERMEDIATE_LAYER_IDX, DPTHead as _DPTHead, _EngineMixin


def DINOv2(model_name):
    return _DINOv2(model_name, guide_type=None)


class DPTHead(_DPTHead):
    def __init__(self, in_channels, features=256, use_bn=False, out_channels=(256, 512, 1024, 1024), use_clstoken=False):
        super().__init__(in_channels, features, use_bn, out_channels, use_clstoken, loss_stategy="", with_input_projection=False)


class DepthAnythingV2(_EngineMixin, nn.Module):
    def __init__(self, encoder="vitg", features=256, out_channels=(256, 512, 1024, 1024), use_bn=False, use_clstoken=False):
        super().__init__()
        self.intermediate_layer_idx = INTERMEDIATE_LAYER_IDX
        self.encoder = encoder
        self.pretrained = DINOv2(model_name=encoder)
        self.depth_head = DPTHead(self.pretrained.embed_dim, features, use_bn, out_channels=out_channels, use_clstoken=use_clstoken)
        self.normalise_input = False

    def forward(self, x, normalise_input=None):
        """``x``: ImageNet-normalised image (reference RAW/dpt.py:176-184, infer.py:19).  ``normalise_input=True`` takes the [0, 1] image
        instead and applies (x - mean) / std inside the patchify kernel (the on-device pipeline uses it: no separate normalise pass)."""
        depth = self._run(x, None, normalise_input)   # tail ReLU of the head is fused; relu(relu(x)) == relu(x)
        return depth.squeeze(1)

    def forward_modular(self, x):
        ph, pw = x.shape[-2] // 14, x.shape[-1] // 14
        feats = self.pretrained.get_intermediate_layers(x, self.intermediate_layer_idx[self.encoder], return_class_token=True)
        return self.depth_head(feats, ph, pw).squeeze(1)
