"""AmodalDAv2 -- the amodal depth model wrapper (reference src/models/amodalsynthdrive/dav2.py:21-90).

Same constructor, ``forward`` signature, ``state_dict`` keys and hub-mixin persistence as the
reference; the arithmetic runs in libada_hip through the guided ``DepthAnythingV2`` engine.
"""
import torch
import torch.nn as nn
from huggingface_hub import PyTorchModelHubMixin
from huggingface_hub.constants import SAFETENSORS_SINGLE_FILE

from .depth_anything_v2.dpt import DepthAnythingV2

# reference dav2.py:31-34 (no 'vitg' entry there either: SURVEY.md §0.3)
MODEL_CONFIGS = {
    "vits": {"encoder": "vits", "features": 64, "out_channels": [48, 96, 192, 384]},
    "vitb": {"encoder": "vitb", "features": 128, "out_channels": [96, 192, 384, 768]},
    "vitl": {"encoder": "vitl", "features": 256, "out_channels": [256, 512, 1024, 1024]},
}
GUIDE_TYPES = ("image+mask+observation", "image+mask", "image+observation", "mask+observation", "observation", "mask", "none")


class AmodalDAv2(nn.Module, PyTorchModelHubMixin):
    def __init__(self, guide_type="image+mask", loss_stategy="invisible_part", encoder="vitg", pretrained=True):
        super().__init__()
        self.guide_type = guide_type
        cfg = MODEL_CONFIGS[encoder]  # KeyError for 'vitg', as in the reference
        self.encoder = DepthAnythingV2(encoder=encoder, features=cfg["features"], out_channels=cfg["out_channels"],
                                       guide_type=guide_type, loss_stategy=loss_stategy)
        self.encoder.normalise_input = True  # (x - mean) / std of dav2.py:65 is fused into the patchify kernel
        self.pretrained = pretrained
        # kept as non-persistent buffers so state_dict() matches the reference (dav2.py:50-51)
        self.register_buffer("pixel_mean", torch.tensor([0.485, 0.456, 0.406]).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor([0.229, 0.224, 0.225]).view(-1, 1, 1), False)
        if guide_type != "none":  # guidance embedding starts at zero (dav2.py:55-61)
            proj = self.encoder.pretrained.patch_embed_guidance.proj
            nn.init.zeros_(proj.weight)
            nn.init.zeros_(proj.bias)

    def build_guide(self, guide_rgb, guide_mask, observation):
        """Channel concatenation selected by guide_type (dav2.py:67-82)."""
        parts = {"image": guide_rgb, "mask": guide_mask, "observation": observation}
        if self.guide_type == "none":
            return None
        if self.guide_type not in GUIDE_TYPES:
            raise NotImplementedError
        sel = [parts[name] for name in self.guide_type.split("+")]
        return sel[0] if len(sel) == 1 else torch.cat(sel, dim=1)

    def forward(self, x, guide_rgb=None, guide_mask=None, observation=None):
        return self.encoder(x, self.build_guide(guide_rgb, guide_mask, observation))

    def _save_pretrained(self, save_directory) -> None:
        from safetensors.torch import save_model
        save_model(self.module if hasattr(self, "module") else self, str(save_directory / SAFETENSORS_SINGLE_FILE))
