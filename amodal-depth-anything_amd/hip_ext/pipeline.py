"""The two-model amodal-depth pipeline of infer.py kept on the device.

    base = raw Depth-Anything-V2(normalised rgb)            reference infer.py:16-20
    norm = (base - min) / (max - min)   per image           infer.py:22          ada_minmax_fwd + ada_normalize_fwd
    pred = AmodalDAv2(rgb, mask*2-1, norm*2-1)              infer.py:88-93
    out  = paste pred inside the mask, 3x3 box blur on the mask border      infer.py:30-44   ada_blend_fwd

The reference moves `base` to the host, normalises it in numpy, sends it back, and blends on the host again; here the
maps never leave HBM and the only host traffic is the final result.
"""
from __future__ import annotations

import torch

from . import HipExtError, blend, minmax, normalize


@torch.no_grad()
def amodal_depth_pipeline(model_raw, amodal_model, rgb: torch.Tensor, mask01: torch.Tensor, rgb_raw: torch.Tensor = None):
    """rgb: [B,3,H,W] fp32 in [0,1] on the HIP device, the image fed to the amodal network; rgb_raw: the image fed to the
    base-depth network (the reference resizes that one bilinearly and the other with nearest: infer.py:17,84) -- defaults
    to rgb.  mask01: [B,1,H,W] fp32 0/1.  Returns (base_norm [B,H,W], amodal_pred [B,H,W], blended [B,H,W]) on the device."""
    if not rgb.is_cuda:
        raise HipExtError("amodal_depth_pipeline: inputs must live on a HIP device")
    B, _, H, W = rgb.shape
    src = rgb if rgb_raw is None else rgb_raw
    # the caller-side ImageNet normalisation of infer.py:19 runs inside the raw model's patchify kernel (normalise_input): no extra pass
    base = model_raw(src.contiguous(), normalise_input=True).contiguous()     # [B,H,W] >= 0
    mm = torch.empty(B, 2, dtype=torch.float32, device=rgb.device)
    minmax(base, mm)
    base_norm = torch.empty_like(base)
    obs = torch.empty(B, 1, H, W, dtype=torch.float32, device=rgb.device)
    normalize(base, mm, norm=base_norm, obs=obs)
    mask01 = mask01.float().contiguous()
    pred = amodal_model(rgb, guide_rgb=None, guide_mask=mask01 * 2 - 1, observation=obs).reshape(B, H, W).contiguous()
    out = torch.empty_like(base_norm)
    blend(pred, base_norm, mask01.reshape(B, H, W).contiguous(), out)
    return base_norm, pred, out
