#!/usr/bin/env python
"""Amodal-Depth-Anything inference CLI -- same flags, pipeline and output files as the reference's infer.py
(reference infer.py:16-141), running both networks on the MI355X-native HIP path:

    python infer.py --input_image_path IMG --input_mask_path MASK --output_folder OUT

    base depth  : raw Depth-Anything-V2 (ViT-G by default) on the 518x518 resized image          (infer.py:16-28)
    amodal depth: AmodalDAv2 ViT-L on image + amodal mask (+-1) + normalised base depth (+-1)      (infer.py:88-93)
    blend       : paste amodal depth inside the mask, 3x3 box-blur on the mask border             (infer.py:30-44)
    outputs     : {name}_raw_depth_rendered.png, {name}_amodal_depth_rendered.png                 (infer.py:118-119)

Differences forced by the environment: the reference hard-codes .cuda() and downloads weights from
the HF hub; here --device selects the device, --amodal_weights / --raw_weights load local checkpoints, and without
them deterministic synthetic weights are used (with a warning) so the CLI is runnable offline.
"""
import argparse
import copy
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(ROOT, "amodal-depth-anything_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from src.util.image_util import (box_blur, chw2hwc, colorize_depth_maps, draw_mask_outline, imread_bgr, imwrite_bgr,  # noqa: E402
                                 resize_bilinear_u8, resize_nearest)

MEAN = torch.tensor([0.485, 0.456, 0.406]).view(-1, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(-1, 1, 1)


def predict_base_depth(input_image_raw, model_raw, device):
    img = resize_bilinear_u8(input_image_raw, 518, 518)
    x = torch.tensor(img).permute(2, 0, 1).unsqueeze(0) / 255
    x = (x - MEAN) / STD
    with torch.no_grad():
        depth_raw = model_raw(x.to(device)).unsqueeze(1).detach().cpu()
    depth_raw = F.interpolate(depth_raw, (518, 518), mode="nearest")
    depth_raw = ((depth_raw - depth_raw.min()) / (depth_raw.max() - depth_raw.min())).squeeze()
    colored = (colorize_depth_maps(depth_raw.numpy(), 0, 1, cmap="Spectral_r").squeeze() * 255).astype(np.uint8)
    return depth_raw, chw2hwc(colored)


def median_filter_blend(depth_amodal_post, depth_agg, mask, filter_width=3):
    """Paste the amodal prediction inside the mask; box-blur (cv2.blur in the reference, despite the name) on the border."""
    mask = torch.as_tensor(mask)
    blended = depth_agg.clone()
    blended[mask > 0] = depth_amodal_post[mask > 0]
    kernel = torch.ones(1, 1, filter_width, filter_width)
    dil = F.conv2d(mask.float()[None, None], kernel, padding=filter_width // 2)
    border = ((dil > 0) & (dil < filter_width ** 2)).squeeze().numpy()
    arr = blended.numpy().copy()
    arr[border] = box_blur(arr, filter_width)[border]
    return torch.tensor(arr)


def highlight_target(depth_colored_hwc, mask, alpha=0.0):
    fg = np.full_like(depth_colored_hwc, (200, 200, 200), dtype=np.uint8)
    out = np.where(mask[..., None] == 0, (1 - alpha) * depth_colored_hwc + alpha * fg, depth_colored_hwc).astype(np.uint8)
    return draw_mask_outline(out, mask, thickness=2, color=(0, 0, 0))


def _synthetic(model, what):
    from src.util.synth_weights import fill_state_dict_
    warnings.warn(f"no checkpoint given for the {what}: using deterministic SYNTHETIC weights (outputs are not meaningful depth)")
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    fill_state_dict_(sd, 0)
    model.load_state_dict(sd, strict=True)
    return model


def load_models(device="cuda", raw_weights=None, amodal_weights=None, raw_encoder="vitg", amodal_encoder="vitl"):
    from src.models import get_model
    from src.models.amodalsynthdrive.depth_anything_v2_raw.dpt import DepthAnythingV2
    raw_cfg = {"vits": (64, [48, 96, 192, 384]), "vitb": (128, [96, 192, 384, 768]), "vitl": (256, [256, 512, 1024, 1024]),
               "vitg": (384, [1536, 1536, 1536, 1536])}[raw_encoder]
    model_raw = DepthAnythingV2(encoder=raw_encoder, features=raw_cfg[0], out_channels=raw_cfg[1])
    if raw_weights:
        model_raw.load_state_dict(torch.load(raw_weights, map_location="cpu"), strict=False)
    else:
        _synthetic(model_raw, "base-depth model")
    model_raw.to(device).eval()

    if amodal_weights:
        cls = get_model("AmodalDAv2", encoder=amodal_encoder, pretrained=False).__class__
        amodal = cls.from_pretrained(amodal_weights, strict=True)  # local directory with config.json + model.safetensors
    else:
        amodal = _synthetic(get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="entire_target_object",
                                      encoder=amodal_encoder, pretrained=False), "amodal model")
    amodal.to(device).eval()
    return model_raw, amodal


def _on_device_pipeline(image_bgr, amodal_mask, model_raw, depth_amodal_model, device):
    """Both networks + min-max normalise + blend without leaving the GPU (hip_ext.pipeline); same arithmetic as the host path."""
    from hip_ext.pipeline import amodal_depth_pipeline
    img518 = resize_bilinear_u8(image_bgr, 518, 518)
    rgb_raw = (torch.tensor(img518).permute(2, 0, 1).unsqueeze(0) / 255).to(device)
    rgb = F.interpolate(torch.tensor(image_bgr).unsqueeze(0).permute(0, 3, 1, 2) / 255, size=(518, 518), mode="nearest").float().to(device)
    mask_ts = (F.interpolate(torch.tensor(amodal_mask).float()[None, None], size=(518, 518), mode="nearest") > 0).float().to(device)
    base_norm, _, blended = amodal_depth_pipeline(model_raw, depth_amodal_model, rgb, mask_ts, rgb_raw=rgb_raw)
    return base_norm[0].cpu(), blended[0].cpu()


def infer_single_image(input_image_path, input_mask_path, output_path, model_raw, depth_amodal_model, device="cuda"):
    file_name = os.path.basename(input_image_path).split(".")[0]
    os.makedirs(output_path, exist_ok=True)
    image_bgr = imread_bgr(input_image_path)
    h0, w0 = image_bgr.shape[:2]
    if str(device).startswith("cuda"):
        amodal_mask = np.asarray(Image.open(input_mask_path)) > 0
        if amodal_mask.ndim == 3:
            amodal_mask = amodal_mask.any(-1)
        base_depth, depth_agg = _on_device_pipeline(image_bgr, amodal_mask, model_raw, depth_amodal_model, device)
        raw_colored = (colorize_depth_maps(base_depth.numpy(), 0, 1, cmap="Spectral_r").squeeze() * 255).astype(np.uint8)
        raw_colored_hwc = resize_nearest(chw2hwc(raw_colored), w0, h0)
        mask518 = (F.interpolate(torch.tensor(amodal_mask).float()[None, None], (518, 518)).squeeze().numpy() > 0).astype(np.uint8) * 255
        agg_colored = (colorize_depth_maps(depth_agg.numpy(), 0, 1, cmap="Spectral_r").squeeze() * 255).astype(np.uint8)
        agg_colored_hwc = resize_nearest(highlight_target(chw2hwc(agg_colored), mask518), w0, h0)
        raw_out, agg_out = raw_colored_hwc[:, :, [2, 1, 0]], agg_colored_hwc[:, :, [2, 1, 0]]
        imwrite_bgr(os.path.join(output_path, f"{file_name}_raw_depth_rendered.png"), raw_out)
        imwrite_bgr(os.path.join(output_path, f"{file_name}_amodal_depth_rendered.png"), agg_out)
        return raw_out, agg_out
    base_depth, raw_colored_hwc = predict_base_depth(image_bgr, model_raw, device)
    raw_colored_hwc = resize_nearest(raw_colored_hwc, w0, h0)

    amodal_mask = np.asarray(Image.open(input_mask_path)) > 0
    if amodal_mask.ndim == 3:
        amodal_mask = amodal_mask.any(-1)
    rgb = torch.tensor(image_bgr).unsqueeze(0).permute(0, 3, 1, 2) / 255
    rgb = F.interpolate(rgb, size=(518, 518), mode="nearest")            # torchvision Resize(NEAREST)
    mask_ts = F.interpolate(torch.tensor(amodal_mask).float()[None, None], size=(518, 518), mode="nearest")
    mask_ts = (mask_ts > 0).float()
    with torch.no_grad():
        pred = depth_amodal_model(rgb.float().to(device), guide_rgb=None, guide_mask=(mask_ts.to(device) * 2) - 1,
                                  observation=(base_depth[None, None].to(device) * 2) - 1)
    pred = pred.detach().cpu()

    depth_raw_post = base_depth.clone()
    depth_amodal_post = pred.squeeze()
    mask518 = (F.interpolate(torch.tensor(amodal_mask).float()[None, None], (518, 518)).squeeze().numpy() > 0).astype(np.uint8) * 255
    depth_agg = median_filter_blend(depth_amodal_post, copy.deepcopy(depth_raw_post), mask518 / 255)
    agg_colored = (colorize_depth_maps(depth_agg.numpy(), 0, 1, cmap="Spectral_r").squeeze() * 255).astype(np.uint8)
    agg_colored_hwc = highlight_target(chw2hwc(agg_colored), mask518)
    agg_colored_hwc = resize_nearest(agg_colored_hwc, w0, h0)

    raw_out = raw_colored_hwc[:, :, [2, 1, 0]]
    agg_out = agg_colored_hwc[:, :, [2, 1, 0]]
    imwrite_bgr(os.path.join(output_path, f"{file_name}_raw_depth_rendered.png"), raw_out)
    imwrite_bgr(os.path.join(output_path, f"{file_name}_amodal_depth_rendered.png"), agg_out)
    return raw_out, agg_out


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Amodal-Depth-Anything inference (MI355X-native HIP path)")
    parser.add_argument("--input_image_path", type=str, help="Path to the input image.")
    parser.add_argument("--input_mask_path", type=str, help="Path to the amodal mask image.")
    parser.add_argument("--output_folder", type=str, help="Output folder.")
    parser.add_argument("--device", type=str, default="cuda")
    parser.add_argument("--raw_weights", type=str, default=None, help="local .pth of the base Depth-Anything-V2 model")
    parser.add_argument("--amodal_weights", type=str, default=None, help="local directory with config.json + model.safetensors")
    parser.add_argument("--raw_encoder", type=str, default="vitg")
    parser.add_argument("--amodal_encoder", type=str, default="vitl")
    args = parser.parse_args()
    m_raw, m_amodal = load_models(args.device, args.raw_weights, args.amodal_weights, args.raw_encoder, args.amodal_encoder)
    infer_single_image(args.input_image_path, args.input_mask_path, args.output_folder, m_raw, m_amodal, args.device)
