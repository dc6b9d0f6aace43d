"""Least-squares scale/shift alignment of a relative depth prediction to the ground truth (``src/util/alignment.py:7-54``).

The reference gathers the valid pixels on the host and calls ``numpy.linalg.lstsq`` on the [n, 2] system; the minimiser is
the solution of the 2x2 normal equations, whose five sums come out of the same device pass as the metrics
(``ada_depth_eval_fwd``: n, sum p, sum g, sum p*p, sum p*g in fp64).  Tensors stay on the device; numpy inputs are moved there.
"""
from typing import Optional

import numpy as np
import torch

import hip_ext as H

__all__ = ["align_depth_least_square", "scale_shift_least_square", "depth2disparity", "disparity2depth"]


def _dev(x, device, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device)


def scale_shift_least_square(gt: torch.Tensor, pred: torch.Tensor, valid_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """[B, 2] fp64 (scale, shift) minimising sum_valid (pred * scale + shift - gt)^2 per image; inputs [B, H, W] on the device."""
    s = H.depth_eval(pred.contiguous().float(), gt.contiguous().float(), None if valid_mask is None else valid_mask.contiguous())
    n, sp, sg, spp, spg = (s[:, i] for i in (H.EVAL_N, H.EVAL_SUM_P, H.EVAL_SUM_G, H.EVAL_SUM_PP, H.EVAL_SUM_PG))
    den = n * spp - sp * sp
    scale = (n * spg - sp * sg) / den
    shift = (sg - scale * sp) / n
    return torch.stack([scale, shift], dim=1)


def align_depth_least_square(gt_arr, pred_arr, valid_mask_arr, return_scale_shift=True, max_resolution=None):
    """Same contract as the reference: one image per call, any leading singleton dims; returns the aligned prediction with the
    input's shape (and type: numpy in -> numpy out) and, optionally, scale and shift."""
    was_numpy = isinstance(pred_arr, np.ndarray)
    device = pred_arr.device if isinstance(pred_arr, torch.Tensor) and pred_arr.is_cuda else torch.device("cuda")
    pred_full = _dev(pred_arr, device, torch.float32)
    ori_shape = pred_full.shape
    gt = _dev(gt_arr, device, torch.float32).squeeze()
    pred = pred_full.squeeze()
    mask = _dev(valid_mask_arr, device).squeeze() != 0
    if max_resolution is not None:   # alignment.py:22-33: nearest down-sampling before the fit
        scale_factor = float(min(max_resolution / ori_shape[-2], max_resolution / ori_shape[-1]))
        if scale_factor < 1:
            down = torch.nn.Upsample(scale_factor=scale_factor, mode="nearest")
            gt = down(gt[None, None])[0, 0]
            pred = down(pred[None, None])[0, 0]
            mask = down(mask[None, None].float())[0, 0] != 0
    assert gt.shape == pred.shape == mask.shape, f"{gt.shape}, {pred.shape}, {mask.shape}"
    ss = scale_shift_least_square(gt[None], pred[None], mask[None])[0]
    scale, shift = ss[0], ss[1]
    aligned = (pred_full.double() * scale + shift).reshape(ori_shape)
    if was_numpy:
        aligned = aligned.cpu().numpy()
        scale, shift = np.array([scale.item()]), np.array([shift.item()])
    if return_scale_shift:
        return aligned, scale, shift
    return aligned


def depth2disparity(depth, return_mask=False):
    """1/depth where depth > 0, else 0 (alignment.py:57-68); works on tensors (any device) and numpy arrays."""
    positive = depth > 0
    if isinstance(depth, torch.Tensor):
        inv = torch.where(positive, 1.0 / depth.clamp_min(torch.finfo(depth.dtype).tiny), torch.zeros_like(depth))
    else:
        depth = np.asarray(depth)
        inv = np.zeros_like(depth)
        np.divide(1.0, depth, out=inv, where=positive)
    return (inv, positive) if return_mask else inv


def disparity2depth(disparity, **kwargs):   # the map is its own inverse (alignment.py:71-72)
    return depth2disparity(disparity, **kwargs)
