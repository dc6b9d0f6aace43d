"""Host-side image helpers used by infer.py (reference src/util/image_util.py:12-59 and the cv2 / torchvision
calls of reference infer.py).  cv2 and torchvision are not dependencies of this build: the few operations the CLI
needs are restated with numpy / PIL / torch.  None of this is on the accelerated path."""
import matplotlib
import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image


def colorize_depth_maps(depth_map, min_depth, max_depth, cmap="Spectral", valid_mask=None):
    """[(B,)H,W] depth -> [B,3,H,W] colours in [0,1] (reference image_util.py:12-50)."""
    assert len(depth_map.shape) >= 2, "Invalid dimension"
    is_tensor = isinstance(depth_map, torch.Tensor)
    depth = depth_map.detach().squeeze().cpu().numpy() if is_tensor else np.asarray(depth_map).copy().squeeze()
    if depth.ndim < 3:
        depth = depth[np.newaxis]
    depth = ((depth - min_depth) / (max_depth - min_depth)).clip(0, 1)
    col = np.rollaxis(matplotlib.colormaps[cmap](depth, bytes=False)[..., :3], 3, 1)
    if valid_mask is not None:
        vm = valid_mask.detach().cpu().numpy() if isinstance(valid_mask, torch.Tensor) else np.asarray(valid_mask)
        vm = vm.squeeze()
        vm = vm[np.newaxis, np.newaxis] if vm.ndim < 3 else vm[:, np.newaxis]
        col[~np.repeat(vm, 3, axis=1)] = 0
    return torch.from_numpy(col).float() if is_tensor else col


def chw2hwc(chw):
    assert len(chw.shape) == 3
    return chw.permute(1, 2, 0) if isinstance(chw, torch.Tensor) else np.moveaxis(chw, 0, -1)


def imread_bgr(path):
    """cv2.imread equivalent: uint8 HxWx3 in BGR order (the reference feeds BGR to the network: infer.py:75,82)."""
    return np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1].copy()


def imwrite_bgr(path, img_bgr):
    Image.fromarray(np.ascontiguousarray(img_bgr[:, :, ::-1])).save(path)


def resize_bilinear_u8(img, width, height):
    """cv2.resize(img, (width, height)) with the default INTER_LINEAR (half-pixel centres, no anti-aliasing)."""
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).unsqueeze(0).float()
    t = F.interpolate(t, size=(height, width), mode="bilinear", align_corners=False, antialias=False)
    return t.round().clamp(0, 255).byte().squeeze(0).permute(1, 2, 0).numpy()


def resize_nearest(img, width, height):
    """cv2.resize(..., interpolation=INTER_NEAREST) on HxW or HxWxC arrays."""
    h, w = img.shape[:2]
    ys = np.minimum((np.arange(height) * (h / height)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(width) * (w / width)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def box_blur(img, k=3):
    """cv2.blur(img, (k, k)): normalised box filter with reflect-101 borders."""
    r = k // 2
    p = np.pad(img, r, mode="reflect")
    out = np.zeros_like(img, dtype=np.float64)
    for dy in range(k):
        for dx in range(k):
            out += p[dy:dy + img.shape[0], dx:dx + img.shape[1]]
    return (out / (k * k)).astype(img.dtype)


def draw_mask_outline(img_hwc, mask, thickness=2, color=(0, 0, 0)):
    """Stand-in for cv2.findContours + drawContours(..., thickness=2): paints the mask boundary."""
    m = mask > 0
    p = np.pad(m, 1, mode="edge")
    interior = p[:-2, 1:-1] & p[2:, 1:-1] & p[1:-1, :-2] & p[1:-1, 2:] & m
    edge = m & ~interior
    for _ in range(max(thickness - 1, 0)):
        q = np.pad(edge, 1)
        edge = q[:-2, 1:-1] | q[2:, 1:-1] | q[1:-1, :-2] | q[1:-1, 2:] | edge
    out = img_hwc.copy()
    out[edge] = np.asarray(color, dtype=out.dtype)
    return out
