"""Deterministic synthetic weights / inputs for the Amodal-Depth-Anything forward pass.

No checkpoint ships with the reference and there is no network, so parity and
throughput are measured on weights produced by this generator.  It is bit-reproducible across
machines running the same torch build: every tensor is drawn from its own CPU ``torch.Generator``
seeded with crc32(key) ^ seed, so the result does not depend on dict order or on which other
keys exist.  It works on *any* state_dict with the reference's key schema --
the reference's, the oracle's or this package's -- which is what lets golden vectors produced
from the reference in the build container be reproduced on the GPU box.

The fill deliberately (a) overwrites the zero-init of ``patch_embed_guidance`` (reference
``src/models/amodalsynthdrive/dav2.py:55-61``) so the mask/observation conditioning is live,
(b) moves LayerNorm / LayerScale gains away from 1, and (c) widens the head's output range so a
relative-L1 comparison of depth maps is meaningful.
"""
import zlib
from typing import Dict

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _is_norm_key(key: str) -> bool:
    leaf = key.rsplit(".", 2)
    if len(leaf) < 2:
        return False
    parent = leaf[-2]
    if parent in ("norm", "norm1", "norm2"):
        return True
    # input_projection.{i}.1 is the channels-first LayerNorm (reference DA2/dpt.py:153-159)
    return "input_projection" in key and parent == "1"


# "heavy" fill: DINOv2-style outlier features.  Real DINOv2 checkpoints carry a handful of hidden dimensions whose residual-stream
# magnitude is 1-2 orders above the rest ("massive activations"); the standard fill above has none (every activation is O(1)).  The heavy
# variant multiplies the LayerScale gains (ls1 / ls2 .gamma) of OUTLIER_CHANNELS by OUTLIER_GAIN in the first third of the blocks -- the
# residual stream then carries those channels at ~30-100x the typical magnitude through every later LayerNorm, attention and MLP -- and
# the final-norm gain of the same channels by 4, so the head sees them too.
OUTLIER_CHANNELS = (5, 77, 191, 300)
OUTLIER_GAIN = 40.0


def _apply_outliers_(key: str, v: torch.Tensor) -> torch.Tensor:
    parts = key.split(".")
    if parts[-1] == "gamma" and "blocks" in parts:
        blk = int(parts[parts.index("blocks") + 1])
        if blk < 4:
            idx = [c for c in OUTLIER_CHANNELS if c < v.numel()]
            v[idx] = v[idx] * OUTLIER_GAIN
    elif key.endswith("pretrained.norm.weight"):
        idx = [c for c in OUTLIER_CHANNELS if c < v.numel()]
        v[idx] = v[idx] * 4.0
    return v


@torch.no_grad()
def fill_state_dict_(sd: Dict[str, torch.Tensor], seed: int = 0, tail: str = "normal") -> Dict[str, torch.Tensor]:
    """Overwrites every floating tensor of ``sd`` in place; returns ``sd``.  ``tail``: "normal" | "heavy" (outlier channels, see above)."""
    assert tail in ("normal", "heavy"), tail

    def fill(item):
        key, t = item
        if not torch.is_floating_point(t):
            return
        g = _gen(key, seed)
        shape = tuple(t.shape)
        leaf = key.rsplit(".", 1)[-1]
        # Draw straight into the destination where it is a contiguous fp32 tensor (torch.randn(shape, generator=g) IS empty(shape).normal_(generator=g):
        # same stream, same values) and scale in place: one pass over memory instead of three temporaries -- the ViT-G fill is 4.4 GB and its cost
        # is memory traffic and page faults, not the generator.  Every in-place sequence below performs the same fp32 operations in the same order
        # as the expression it replaces (x * a then + b), so the values are bit-identical (tests/test_synth_weights_cpu.py pins them).
        inplace = t.device.type == "cpu" and t.dtype == torch.float32 and t.is_contiguous() and not t.requires_grad
        v = t if inplace else torch.empty(shape, dtype=torch.float32)

        def randn_():
            return v.normal_(generator=g)

        def rand_():
            return v.uniform_(generator=g)

        if leaf == "running_var":      # BatchNorm2d statistics (use_bn=True heads): strictly positive
            rand_().add_(0.5)
        elif leaf == "running_mean":
            randn_().mul_(0.2)
        elif ".bn1." in key or ".bn2." in key or _is_norm_key(key):
            if leaf == "weight":
                randn_().mul_(0.1).add_(1.0)
            else:
                randn_().mul_(0.1)
        elif leaf == "gamma":  # LayerScale
            rand_().mul_(0.7).add_(0.3)
        elif leaf in ("cls_token", "mask_token", "pos_embed"):
            randn_().mul_(0.5)
        elif t.ndim >= 2:
            if "resize_layers.0" in key or "resize_layers.1" in key:
                fan_in = shape[0]  # ConvTranspose2d [Cin, Cout, k, k], stride == k: Cin terms per output
            else:
                fan_in = 1
                for s_ in shape[1:]:
                    fan_in *= s_
            gain = 1.0
            if "patch_embed_guidance" in key:
                gain = 2.0
            randn_()
            if "output_conv2.2" in key:
                # final 1x1 conv sees post-ReLU (all-positive) features: a zero-mean filter keeps the
                # logits centred so the sigmoid output spans (0,1) instead of saturating at one end.
                v.sub_(v.mean())
                gain = 3.0
            v.mul_(gain / fan_in ** 0.5)
        else:  # biases
            randn_().mul_(0.1)
        if tail == "heavy":
            _apply_outliers_(key, v)
        if not inplace:
            t.copy_(v.to(t.dtype))

    # every tensor has its own generator, so the keys can be filled concurrently (torch releases the GIL inside randn): the ViT-G fixtures
    # (1.1 G parameters) spend most of their test time here
    items = list(sd.items())
    big = sum(t.numel() for _, t in items) > (1 << 26)
    if big:
        from concurrent.futures import ThreadPoolExecutor
        import os as _os
        with ThreadPoolExecutor(max_workers=min(32, _os.cpu_count() or 1)) as ex:
            list(ex.map(fill, items))
    else:
        for it in items:
            fill(it)
    return sd


def _structured_inputs(batch: int, height: int, width: int, g: torch.Generator):
    """Image-like inputs (what the reference's caller actually feeds, infer.py:17-27,88-93): smooth low-frequency RGB with a little
    fine texture, a filled-ellipse amodal mask, and an observation that is a smooth depth-like map min-max normalised to [-1, 1]
    (infer.py:22 ``(d - d.min()) / (d.max() - d.min())`` then ``* 2 - 1``)."""
    yy = torch.linspace(0.0, 1.0, height).view(1, 1, height, 1)
    xx = torch.linspace(0.0, 1.0, width).view(1, 1, 1, width)

    def smooth(channels):   # sum of four low-frequency plane waves + a linear ramp per channel
        f = torch.zeros(batch, channels, height, width)
        for _ in range(4):
            fy = torch.rand(batch, channels, 1, 1, generator=g) * 3.0
            fx = torch.rand(batch, channels, 1, 1, generator=g) * 3.0
            ph = torch.rand(batch, channels, 1, 1, generator=g) * 6.2831853
            amp = 0.15 + 0.25 * torch.rand(batch, channels, 1, 1, generator=g)
            f = f + amp * torch.sin(6.2831853 * (fy * yy + fx * xx) + ph)
        ry = torch.rand(batch, channels, 1, 1, generator=g) - 0.5
        rx = torch.rand(batch, channels, 1, 1, generator=g) - 0.5
        return f + ry * yy + rx * xx

    def rgb():
        tex = 0.03 * (torch.rand(batch, 3, height, width, generator=g) - 0.5)   # sensor-noise-sized texture
        return (0.5 + 0.6 * smooth(3) + tex).clamp_(0.0, 1.0)

    x, guide_rgb = rgb(), rgb()
    d = smooth(1)
    lo = d.amin(dim=(2, 3), keepdim=True)
    hi = d.amax(dim=(2, 3), keepdim=True)
    obs = (d - lo) / (hi - lo) * 2 - 1
    mask = -torch.ones(batch, 1, height, width)
    for b in range(batch):
        cy = 0.25 + 0.5 * float(torch.rand(1, generator=g))
        cx = 0.25 + 0.5 * float(torch.rand(1, generator=g))
        ay = 0.08 + 0.25 * float(torch.rand(1, generator=g))
        ax = 0.08 + 0.25 * float(torch.rand(1, generator=g))
        inside = ((yy - cy) / ay) ** 2 + ((xx - cx) / ax) ** 2 <= 1.0
        mask[b][inside[0]] = 1.0
    return x, guide_rgb, mask, obs


def make_inputs(batch: int, height: int = 518, width: int = 518, seed: int = 0, device="cpu", style: str = "noise"):
    """Synthetic inputs of SURVEY.md §8(d).  ``style="noise"``: i.i.d. uniform RGB in [0,1), rectangular amodal mask (+-1), i.i.d. uniform
    observation in [-1,1].  ``style="structured"``: image-like inputs (_structured_inputs).  ``"zeros"`` / ``"checker"``: degenerate inputs (constant;
    per-pixel checkerboard) -- every token equal or periodic, so operand rounding errors add coherently over positions instead of averaging out.  Returns (x, guide_rgb, guide_mask, observation)."""
    assert style in ("noise", "structured", "zeros", "checker"), style
    if style == "zeros":      # the all-zero image / mask / observation (a constant input: every patch token equal up to its position)
        z = torch.zeros(batch, 3, height, width, device=device)
        return z, z.clone(), z[:, :1].clone(), z[:, :1].clone()
    if style == "checker":    # per-pixel 0/1 checkerboard image and mask, the complementary board as observation
        yy, xx = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
        cb = ((yy + xx) % 2).float().to(device)
        return (cb.expand(batch, 3, height, width).clone(), cb.expand(batch, 3, height, width).clone(), cb.expand(batch, 1, height, width).clone(),
                (1 - cb).expand(batch, 1, height, width).clone())
    g = torch.Generator(device="cpu")
    g.manual_seed(1000003 * seed + 17)
    if style == "structured":
        x, guide_rgb, mask, obs = _structured_inputs(batch, height, width, g)
        return x.to(device), guide_rgb.to(device), mask.to(device), obs.to(device)
    x = torch.rand(batch, 3, height, width, generator=g)
    guide_rgb = torch.rand(batch, 3, height, width, generator=g)
    obs = torch.rand(batch, 1, height, width, generator=g) * 2 - 1
    mask = -torch.ones(batch, 1, height, width)
    for b in range(batch):
        y0 = int(torch.randint(0, height // 2, (1,), generator=g))
        x0 = int(torch.randint(0, width // 2, (1,), generator=g))
        hh = int(torch.randint(height // 8, height // 2, (1,), generator=g))
        ww = int(torch.randint(width // 8, width // 2, (1,), generator=g))
        mask[b, :, y0:y0 + hh, x0:x0 + ww] = 1.0
    return x.to(device), guide_rgb.to(device), mask.to(device), obs.to(device)


def centred_final_bias(encoder: str, repo_root: str):
    """The logit-centring value of ``output_conv2.2.bias`` that the reference fixtures of the seed-0 fill carry for AmodalDAv2 (guide mask+observation):
    tests/golden/<encoder>_518.npz metadata.  Tools that time the synthetic model use it so that the depth maps span (0, 1) -- the un-centred fill's maps
    sit near 0, where the sigmoid heads' precision ladder (rightly) re-runs the head in split precision and the timing is not the default path's."""
    import json
    import os

    import numpy as np
    name = {"vits": "vits_518", "vitb": "vitb_518", "vitl": "vitl_518"}.get(encoder)
    path = os.path.join(repo_root, "tests", "golden", f"{name}.npz") if name else None
    if not path or not os.path.exists(path):
        return None
    meta = json.loads(str(np.load(path)["meta"]))
    return meta["final_bias_key"], float(meta["final_bias"])
