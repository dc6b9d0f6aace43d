"""TEST INFRASTRUCTURE ONLY -- CPU fp32 oracle for the Amodal-Depth-Anything forward pass.

A plain-PyTorch, functional (state_dict in, tensors out) restatement of the reference's
algorithm for the hot path of SURVEY.md §8(a).  It exists to *check* the HIP product path;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product package never does (it has no CPU fallback at all).

Parity pinning: the reference ships no tests or golden vectors for this path,
so the oracle is pinned against the reference *itself*, imported in the build container via
``oracle/_refshim.py``: ``tests/test_oracle_vs_reference.py`` compares them tensor-for-tensor
when /root/reference is present, and ``oracle/make_golden.py`` commits the reference's outputs
as fixtures under ``tests/golden/`` which ``tests/test_oracle_golden.py`` re-checks anywhere.

All citations are relative to /root/reference/; DA2 = src/models/amodalsynthdrive/depth_anything_v2,
RAW = src/models/amodalsynthdrive/depth_anything_v2_raw.

``operand_dtype=torch.bfloat16`` additionally emulates the product's numerics contract
(bf16 GEMM/conv/bmm operands, fp32 accumulate, fp32 everything else) so the precision budget
can be studied on CPU.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------
# configuration tables
# ---------------------------------------------------------------------------------------
# DA2/dinov2.py:367-427 (factories), 430-448 (DINOv2()).
VIT = {
    "vits": dict(dim=384, depth=12, heads=6, ffn="mlp"),
    "vitb": dict(dim=768, depth=12, heads=12, ffn="mlp"),
    "vitl": dict(dim=1024, depth=24, heads=16, ffn="mlp"),
    "vitg": dict(dim=1536, depth=40, heads=24, ffn="swiglu"),
}
# DA2/dpt.py:213-218
TAPS = {"vits": [2, 5, 8, 11], "vitb": [2, 5, 8, 11], "vitl": [4, 11, 17, 23], "vitg": [9, 19, 29, 39]}
# src/models/amodalsynthdrive/dav2.py:31-34
AMODAL_HEAD = {
    "vits": dict(features=64, out_channels=[48, 96, 192, 384]),
    "vitb": dict(features=128, out_channels=[96, 192, 384, 768]),
    "vitl": dict(features=256, out_channels=[256, 512, 1024, 1024]),
}
# DA2/dinov2.py:109-125
GUIDE_CHANNELS = {
    "image+mask+observation": 5, "image+mask": 4, "image+observation": 4,
    "mask+observation": 2, "mask": 1, "observation": 1, "none": 0,
}
PATCH = 14
LN_EPS = 1e-6  # DA2/dinov2.py:96 and DA2/dpt.py:43
PIXEL_MEAN = (0.485, 0.456, 0.406)  # dav2.py:50
PIXEL_STD = (0.229, 0.224, 0.225)   # dav2.py:51


class _Numerics:
    """Rounds contraction operands to ``operand_dtype`` (fp32 = exact reference numerics)."""

    def __init__(self, operand_dtype=torch.float32):
        self.dt = operand_dtype

    def q(self, t):
        return t if self.dt == torch.float32 else t.to(self.dt).to(torch.float32)

    def linear(self, x, w, b=None):
        return F.linear(self.q(x), self.q(w), b)

    def conv(self, x, w, b=None, stride=1, padding=0):
        return F.conv2d(self.q(x), self.q(w), b, stride=stride, padding=padding)

    def convT(self, x, w, b, stride):
        return F.conv_transpose2d(self.q(x), self.q(w), b, stride=stride)

    def matmul(self, a, b):
        return self.q(a) @ self.q(b)


# ---------------------------------------------------------------------------------------
# encoder
# ---------------------------------------------------------------------------------------
def patch_embed(nm, x, w, b):
    """DA2/dinov2_layers/patch_embed.py:69-82: Conv2d(C, D, 14, stride 14) -> flatten(2).T."""
    H, W = x.shape[-2:]
    assert H % PATCH == 0, f"Input image height {H} is not a multiple of patch height {PATCH}"
    assert W % PATCH == 0, f"Input image width {W} is not a multiple of patch width: {PATCH}"
    y = nm.conv(x, w, b, stride=PATCH)
    return y.flatten(2).transpose(1, 2)


def interpolate_pos_encoding(pos_embed, npatch, w, h, offset=0.1):
    """DA2/dinov2.py:199-230.  Identity when the grid is the native 37x37 square one."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    pos = pos_embed.float()
    cls_pos, patch_pos = pos[:, 0], pos[:, 1:]
    dim = pos.shape[-1]
    w0, h0 = w // PATCH + offset, h // PATCH + offset
    sqrt_n = math.sqrt(N)
    sx, sy = float(w0) / sqrt_n, float(h0) / sqrt_n
    grid = patch_pos.reshape(1, int(sqrt_n), int(sqrt_n), dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=(sx, sy), mode="bicubic", antialias=False)
    assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pos.unsqueeze(0), grid), dim=1)


def prepare_tokens(nm, sd, pfx, x, guide):
    """DA2/dinov2.py:232-258 (masks=None, no register tokens)."""
    B, _, w, h = x.shape
    t = patch_embed(nm, x, sd[pfx + "patch_embed.proj.weight"], sd[pfx + "patch_embed.proj.bias"])
    if guide is not None:  # DA2/dinov2.py:237-240
        t = t + patch_embed(nm, guide, sd[pfx + "patch_embed_guidance.proj.weight"],
                            sd[pfx + "patch_embed_guidance.proj.bias"])
    t = torch.cat((sd[pfx + "cls_token"].expand(B, -1, -1), t), dim=1)  # :245
    return t + interpolate_pos_encoding(sd[pfx + "pos_embed"], t.shape[1] - 1, w, h)  # :246


def attention(nm, sd, p, x, heads):
    """DA2/dinov2_layers/attention.py:49-62 (the non-xformers path: q is scaled before QK^T)."""
    B, N, C = x.shape
    d = C // heads
    qkv = nm.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * d ** -0.5, qkv[1], qkv[2]
    attn = nm.matmul(q, k.transpose(-2, -1)).softmax(dim=-1)
    o = nm.matmul(attn, v).transpose(1, 2).reshape(B, N, C)
    return nm.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def ffn(nm, sd, p, x, kind):
    if kind == "mlp":  # DA2/dinov2_layers/mlp.py:35-41, exact-erf GELU (nn.GELU default, :23)
        h = F.gelu(nm.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
        return nm.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    # DA2/dinov2_layers/swiglu_ffn.py:29-33
    x12 = nm.linear(x, sd[p + "w12.weight"], sd[p + "w12.bias"])
    x1, x2 = x12.chunk(2, dim=-1)
    return nm.linear(F.silu(x1) * x2, sd[p + "w3.weight"], sd[p + "w3.bias"])


def block(nm, sd, p, x, heads, kind):
    """DA2/dinov2_layers/block.py:82-88,104-107 (eval branch) + layer_scale.py:27-28."""
    D = x.shape[-1]
    y = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
    x = x + attention(nm, sd, p + "attn.", y, heads) * sd[p + "ls1.gamma"]
    y = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
    return x + ffn(nm, sd, p + "mlp.", y, kind) * sd[p + "ls2.gamma"]


def encoder_taps(nm, sd, pfx, encoder, x, guide, trace=None, return_class_token=False):
    """DA2/dinov2.py:298-308 + 324-349: taps taken *after* block i, shared final LN, cls dropped."""
    cfg = VIT[encoder]
    t = prepare_tokens(nm, sd, pfx, x, guide)
    if trace is not None:
        trace["tokens0"] = t
    D = t.shape[-1]
    outs = []
    for i in range(cfg["depth"]):
        t = block(nm, sd, f"{pfx}blocks.{i}.", t, cfg["heads"], cfg["ffn"])
        if trace is not None and (i == 0 or trace.get("_every_block")):      # "_every_block": the residual stream behind every block (tools/stage_errors.py)
            trace[f"block{i}"] = t
        if i in TAPS[encoder]:
            outs.append(t)
    normed = [F.layer_norm(o, (D,), sd[pfx + "norm.weight"], sd[pfx + "norm.bias"], LN_EPS) for o in outs]
    outs = [o[:, 1:] for o in normed]
    if trace is not None:
        for j, o in enumerate(outs):
            trace[f"tap{j}"] = o
    if return_class_token:     # DA2/dinov2.py:341-349: tuple(zip(patch tokens, class tokens))
        return outs, [o[:, 0] for o in normed]
    return outs


# ---------------------------------------------------------------------------------------
# DPT head
# ---------------------------------------------------------------------------------------
def cf_layernorm(x, w, b):
    """DA2/dpt.py:55-61: channels-first LN, biased variance, eps inside the sqrt."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + LN_EPS)
    return w[:, None, None] * x + b[:, None, None]


def residual_conv_unit(nm, sd, p, x):
    """DA2/util/blocks.py:57-80 (ReLU is *not* in place so +x is the pre-activation x; bn=True adds an inference BatchNorm2d
    behind each conv, :70-76 -- present iff the state_dict carries bn1/bn2)."""
    def bn(t, q):
        if q + "running_var" not in sd:
            return t
        return F.batch_norm(t, sd[q + "running_mean"], sd[q + "running_var"], sd[q + "weight"], sd[q + "bias"], False, 0.0, 1e-5)
    out = bn(nm.conv(F.relu(x), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1), p + "bn1.")
    out = bn(nm.conv(F.relu(out), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1), p + "bn2.")
    return out + x


def fusion_block(nm, sd, p, xs, size=None):
    """DA2/util/blocks.py:123-148."""
    out = xs[0]
    if len(xs) == 2:
        out = out + residual_conv_unit(nm, sd, p + "resConfUnit1.", xs[1])
    out = residual_conv_unit(nm, sd, p + "resConfUnit2.", out)
    if size is None:
        out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        out = F.interpolate(out, size=tuple(size), mode="bilinear", align_corners=True)
    return nm.conv(out, sd[p + "out_conv.weight"], sd[p + "out_conv.bias"])


def dpt_head(nm, sd, pfx, taps, ph, pw, amodal, final_act, trace=None, cls_tokens=None):
    """DA2/dpt.py:161-197 (amodal) / RAW/dpt.py:117-150 (raw: no input_projection)."""
    layers = []
    for i, x in enumerate(taps):
        B, _, D = x.shape
        if cls_tokens is not None:   # use_clstoken read-out, DA2/dpt.py:164-167 (Linear(2D, D) + exact GELU, :110-117)
            readout = cls_tokens[i].unsqueeze(1).expand_as(x)
            x = F.gelu(nm.linear(torch.cat((x, readout), -1), sd[f"{pfx}readout_projects.{i}.0.weight"], sd[f"{pfx}readout_projects.{i}.0.bias"]))
        x = x.permute(0, 2, 1).reshape(B, D, ph, pw)
        x = nm.conv(x, sd[f"{pfx}projects.{i}.weight"], sd[f"{pfx}projects.{i}.bias"])
        if i == 0:
            x = nm.convT(x, sd[pfx + "resize_layers.0.weight"], sd[pfx + "resize_layers.0.bias"], 4)
        elif i == 1:
            x = nm.convT(x, sd[pfx + "resize_layers.1.weight"], sd[pfx + "resize_layers.1.bias"], 2)
        elif i == 3:
            x = nm.conv(x, sd[pfx + "resize_layers.3.weight"], sd[pfx + "resize_layers.3.bias"], stride=2, padding=1)
        layers.append(x)
    if amodal:  # DA2/dpt.py:178-179, 153-159
        for i in range(4):
            p = f"{pfx}input_projection.{i}."
            y = nm.conv(layers[i], sd[p + "0.weight"], sd[p + "0.bias"], padding=1)
            layers[i] = F.relu(cf_layernorm(y, sd[p + "1.weight"], sd[p + "1.bias"]))
    rn = [nm.conv(layers[i], sd[f"{pfx}scratch.layer{i + 1}_rn.weight"], None, padding=1) for i in range(4)]
    if trace is not None:
        for i in range(4):
            trace[f"layer{i + 1}_rn"] = rn[i]
    s = pfx + "scratch."
    path4 = fusion_block(nm, sd, s + "refinenet4.", [rn[3]], size=rn[2].shape[2:])
    path3 = fusion_block(nm, sd, s + "refinenet3.", [path4, rn[2]], size=rn[1].shape[2:])
    path2 = fusion_block(nm, sd, s + "refinenet2.", [path3, rn[1]], size=rn[0].shape[2:])
    path1 = fusion_block(nm, sd, s + "refinenet1.", [path2, rn[0]])
    if trace is not None:
        trace.update(path4=path4, path3=path3, path2=path2, path1=path1)
    out = nm.conv(path1, sd[s + "output_conv1.weight"], sd[s + "output_conv1.bias"], padding=1)
    out = F.interpolate(out, (ph * PATCH, pw * PATCH), mode="bilinear", align_corners=True)
    out = F.relu(nm.conv(out, sd[s + "output_conv2.0.weight"], sd[s + "output_conv2.0.bias"], padding=1))
    out = nm.conv(out, sd[s + "output_conv2.2.weight"], sd[s + "output_conv2.2.bias"])
    if trace is not None:
        trace["logits"] = out
    if final_act == "sigmoid":
        return torch.sigmoid(out)
    if final_act == "relu":
        return F.relu(out)
    return out


# ---------------------------------------------------------------------------------------
# top-level forwards
# ---------------------------------------------------------------------------------------
def build_guide(guide_type, guide_rgb, guide_mask, observation):
    """src/models/amodalsynthdrive/dav2.py:67-82."""
    if guide_type == "image+mask+observation":
        return torch.cat([guide_rgb, guide_mask, observation], dim=1)
    if guide_type == "image+mask":
        return torch.cat([guide_rgb, guide_mask], dim=1)
    if guide_type == "image+observation":
        return torch.cat([guide_rgb, observation], dim=1)
    if guide_type == "mask+observation":
        return torch.cat([guide_mask, observation], dim=1)
    if guide_type == "observation":
        return observation
    if guide_type == "mask":
        return guide_mask
    if guide_type == "none":
        return None
    raise NotImplementedError


@torch.no_grad()
def amodal_forward(sd: Dict[str, torch.Tensor], encoder: str, guide_type: str, loss_stategy: str,
                   x, guide_rgb=None, guide_mask=None, observation=None,
                   operand_dtype=torch.float32, trace: Optional[dict] = None):
    """AmodalDAv2.forward, dav2.py:64-85 -> DA2/dpt.py:225-231.  Returns [B,1,H,W]."""
    nm = _Numerics(operand_dtype)
    mean = torch.tensor(PIXEL_MEAN, dtype=x.dtype).view(-1, 1, 1)
    std = torch.tensor(PIXEL_STD, dtype=x.dtype).view(-1, 1, 1)
    x = (x - mean) / std
    guide = build_guide(guide_type, guide_rgb, guide_mask, observation)
    ph, pw = x.shape[-2] // PATCH, x.shape[-1] // PATCH
    taps = encoder_taps(nm, sd, "encoder.pretrained.", encoder, x, guide, trace)
    final = "none" if "ssi" in loss_stategy else "sigmoid"  # DA2/dpt.py:138-151
    return dpt_head(nm, sd, "encoder.depth_head.", taps, ph, pw, True, final, trace)


@torch.no_grad()
def raw_forward(sd: Dict[str, torch.Tensor], encoder: str, x, operand_dtype=torch.float32,
                trace: Optional[dict] = None):
    """RAW DepthAnythingV2.forward, RAW/dpt.py:176-184: x is already normalised; returns [B,H,W]."""
    nm = _Numerics(operand_dtype)
    ph, pw = x.shape[-2] // PATCH, x.shape[-1] // PATCH
    cls = None
    if "depth_head.readout_projects.0.0.weight" in sd:   # use_clstoken=True (RAW/dpt.py:83-90,120-123)
        taps, cls = encoder_taps(nm, sd, "pretrained.", encoder, x, None, trace, return_class_token=True)
    else:
        taps = encoder_taps(nm, sd, "pretrained.", encoder, x, None, trace)
    depth = dpt_head(nm, sd, "depth_head.", taps, ph, pw, False, "relu", trace, cls_tokens=cls)  # head ReLU RAW/dpt.py:113
    return F.relu(depth).squeeze(1)  # RAW/dpt.py:182-184


def rel_l1(a: torch.Tensor, b: torch.Tensor) -> float:
    """Parity metric of BASELINE.md §2: mean|a-b| / mean|b| with b the fp32 oracle."""
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().mean() / b.abs().mean())
