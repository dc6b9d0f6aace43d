"""Forward engine: runs the whole Depth-Anything-V2 / Amodal-DAv2 forward pass as a fixed sequence of
libada_hip launches on the current HIP stream.

The nn.Module tree in ``src/models`` owns the fp32 parameters (reference ``state_dict`` schema); this
engine owns everything derived from them:

* ``PackedWeights`` -- operand-typed, MFMA-friendly copies made once per parameter version: linear /
  1x1 weights as [N, K]; 3x3 weights as [N, 9*Cp] (tap-major, channels padded to 64); transposed-conv
  weights as [s*s*Cout, Cin] with the bias expanded; the RGB and guidance patch-embed filters fused
  into one [D, 1024] matrix (reference DA2/dinov2.py:237-240 adds the two embeddings, so one GEMM over
  the concatenated K does both); the q rows of qkv pre-multiplied by head_dim**-0.5 * log2(e) (base-2 softmax in the attention kernel).
* ``Workspace`` -- every activation buffer for a (batch, H, W), allocated once and kept resident in HBM.

Data layout: the residual stream is fp32 [B*N, D]; every tensor that is only ever a contraction
operand is stored in the operand type (fp16 by default); DPT-head activations are NHWC, 3x3-conv
inputs carry a one-pixel zero border so the implicit-GEMM gather never branches.
"""
from __future__ import annotations

import math
import os
import threading
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import (A_CONV3, ACT_NONE, ACT_RELU, ACT_SIGMOID, EP_BIAS, EP_GAMMA, EP_GELU, EP_RELU_OP, EP_RESIDUAL,
               EP_SWIGLU, EP_TAIL, MAP_PAD, MAP_SHUFFLE, MAP_TOKEN, HipExtError)
from . import attention as k_attention
from . import bilinear as _bilinear
from . import count_saturated
from . import dpt_tail as k_dpt_tail
from . import igemm as _igemm
from . import layernorm as _layernorm
from . import debug_epoch, instrumented, operand_dtype
from . import depth_stats as k_depth_stats
from . import token_diversity as k_token_diversity
from . import patchify as k_patchify
from . import pos_embed_resize as k_pos_embed_resize
from . import tapsum_resize as k_tapsum_resize
from . import write_cls as k_write_cls

_probe = None   # list of (label, device counter) while DepthEngine.saturation_report runs, else None


def _probed(kernel, launch):
    """Wraps a launcher: while a saturation probe is active, the operand-typed output of every launch is swept by ada_debug_count_saturated."""
    def run(*a, **kw):
        launch(*a, **kw)
        t = kw.get("out_op")
        if _probe is not None and t is not None and t.is_contiguous():
            c = torch.zeros(1, dtype=torch.int64, device=t.device)
            count_saturated(t, c)
            _probe.append((f"{kernel} #{len(_probe)} {list(t.shape)}", c))
    return run


k_igemm, k_layernorm, k_bilinear = _probed("igemm", _igemm), _probed("layernorm", _layernorm), _probed("bilinear", _bilinear)

PATCH = 14
LN_EPS = 1e-6
# Contractions of the DPT head that can run in split precision (PackedWeights.split), by name: "proj" = read-out + the four 1x1 `projects`,
# "rs0" / "rs1" / "rs3" = resize_layers (ConvT 4x4, ConvT 2x2, conv3x3 stride 2), "ip<i>" = input_projection conv of level i,
# "rn<i>" = layer<i+1>_rn, "rcu<i>" = the ResidualConvUnit convs of refinenet<i+1>, "out<i>" = its 1x1 out_conv, "oc1" / "oc2" = the tail
# convs.  Level 0 is the finest grid (4x the patch grid), level 3 the coarsest.  HEAD_ALIASES name whole families.
# "projw" = the four 1x1 projects with WEIGHT-ONLY split precision ([w_hi | w_lo] against the plain LayerNorm output walked twice, ada_igemm a_wrap:
# the weight's rounding error goes, the activation's stays; 2x the MACs instead of 3x and no [hi | lo] copy of the tap).  Ignored when "proj" is set.
HEAD_GROUPS = ("proj", "projw", "rs0", "rs1", "rs3") + tuple(f"{f}{i}" for f in ("ip", "rn", "rcu", "out") for i in range(4)) + ("oc1", "oc2")
HEAD_ALIASES = {"tok": ("proj", "rs0", "rs1", "rs3"), "ip": tuple(f"ip{i}" for i in range(4)), "rn": tuple(f"rn{i}" for i in range(4)),
                "rcu": tuple(f"rcu{i}" for i in range(4)), "out": tuple(f"out{i}" for i in range(4))}
# ada_dpt_tail_fwd (resize + output_conv2 fused; the up-sampled map "fin" -- 2.2 GB at ViT-L bs=32 -- is neither allocated nor written) is the
# default tail wherever it applies (fused_tail_applies): 1.18 ms against 2.13 ms for the resize kernel + tail GEMM at ViT-L bs=32
# (profiles/r03_p_fused_tail.txt).  ADA_FUSED_TAIL=0 selects the two-launch tail everywhere (A/B, and the only path for split-precision "oc2").
FUSED_TAIL = os.environ.get("ADA_FUSED_TAIL", "1") == "1"
# Sub-pixel merge (round 4): resize_layers[0/1] (ConvTranspose 4x4 s4 / 2x2 s2, reference DA2/dpt.py:88-100,173) and the 3x3 convolution of
# input_projection[0/1] that follows with nothing in between (:153-159,178-179) run as ONE 3x3 convolution over the patch grid whose
# s*s*C output columns are the output phases (functional.subpixel_merge; ada_igemm_args.tap_cols): 36 C^2 instead of 160 C^2 MACs per patch
# at level 0, 16 C^2 instead of 40 C^2 at level 1 (39 GFLOP of 1390 per ViT-L image), one fp16 rounding of the intermediate map less, and
# the up-sampled maps L[0] / L[1] are never written.  Exact in real arithmetic, weights composed in fp64 when they are packed.
# ADA_SUBPIXEL=0 keeps the two launches (A/B; also the path of every split-precision level).
SUBPIXEL = os.environ.get("ADA_SUBPIXEL", "1") == "1"
# output_conv1 in front of its resize (round 4): path_1 = resize_x2(out_conv(RCU2(...))) (util/blocks.py:140-146, out_conv already commuted in front of
# the resize) feeds scratch.output_conv1, a 3x3 convolution (DA2/dpt.py:192-193).  A 1x1 channel mix commutes with a per-channel resample, so
#   conv3x3(resize(z)) = b + sum over the 9 taps t of  shift_t(resize(W_t z)),   z = out_conv(u):   W_t z = (W_t W_out) u + W_t b_out
# -- ONE GEMM over the 148^2 grid with N = 9 * features / 2 (0.41 TFLOP at ViT-L bs = 32 instead of out_conv 0.09 + output_conv1 1.65) whose nine
# operand-typed tap maps ada_tapsum_resize_fwd gathers (9 taps x 4 bilinear corners per output element; a tap whose position falls into the
# zero padding drops out whole, out_conv's bias included).  The up-sampled operand map p1 is never written.  ADA_OC1_COMMUTE=0: the old path.
SUBPIXEL_SPLIT = os.environ.get("ADA_SUBPIXEL_SPLIT", "1") != "0"           # A/B: 0 = two launches wherever a level's resize / first conv are split groups (round 5)
OC1_COMMUTE_SPLIT = os.environ.get("ADA_OC1_COMMUTE_SPLIT", "1") != "0"     # A/B: 0 = round 5's resize -> 3x3 conv path wherever "oc1" is a split group
STAT_CHUNKS = 8
OC1_COMMUTE = os.environ.get("ADA_OC1_COMMUTE", "1") != "0"     # operand-typed tap maps; 0: the resize -> conv path (A/B, profiles/r04_g_*; also the path of a split-precision oc1)


# fp8 correction terms (ada_igemm_args.f8_from): a split-precision product whose two small terms x_lo w_hi + x_hi w_lo run on the fp8 matrix pipe
# (v_mfma_scale_f32_16x16x128_f8f6f4, twice the fp16 rate) -- 2x the MACs' time instead of 3x.  oracle/study_fp8_correction.py: three mantissa bits
# are enough for terms that are 2^-11 of the product.  ADA_F8_CORR=0 keeps the three fp16 terms everywhere (A/B).
_F8_ENV = os.environ.get("ADA_F8_CORR", "1")     # 1 | 0 | enc | head (A/B: only the encoder's split blocks / only the head's split groups)
F8_CORR = _F8_ENV != "0"
F8_ENC, F8_HEAD = _F8_ENV in ("1", "enc"), _F8_ENV in ("1", "head")
F8_A_SCALES = 117 | (127 << 16)     # E8M0 bytes of the activation's two byte segments: lo8 = e5m2((x - x_hi) 2^10), hi8 = e5m2(x)


def f8_weight_split(wm: torch.Tensor, op, taps: int = 1):
    """[N, taps * K] fp32 (K a multiple of 128) -> ([N, taps * 2 K] operand-typed storage, f8_scales word): per tap [w_hi | w_hi8 | w_lo8] with
    w_hi = round(w) in the operand type (K slots), then K bytes e4m3(w_hi 2^s_hi) and K bytes e4m3((w - w_hi) 2^s_lo), one power-of-two scale per
    tensor and segment chosen so that the largest magnitude lands in [224, 448].  The contraction against an activation stored [hi | lo8 | hi8]
    (split_seg = -K) evaluates x_hi w_hi + 2^-10 x_lo8 w_hi8 2^-s_hi + x_hi8 w_lo8 2^-s_lo (include/ada_hip.h, ada_igemm_args.f8_from)."""
    n = wm.shape[0]
    k = wm.shape[1] // taps
    assert wm.shape[1] == taps * k and k % 128 == 0, (wm.shape, taps)
    w = wm.reshape(n, taps, k).float()
    hi = w.to(op)
    lo = w - hi.float()

    def enc(t):
        m = float(t.abs().max())
        sh = max(-100, min(100, int(math.floor(math.log2(448.0 / m))))) if m > 0 else 0
        q = (t.float() * (2.0 ** sh)).clamp(-448.0, 448.0)
        try:
            b = q.to(torch.float8_e4m3fn).view(torch.uint8)
        except (RuntimeError, TypeError):      # no device cast for the dtype on this backend
            b = q.cpu().to(torch.float8_e4m3fn).view(torch.uint8).to(t.device)
        return b, 127 - sh
    hi8, sb_hi = enc(hi)
    lo8, sb_lo = enc(lo)
    packed = torch.cat([hi.contiguous().view(torch.uint8).reshape(n, taps, 2 * k), hi8, lo8], dim=2).reshape(n, taps * 4 * k)
    return packed.contiguous().view(op), F8_A_SCALES | (sb_hi << 8) | (sb_lo << 24)


def fused_tail_applies(half, halfp, hi, ho, split):
    """Whether ada_dpt_tail_fwd can take the tail (include/ada_hip.h): single-precision oc2, 64 or 128 (un-padded) channels, and a vertical
    scale whose 10-row halo tiles span at most 7 source row intervals -- always true for the model's 14 / 8 ratio."""
    if "oc2" in split or half != halfp or halfp // 64 > 2:
        return False
    sy = (hi - 1) / (ho - 1) if ho > 1 else 0.0
    return int(sy * 9) + 2 <= 7


VIT = {
    "vits": dict(dim=384, depth=12, heads=6, ffn="mlp"),
    "vitb": dict(dim=768, depth=12, heads=12, ffn="mlp"),
    "vitl": dict(dim=1024, depth=24, heads=16, ffn="mlp"),
    "vitg": dict(dim=1536, depth=40, heads=24, ffn="swiglu"),
}
TAPS = {"vits": [2, 5, 8, 11], "vitb": [2, 5, 8, 11], "vitl": [4, 11, 17, 23], "vitg": [9, 19, 29, 39]}
Q_PRESCALE = 0.125 * 1.4426950408889634  # head_dim**-0.5 * log2(e), head_dim = 64
MAX_ROWS = (1 << 24) - 1  # row-index limit of the kernels' fast division


def _r64(c: int) -> int:
    return (c + 63) // 64 * 64




class PackedWeights:
    """Operand-typed copies of the parameters, laid out for the kernels.  ``sd``: name -> fp32 CUDA tensor
    with the *raw* model's key names (``pretrained.*``, ``depth_head.*``)."""

    def __init__(self, sd: Dict[str, torch.Tensor], encoder: str, guided: bool, amodal_head: bool, split_head=False,
                 enc_split_blocks: int = 0, head_only: bool = False, tap_split: bool = False,
                 f8: str = "both", tap_f8: Optional[bool] = None, f8_only=None):
        op = operand_dtype()
        # head_only: the DPT head's weights only (the second rung of the precision ladder, DepthEngine._escalate, re-runs the head from the taps)
        # tap_split: the four taps are stored [hi | lo] whatever the head's own policy -- so that a split-precision head can be re-run from them
        self.head_only, self.tap_split = bool(head_only), bool(tap_split)
        cfg = VIT[encoder]
        D = cfg["dim"]
        self.encoder, self.guided, self.amodal_head = encoder, guided, amodal_head
        # split_head: contractions of the DPT head run in split precision -- activations are stored as [hi | lo] column segments by the
        # producing kernel (split_seg), weights are packed [w_hi | w_hi | w_lo], and one fp16 GEMM over the three k segments (hi, lo, hi:
        # the third re-reads the first, ada_igemm a_dup_seg) evaluates x_hi w_hi + x_lo w_hi + x_hi w_lo (~fp32 operand accuracy).  Used where the head's operand rounding
        # is what limits parity: the unbounded-output models (raw ReLU / 'ssi' heads) and ViT-S.
        # The choice is PER LAYER GROUP (HEAD_GROUPS): a contraction of a split group reads a [hi | lo] activation (its producer is
        # told through split_seg) and [w_hi | w_hi | w_lo] weights; the others run at 1x the MACs.  True = every group.
        if split_head is True:
            split_head = HEAD_GROUPS
        self.split = frozenset(g for name in (split_head or ()) for g in HEAD_ALIASES.get(name, (name,)))
        unknown = self.split - set(HEAD_GROUPS)
        if unknown:
            raise HipExtError(f"unknown head layer group(s) {sorted(unknown)}; known: {HEAD_GROUPS}")
        self.split_head = bool(self.split)
        # enc_split_blocks = K: the linear layers of the FIRST K transformer blocks run in split precision.  Operand rounding noise injected
        # early is amplified by every later block (oracle study, profiles/r04_e_raw_vitg_operand_noise_by_block.txt: blocks 0-9 of ViT-G carry
        # half of the encoder's share, blocks 30-39 a fiftieth), so for the unbounded-output ViT-G model, whose encoder alone reaches
        # 0.7e-3 ... 1.05e-3 depending on the weight draw, the head's split precision is not enough.  qkv and fc1 / w12 read the LayerNorm
        # output as [hi | lo] (full three-term product).  Round 6: so do proj and fc2 / w3 -- the attention kernel and the GELU / SwiGLU epilogues write
        # the attention output and the MLP hidden of a split block in the split form too (rounds 4-5 walked the plain activation twice against
        # [w_hi | w_lo]: the weight's rounding error went, the activation's stayed -- and that was what bounded the everything-split engine: raw ViT-G
        # 6.2e-4 of output error from these two activations alone against 2.9e-4 from the attention core, oracle/study_rung3_floor.py).
        self.enc_split_blocks = max(0, min(int(enc_split_blocks), cfg["depth"]))
        # ... with the two correction terms of qkv / fc1 / w12 on the fp8 matrix pipe where the build has it (fp16 operands, D a multiple of 128)
        # f8: which of the two users take it -- "both" | "enc" | "head" | "none" (the caller's precision policy, DA2/dpt.py::_f8_policy)
        if f8 not in ("both", "enc", "head", "none"):
            raise HipExtError(f"PackedWeights: f8={f8!r} (both | enc | head | none)")
        self.f8_head = F8_HEAD and f8 in ("both", "head") and op == torch.float16
        self.f8_only = None if f8_only is None else frozenset(f8_only)      # head groups that may take it (None: every split group)
        self.enc_f8 = F8_ENC and f8 in ("both", "enc") and self.enc_split_blocks > 0 and op == torch.float16 and D % 128 == 0
        # the form of the taps when they are kept split: that of their reader -- this object's own "proj" group, or (tap_f8 given) the weights of
        # the ladder's second rung, which are packed later from the same state_dict
        self.tap_f8 = (self.f8_head if tap_f8 is None else (bool(tap_f8) and F8_HEAD and op == torch.float16)) and D % 128 == 0
        self.dim, self.depth, self.heads, self.ffn = D, cfg["depth"], cfg["heads"], cfg["ffn"]

        def f32(name):
            return sd[name].detach().to(torch.float32).contiguous()

        def lin(w):  # [N, K] -> operand type, K padded to 64
            w = w.reshape(w.shape[0], -1)
            k = w.shape[1]
            if k % 64:
                w = F.pad(w, (0, _r64(k) - k))
            return w.to(op).contiguous()

        def conv3(w):  # [Co, Ci, 3, 3] -> [Co, 9 * Cip], tap-major
            co, ci = w.shape[:2]
            w = w.permute(0, 2, 3, 1)
            if ci % 64:
                w = F.pad(w, (0, _r64(ci) - ci))
            return w.reshape(co, -1).to(op).contiguous()

        def convT(w, b, s):  # [Ci, Co, s, s] -> [s*s*Co, Cip], bias expanded to [s*s*Co]
            ci, co = w.shape[:2]
            wt = w.permute(2, 3, 1, 0).reshape(s * s * co, ci)
            return lin(wt), b.repeat(s * s).contiguous()

        p = "pretrained."
        self.blocks = []
        self.guide_channels = 0
        if not head_only:
            self._pack_encoder(sd, f32, lin, op, D, guided)
        self._pack_head(sd, f32, lin, conv3, op, D, amodal_head)

    def _pack_encoder(self, sd, f32, lin, op, D, guided):
        p = "pretrained."
        w_rgb = f32(p + "patch_embed.proj.weight").reshape(D, -1)
        b_pe = f32(p + "patch_embed.proj.bias")
        if guided:
            w_g = f32(p + "patch_embed_guidance.proj.weight")
            self.guide_channels = w_g.shape[1]
            w_rgb = torch.cat([w_rgb, w_g.reshape(D, -1)], dim=1)
            b_pe = b_pe + f32(p + "patch_embed_guidance.proj.bias")
        # split-precision patch embedding: the image is the one operand whose fp16 rounding error (2.5e-4 relative) is
        # injected into every token, so it is carried as hi + lo halves and the weights likewise; one GEMM over
        # K = 3 * 1024 computes x_hi w_hi + x_lo w_hi + x_hi w_lo (0.6 % of the model's MACs instead of 0.2 %).
        k_real = w_rgb.shape[1]
        seg = _r64(k_real)
        w_pad = F.pad(w_rgb, (0, seg - k_real))
        w_hi = w_pad.to(op)
        w_lo = (w_pad - w_hi.float()).to(op)
        self.pe_w = torch.cat([w_hi, w_hi, w_lo], dim=1).contiguous()
        self.pe_b = b_pe.contiguous()
        self.pe_k = self.pe_w.shape[1]      # K of the embedding GEMM: three segments (hi, lo, hi) of pe_seg
        self.pe_seg = seg
        self.pe_k_alg = k_real
        self.cls = f32(p + "cls_token").reshape(D)
        self.pos_native = f32(p + "pos_embed")  # [1, 1 + 37*37, D]
        self._pos_cache: Dict[tuple, torch.Tensor] = {}

        self.blocks = []
        for i in range(self.depth):
            b = f"{p}blocks.{i}."
            qw, qb = f32(b + "attn.qkv.weight").clone(), f32(b + "attn.qkv.bias").clone()
            # q * head_dim**-0.5 (reference attention.py:41,53) folded into the weights, together with log2(e): the attention
            # kernel runs its softmax in base 2 (exp2 is the native transcendental), softmax_e(s) == softmax_2(s * log2 e)
            qw[:D] *= Q_PRESCALE
            qb[:D] *= Q_PRESCALE
            esplit = i < self.enc_split_blocks

            def lin3(wm):   # [N, K] -> [w_hi | w_hi | w_lo] (against a [hi | lo] activation, a_dup_seg) for the split blocks
                if not esplit:
                    return lin(wm), 0
                if self.enc_f8:  # ... or [w_hi | w_hi8 | w_lo8] against [hi | lo8 | hi8] (f8_from): the two correction terms on the fp8 pipe
                    return f8_weight_split(wm, op)
                hi_ = wm.to(op)
                return torch.cat([hi_, hi_, (wm - hi_.float()).to(op)], dim=1).contiguous(), 0

            qkv_w, qkv_f8 = lin3(qw)
            proj_w, proj_f8 = lin3(f32(b + "attn.proj.weight"))
            blk = dict(
                ln1_w=f32(b + "norm1.weight"), ln1_b=f32(b + "norm1.bias"), esplit=esplit,
                qkv_w=qkv_w, qkv_f8=qkv_f8, qkv_b=qb,
                proj_w=proj_w, proj_f8=proj_f8, proj_b=f32(b + "attn.proj.bias"), ls1=f32(b + "ls1.gamma"),
                ln2_w=f32(b + "norm2.weight"), ln2_b=f32(b + "norm2.bias"), ls2=f32(b + "ls2.gamma"),
            )
            if self.ffn == "mlp":
                fc1_w, fc1_f8 = lin3(f32(b + "mlp.fc1.weight"))
                fc2_w, fc2_f8 = lin3(f32(b + "mlp.fc2.weight"))
                blk.update(fc1_w=fc1_w, fc1_f8=fc1_f8, fc1_b=f32(b + "mlp.fc1.bias"), fc2_w=fc2_w, fc2_f8=fc2_f8, fc2_b=f32(b + "mlp.fc2.bias"))
                blk["hidden"] = blk["fc1_w"].shape[0]
            else:
                w12, b12 = f32(b + "mlp.w12.weight"), f32(b + "mlp.w12.bias")
                hid = w12.shape[0] // 2
                assert hid % 32 == 0
                # interleave x1 / x2 rows in groups of 32 so one wave's two MFMA column tiles hold the gate pair
                idx = torch.arange(hid, device=w12.device).reshape(-1, 32)
                order = torch.stack([idx, idx + hid], dim=1).reshape(-1)
                w12_w, w12_f8 = lin3(w12[order])
                w3_w, w3_f8 = lin3(f32(b + "mlp.w3.weight"))
                blk.update(w12_w=w12_w, w12_f8=w12_f8, w12_b=b12[order].contiguous(), w3_w=w3_w, w3_f8=w3_f8, w3_b=f32(b + "mlp.w3.bias"))
                blk["hidden"] = hid
            self.blocks.append(blk)
        self.norm_w, self.norm_b = f32(p + "norm.weight"), f32(p + "norm.bias")

    def _pack_head(self, sd, f32, lin, conv3, op, D, amodal_head):
        lin1, conv3_1 = lin, conv3

        def triple(w2d):   # [..., K] fp32, K already padded to a multiple of 64
            hi = w2d.to(op)
            lo = (w2d - hi.float()).to(op)
            return torch.cat([hi, hi, lo], dim=-1)

        # Groups whose contractions take their two correction terms on the fp8 pipe (module comment at F8_CORR): the weights are [w_hi | w_hi8 | w_lo8]
        # per tap, the activation [hi | lo8 | hi8] -- the group's PRODUCERS are told through a negative split_seg (_head: S()).  Needs the operand's
        # channel count (padded) to be a multiple of 128; a group of another width keeps the three fp16 terms.
        self.f8_groups = set()
        f8_ok = self.f8_head

        def f8_pack(w2d, group, taps):
            t, word = f8_weight_split(w2d, op, taps=taps)
            t.f8_scales = word          # read by DepthEngine._kdup: the packed shape alone does not tell this form from a plain operand
            self.f8_groups.add(group)
            return t

        def lin(w, group):  # noqa: F811
            if group not in self.split:
                return lin1(w)
            w = w.reshape(w.shape[0], -1)
            k = w.shape[1]
            if k % 64:
                w = F.pad(w, (0, _r64(k) - k))
            if f8_ok and (self.f8_only is None or group in self.f8_only) and w.shape[1] % 128 == 0:
                return f8_pack(w, group, 1)
            assert group not in self.f8_groups, group
            return triple(w).contiguous()

        def conv3(w, group):  # noqa: F811  [Co, Ci, 3, 3] -> [Co, 9 * 3 * Cip]: per tap [hi | hi | lo]
            if group not in self.split:
                return conv3_1(w)
            co, ci = w.shape[:2]
            w = w.permute(0, 2, 3, 1)
            if ci % 64:
                w = F.pad(w, (0, _r64(ci) - ci))
            if f8_ok and (self.f8_only is None or group in self.f8_only) and w.shape[-1] % 128 == 0:
                return f8_pack(w.reshape(co, -1), group, 9)
            assert group not in self.f8_groups, group
            return triple(w).reshape(co, -1).contiguous()

        def convT(w, b, s_, group):  # noqa: F811  [Ci, Co, s, s] -> [s*s*Co, Cip], bias expanded to [s*s*Co]
            ci, co = w.shape[:2]
            wt = w.permute(2, 3, 1, 0).reshape(s_ * s_ * co, ci)
            return lin(wt, group), b.repeat(s_ * s_).contiguous()

        h = "depth_head."
        # use_clstoken read-out (reference DA2/dpt.py:110-117,164-167): Linear(2D -> D) on [patch token | class token] + GELU.  The class
        # token half is the same for every patch of an image, so it becomes a per-image bias  c_b = W_cls cls_b + b  (one tiny GEMM) and
        # the patch half runs as a D -> D GEMM per image with that bias.
        self.readout = f"{h}readout_projects.0.0.weight" in sd
        if self.readout:
            self.ro_wx, self.ro_wc, self.ro_b = [], [], []
            for i in range(4):
                wr = f32(f"{h}readout_projects.{i}.0.weight")
                self.ro_wx.append(lin(wr[:, :D], "proj"))
                self.ro_wc.append(lin(wr[:, D:], "proj"))
                self.ro_b.append(f32(f"{h}readout_projects.{i}.0.bias"))
        self.oc = [sd[f"{h}projects.{i}.weight"].shape[0] for i in range(4)]
        self.features = sd[h + "scratch.layer1_rn.weight"].shape[0]
        def proj_lin(w):   # "projw": [w_hi | w_lo] against the plain tap (HEAD_GROUPS comment); "proj" (full split) takes precedence
            w2 = w.reshape(w.shape[0], -1)
            if "projw" in self.split and "proj" not in self.split and w2.shape[1] % 64 == 0:
                hi = w2.to(op)
                return torch.cat([hi, (w2 - hi.float()).to(op)], dim=1).contiguous()
            return lin(w, "proj")
        self.proj_w = [proj_lin(f32(f"{h}projects.{i}.weight")) for i in range(4)]
        self.proj_b = [f32(f"{h}projects.{i}.bias") for i in range(4)]
        self.rs0_w, self.rs0_b = convT(f32(h + "resize_layers.0.weight"), f32(h + "resize_layers.0.bias"), 4, "rs0")
        self.rs1_w, self.rs1_b = convT(f32(h + "resize_layers.1.weight"), f32(h + "resize_layers.1.bias"), 2, "rs1")
        self.rs3_w, self.rs3_b = conv3(f32(h + "resize_layers.3.weight"), "rs3"), f32(h + "resize_layers.3.bias")
        # level -> merged sub-pixel convolution (single-precision levels only): resize_layers[i] + input_projection[i][0] in the amodal head,
        # resize_layers[i] + scratch.layer{i+1}_rn (no bias, util/blocks.py:20-24) in the raw head
        self.sp = {}
        if SUBPIXEL:
            from .functional import subpixel_merge
            nxt = "ip" if amodal_head else "rn"
            for i, s_ in ((0, 4), (1, 2)):
                # (round 6) a level whose transposed conv AND the 3x3 conv behind it are split groups with fp8 correction terms is merged too: the merged weights are
                # packed [w_hi | w_hi8 | w_lo8] per tap and `projects[i]` writes the patch-grid tensor [hi | lo8 | hi8] (needs a channel count that is a multiple of 128)
                both = f"rs{i}" in self.split and f"{nxt}{i}" in self.split
                if (f"rs{i}" in self.split or f"{nxt}{i}" in self.split) and not (
                        SUBPIXEL_SPLIT and both and f8_ok and _r64(sd[f"{h}projects.{i}.weight"].shape[0]) % 128 == 0
                        and (self.f8_only is None or {f"rs{i}", f"{nxt}{i}"} <= self.f8_only)):
                    continue
                if amodal_head:
                    w3_, b3_ = f32(f"{h}input_projection.{i}.0.weight"), f32(f"{h}input_projection.{i}.0.bias")
                else:
                    w3_, b3_ = f32(f"{h}scratch.layer{i + 1}_rn.weight"), None
                wm, bias, tapb, masks = subpixel_merge(f32(f"{h}resize_layers.{i}.weight"), f32(f"{h}resize_layers.{i}.bias"), w3_, b3_, s_)
                ci = wm.shape[2]
                if ci % 64:
                    wm = F.pad(wm, (0, _r64(ci) - ci))
                wq = f8_pack(wm.reshape(wm.shape[0], -1), f"{nxt}{i}", 9) if both else wm.reshape(wm.shape[0], -1).to(op).contiguous()
                self.sp[i] = dict(s=s_, w=wq, b=bias, tapb=tapb, masks=masks, split=both,
                                  taps_per_col=sum(bin(m).count("1") for m in masks) / float(s_ * s_))
        if amodal_head:
            self.ip_w = [conv3(f32(f"{h}input_projection.{i}.0.weight"), f"ip{i}") for i in range(4)]
            self.ip_b = [f32(f"{h}input_projection.{i}.0.bias") for i in range(4)]
            self.ip_ln_w = [f32(f"{h}input_projection.{i}.1.weight") for i in range(4)]
            self.ip_ln_b = [f32(f"{h}input_projection.{i}.1.bias") for i in range(4)]
        s = h + "scratch."
        self.rn_w = [conv3(f32(f"{s}layer{i + 1}_rn.weight"), f"rn{i}") for i in range(4)]
        self.fuse = []
        for k in range(1, 5):
            r = f"{s}refinenet{k}."
            d = dict(out_w=lin(f32(r + "out_conv.weight"), f"out{k - 1}"), out_b=f32(r + "out_conv.bias"))
            for u in (1, 2):
                for c in (1, 2):
                    w, b = f32(f"{r}resConfUnit{u}.conv{c}.weight"), f32(f"{r}resConfUnit{u}.conv{c}.bias")
                    bn = f"{r}resConfUnit{u}.bn{c}."
                    if bn + "running_var" in sd:      # use_bn=True: inference BatchNorm folded into the conv (reference blocks.py:70-76)
                        from .functional import fold_batchnorm
                        w, b = fold_batchnorm(w, b, f32(bn + "weight"), f32(bn + "bias"), f32(bn + "running_mean"), f32(bn + "running_var"))
                    d[f"u{u}c{c}_w"] = conv3(w, f"rcu{k - 1}")
                    d[f"u{u}c{c}_b"] = b.contiguous()
            self.fuse.append(d)  # index k-1
        self.oc1_w, self.oc1_b = conv3(f32(s + "output_conv1.weight"), "oc1"), f32(s + "output_conv1.bias")
        self.oc1c = None     # output_conv1 composed with refinenet1.out_conv, one 1x1 per tap: [9 * half, Fp] (+ the bias each tap map carries)
        half_ = self.features // 2
        # (round 6) ... ALSO when "oc1" / "out0" are split groups with fp8 correction terms: the tap-map GEMM then reads u[0] as [hi | lo8 | hi8] against the composed
        # matrix packed [w_hi | w_hi8 | w_lo8]; the nine tap maps stay operand-typed.  The ladder's second rung (-2 ms of its 27) and the raw ViT-B / ViT-L heads.
        split_commute = (OC1_COMMUTE_SPLIT and "oc1" in self.split and "out0" in self.split and f8_ok and self.features % 128 == 0
                         and (self.f8_only is None or {"oc1", "out0"} <= self.f8_only))
        if OC1_COMMUTE and (split_commute or ("oc1" not in self.split and "out0" not in self.split)) and half_ in (32, 64, 128):
            from .functional import compose_f32
            w1 = f32(s + "output_conv1.weight")                                  # [half, F, 3, 3]
            wo_ = f32(s + "refinenet1.out_conv.weight").reshape(self.features, self.features)   # [F(cm), F(ci)]
            bo_ = f32(s + "refinenet1.out_conv.bias")
            wt_ = w1.permute(2, 3, 0, 1).reshape(9 * half_, self.features)       # rows (tap, co), columns cm
            # [w_hi | w_lo] (ada_igemm a_wrap): rounding the COMPOSED matrix to the operand type once is not harmless -- its error acts on the
            # activation's large common-mode part coherently over all taps and positions and survives the resizes and convolutions behind it
            # (oracle study: 5.3e-4 at the output of ViT-B from this rounding alone, against 1.0e-4 / 0.9e-4 for the two factors rounded
            # separately; profiles/r04_g_output_conv1_commute.txt).  K = 2 * 256 on a GEMM that is bound by its 1.6 GB of output anyway.
            wc_ = compose_f32(wt_, wo_.t())       # the library's own GEMM in split precision (functional.compose_f32); W_t b_out is a reduction
            wc_hi = wc_.to(op)
            tap_b = (wt_.double() * bo_.double()[None, :]).sum(1).float().contiguous()
            if split_commute:
                wq, word = f8_weight_split(wc_, op)
                self.oc1c = dict(w=wq, f8=word, b=tap_b)
            else:
                self.oc1c = dict(w=torch.cat([wc_hi, (wc_ - wc_hi.float()).to(op)], dim=1).contiguous(), b=tap_b)
        self.oc2_w, self.oc2_b = conv3(f32(s + "output_conv2.0.weight"), "oc2"), f32(s + "output_conv2.0.bias")
        self.tail_w = f32(s + "output_conv2.2.weight").reshape(-1).contiguous()
        self.tail_b = float(f32(s + "output_conv2.2.bias").reshape(-1)[0].item())

    def pos_embed(self, ph: int, pw: int) -> torch.Tensor:
        """[1 + ph*pw, D] fp32 position table for a ph x pw patch grid (reference DA2/dinov2.py:199-230).
        The native 37x37 square grid is used as is; anything else is a one-off bicubic resample of the
        parameter by ada_pos_embed_resize (cached per grid) -- a parameter transform like weight packing, not part of the per-image path."""
        key = (ph, pw)
        if key in self._pos_cache:
            return self._pos_cache[key]
        pos = self.pos_native
        n = pos.shape[1] - 1
        if not (ph * pw == n and ph == pw):
            sq = int(math.sqrt(n))
            # reference quirk kept on purpose: w0 is derived from x.shape[2] (image *height*) -- dinov2.py:233,209
            w0, h0 = ph + 0.1, pw + 0.1
            src = pos[0].contiguous()
            res = torch.empty(1 + ph * pw, src.shape[1], dtype=torch.float32, device=src.device)
            k_pos_embed_resize(src, sq, src.shape[1], ph, pw, float(w0) / sq, float(h0) / sq, res)
            pos = res[None]
        out = pos[0].contiguous()
        self._pos_cache[key] = out
        return out


class Workspace:
    def __init__(self, pw_: PackedWeights, B: int, H: int, W: int, device, head_only: bool = False):
        """head_only: the buffers of the DPT head and the four taps only (the precision ladder's second rung re-runs the head from the taps)."""
        op = operand_dtype()
        self.B, self.H, self.W = B, H, W
        ph, pw = H // PATCH, W // PATCH
        self.ph, self.pw = ph, pw
        Np = ph * pw
        N = Np + 1
        D = pw_.dim
        T, P = B * N, B * Np
        Fch = pw_.features
        Fp = _r64(Fch)
        def mm(group):     # an op-typed head tensor holds [hi | lo] segments iff the contraction that READS it is in a split group
            return 2 if group in pw_.split else 1
        # the taps are [hi | lo] when the projects read them in split precision -- or when the engine keeps them so for the ladder's second rung
        m = 2 if (pw_.tap_split or "proj" in pw_.split) else 1
        self.tap_seg = (-D if pw_.tap_f8 else D) if m == 2 else 0     # < 0: the [hi | lo8 | hi8] form (ada_igemm_args.f8_from)
        first = "ip" if pw_.amodal_head else "rn"     # the contraction family that reads the reassembled maps L[i]

        def z(*shape, dtype=op):
            return torch.zeros(*shape, dtype=dtype, device=device)

        if not head_only:
            self.a_pe = z(P, 2 * pw_.pe_seg)     # split-precision patches: [hi | lo]
            self.x = z(T, D, dtype=torch.float32)
            # LayerNorm output ([hi | lo] column segments for the split-precision blocks: PackedWeights.enc_split_blocks)
            self.y = z(T, 2 * D if pw_.enc_split_blocks > 0 else D)
            self.qkv = z(T, 3 * D)
            wide = 2 if pw_.enc_split_blocks > 0 else 1       # attention output / MLP hidden: [hi | lo] / [hi | lo8 | hi8] rows in the split blocks
            self.o = z(T, wide * D)
            hidden = pw_.blocks[0]["hidden"]
            self.hd = z(T, wide * hidden)
        self.taps = [z(P, m * D) for _ in range(4)]
        # the precision ladder's per-image statistics, one buffer (one host read): stat_sums = (sum s, sum s (1 - s)) of the depth map in chunks
        # (ada_depth_stats_fwd), stat_div = (sum of column variances, sum of column mean squares) of the last tap in 64-column chunks (ada_token_diversity_fwd)
        # ... and stat_in = the same pair over the PATCHIFIED INPUT (a_pe: one row per patch, pixels + guide channels): exactly zero variance when every patch of the
        # image is the same (constant images, pixel checkerboards) -- the input-side trigger of the flat-input rung
        G = (D + 63) // 64
        Gi = 0 if head_only else pw_.pe_seg // 64
        self.stat_buf = z(B * (STAT_CHUNKS + G + Gi) * 2, dtype=torch.float32)
        self.stat_sums = self.stat_buf[:B * STAT_CHUNKS * 2].view(B, STAT_CHUNKS, 2)
        self.stat_div = self.stat_buf[B * STAT_CHUNKS * 2:B * (STAT_CHUNKS + G) * 2].view(B, G, 2)
        self.stat_in = self.stat_buf[B * (STAT_CHUNKS + G) * 2:].view(B, Gi, 2) if Gi else None
        if pw_.readout:
            self.cls_op = [z(B, m * D) for _ in range(4)]                          # final-LayerNorm'd class tokens, operand-typed
            self.cls_bias = [z(B, D, dtype=torch.float32) for _ in range(4)]       # W_cls cls + b per image
            self.taps_ro = [z(P, m * D) for _ in range(4)]                         # read-out tokens (inputs of `projects`)
        # head grids
        self.grid = [(4 * ph, 4 * pw), (2 * ph, 2 * pw), (ph, pw), ((ph - 1) // 2 + 1, (pw - 1) // 2 + 1)]
        oc = pw_.oc
        ocp = [_r64(c) for c in oc]
        self.ocp, self.Fp = ocp, Fp
        # levels whose resize + first 3x3 conv run as one sub-pixel convolution: `projects[i]` writes a zero-bordered patch-grid tensor tp[i]
        # (the conv's A operand); t<i> and the up-sampled map L[i] do not exist
        self.tp = {i: z(B, ph + 2, pw + 2, (2 if pw_.sp[i].get("split") else 1) * ocp[i]) for i in pw_.sp}
        self.t0 = None if 0 in pw_.sp else z(P, mm("rs0") * ocp[0])
        self.t1 = None if 1 in pw_.sp else z(P, mm("rs1") * ocp[1])
        self.pre3 = z(B, ph + 2, pw + 2, mm("rs3") * ocp[3])
        self.L = [None if i in pw_.sp else z(B, g[0] + 2, g[1] + 2, mm(f"{first}{i}") * ocp[i]) for i, g in enumerate(self.grid)]
        if pw_.amodal_head:
            self.ipf = [z(B * g[0] * g[1], oc[i], dtype=torch.float32) for i, g in enumerate(self.grid)]
            self.L2 = [z(B, g[0] + 2, g[1] + 2, mm(f"rn{i}") * ocp[i]) for i, g in enumerate(self.grid)]
        self.rnx = [z(B * g[0] * g[1], Fch, dtype=torch.float32) for g in self.grid]
        # raw head: [patch, s*s*features] fp32 output of a merged resize + layer_rn convolution, re-laid out into rnx / rnr by one pass
        self.spf = {} if pw_.amodal_head else {i: z(B * self.grid[i][0] * self.grid[i][1], Fch, dtype=torch.float32) for i in pw_.sp}
        self.rnr = [z(B, g[0] + 2, g[1] + 2, mm(f"rcu{i}") * Fp) for i, g in enumerate(self.grid)]
        self.tmpa = [z(B, g[0] + 2, g[1] + 2, mm(f"rcu{i}") * Fp) for i, g in enumerate(self.grid)]
        # fp32 side buffers whose lifetimes do not overlap share storage (plain row buffers only: the zero-bordered operand tensors keep their own,
        # their borders are written once).  Launch order of _head: ip conv -> LN (ipf dead) ... rn convs ... per level i = 3..0:
        # RCU2(i) [reads s[i]] -> out_conv [writes zf[i]] -> RCU1(i-1) [reads rnx[i-1], writes r[i-1]] -> resize [reads zf[i], r[i-1], writes s[i-1]];
        # then resize zf[0] -> p1, output_conv1 [writes oc1].  So: zf[i] lives in rnx[i] (dead since RCU1(i) / RCU2(3)), r[i] in ipf[i] where that is
        # large enough, and oc1 (4 x the pixels, half the channels = 2 x one level-0 buffer) in the level-0 pair (r[0], s[0]).
        rows = [B * g[0] * g[1] for g in self.grid]
        lvl0 = z(2 * rows[0] * Fch, dtype=torch.float32)
        self.r = [lvl0[:rows[0] * Fch].view(rows[0], Fch)]
        self.s = [lvl0[rows[0] * Fch:].view(rows[0], Fch)]
        for i in range(1, 4):
            if pw_.amodal_head and oc[i] >= Fch:
                self.r.append(self.ipf[i].view(-1)[:rows[i] * Fch].view(rows[i], Fch))
            else:
                self.r.append(z(rows[i], Fch, dtype=torch.float32))
            self.s.append(z(rows[i], Fch, dtype=torch.float32))
        self.sr = [z(B, g[0] + 2, g[1] + 2, mm(f"rcu{i}") * Fp) for i, g in enumerate(self.grid)]
        self.u = [z(B * g[0] * g[1], mm(f"out{i}") * Fp) for i, g in enumerate(self.grid)]
        self.zf = self.rnx
        g0 = self.grid[0]
        self.g296 = (2 * g0[0], 2 * g0[1])
        self.p1 = None if pw_.oc1c is not None else z(B, self.g296[0] + 2, self.g296[1] + 2, mm("oc1") * Fp)
        # the nine tap maps of output_conv1 on the level-0 grid
        self.tmaps = z(rows[0], 9 * (Fch // 2)) if pw_.oc1c is not None else None
        half = Fch // 2
        self.half, self.halfp = half, _r64(half)
        self.oc1 = lvl0.view(B * self.g296[0] * self.g296[1], half)
        # fused tail (ada_dpt_tail_fwd): resize + output_conv2 in one kernel, the up-sampled map is never materialised.  Needs the
        # single-precision head and a channel count that is already a multiple of 64 (ViT-B / ViT-L heads)
        self.fused_tail = FUSED_TAIL and fused_tail_applies(half, self.halfp, self.g296[0], H, pw_.split)
        self.fin = None if self.fused_tail else z(B, H + 2, W + 2, mm("oc2") * self.halfp)


# Head branches on side streams (round 6): the four reassemble -> input_projection -> layerN_rn chains (DA2/dpt.py:161-187) and the
# ResidualConvUnit 1 of levels 0-2 (util/blocks.py:131-133) are independent of each other until the refinenets join them; at small batches their launches
# fill 0.04-0.7 of a round of the 256 CUs (profiles/r05_e_config2_shapes.txt), so below HEAD_STREAMS_ROWS patch rows they are issued on four HIP streams
# (fork / join with events: capturable into the forward's graph) and the coarse levels run in the CUs the fine level leaves idle.  Same kernels, same
# arguments, same bits.  ADA_HEAD_STREAMS=0 / 1: never / always.
# _second_rung_first: relative guard band around the ladder's thresholds inside which the first rung's own r decides (the rungs' maps differ by ~1e-3 of their logits)
LADDER_GUARD = float(os.environ.get("ADA_LADDER_GUARD", "0.02"))
# ADA_LADDER_STICKY=0: always run the first rung's head first (A/B)
LADDER_STICKY = os.environ.get("ADA_LADDER_STICKY", "1") != "0"
HEAD_STREAMS = os.environ.get("ADA_HEAD_STREAMS", "auto")
HEAD_STREAMS_ROWS = int(os.environ.get("ADA_HEAD_STREAMS_ROWS", str(12 * 1369)))
GRAPH_MODE = os.environ.get("ADA_GRAPH", "auto")
# "auto": graph replay for calls of up to one 518x518 image.  Measured (profiles/r02_k_hip_graph_latency_ab.txt): the forward is device-bound
# from ViT-B upwards (replay == launches within 0.5 % at B = 1..8) and host-bound only for a single ViT-S image (3.55 -> 2.62 ms).
GRAPH_AUTO_PIXELS = int(os.environ.get("ADA_GRAPH_AUTO_PIXELS", str(518 * 518)))
GRAPH_AUTO_ENCODERS = tuple(e for e in os.environ.get("ADA_GRAPH_AUTO_ENCODERS", "vits").split(",") if e)
# Bounds of the per-shape caches (LRU): a variable-resolution stream of single images must not grow device memory without limit.  A shape is
# captured only on its SECOND sighting (a one-off resolution pays no extra warm-up forward, synchronise and empty_cache).
MAX_GRAPHS = int(os.environ.get("ADA_GRAPH_CACHE", "4"))
MAX_WORKSPACES = int(os.environ.get("ADA_WORKSPACE_CACHE", "6"))


class _GraphedForward:
    """The launch sequence of DepthEngine._forward for one input shape, captured into a HIP graph.  The forward is a fixed list of
    kernel launches over a pre-allocated workspace (no host synchronisation, no data-dependent control flow), so the capture is exact;
    inputs are copied into the graph's static buffers and the result is cloned out of its pool."""

    def __init__(self, eng: "DepthEngine", x: torch.Tensor, guide: Optional[torch.Tensor], norm: Optional[bool] = None):
        dev = x.device
        self.x = x.detach().contiguous().float().clone()
        self.guide = None if guide is None else guide.detach().contiguous().float().clone()
        # The captured launches carry raw pointers into this shape's Workspace (allocated by the warm-up forward below, in the ordinary
        # allocator pool, not the graph's): the graph must keep it alive itself.  Replays never pass through DepthEngine.workspace(), so
        # the engine's LRU sees a captured shape as idle and may drop its entry -- with this reference that only removes the dict entry.
        self.ws = eng.workspace(x.shape[0], x.shape[2], x.shape[3], dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):   # warm-up outside the capture: workspace, position table, per-kernel function attributes
            eng._forward(self.x, self.guide, norm)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: other host threads may keep using the device (a serving process) while this one captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = eng._forward(self.x, self.guide, norm)
        if eng.workspace(x.shape[0], x.shape[2], x.shape[3], dev) is not self.ws:   # cannot happen under the engine lock; never replay a graph over foreign memory
            raise RuntimeError("workspace changed during graph capture")
        self.lock = threading.Lock()

    def __call__(self, x: torch.Tensor, guide: Optional[torch.Tensor], post=None) -> torch.Tensor:
        # the static input / output buffers are shared by every caller of this shape: copy-in, replay and copy-out are one critical section
        # (`post(ws, out, x, guide)`: the engine's precision ladder, which reads the taps this replay left in the graph's workspace)
        with self.lock:
            self.x.copy_(x)
            if self.guide is not None:
                if guide is None or guide.shape != self.guide.shape:
                    raise HipExtError(f"guide tensor of shape {tuple(self.guide.shape)} required")
                self.guide.copy_(guide)
            self.graph.replay()
            out = self.out.clone()
            return out if post is None else post(self.ws, out, x, guide)


def ladder_curve(zk: torch.Tensor, z3: torch.Tensor, act: int):
    """The calibration's raw material (DepthEngine.calibrate; pure torch, any device): for every image [n, ...] and every shift d of the grid, (r, e) =
    (the image's sensitivity  sum w(z3 + d) / sum |f(z3 + d)|,  the metric  sum |f(zk + d) - f(z3 + d)| / sum |f(z3 + d)|  of rung k against the third),
    f / w = sigmoid / s(1-s), ReLU / [z > 0], identity / 1.  Returns R, E as [n, shifts] float64 tensors."""
    a, t = zk.flatten(1).double(), z3.flatten(1).double()
    if act == ACT_NONE:      # a shift of bare logits changes nothing but the denominator: one point per image, e = eps r exactly
        den = t.abs().sum(1).clamp_min(1e-300)
        return (t.shape[1] / den).unsqueeze(1), ((a - t).abs().sum(1) / den).unsqueeze(1)
    if act == ACT_SIGMOID:      # shifts that put the map's mean between ~0.9 and ~0.02 -- in units of the logits' own spread where that is wide, so that r reaches -> 1
        scale = t.std(dim=1, keepdim=True).clamp_min(1.0)
        grid = -t.median(dim=1, keepdim=True).values + scale * torch.linspace(-5.0, 2.0, 29, dtype=t.dtype, device=t.device)[None, :]
    else:                       # ReLU: shifts that leave 97 % ... 3 % of the map positive
        qs = torch.tensor([0.03, 0.08, 0.15, 0.25, 0.35, 0.5, 0.65, 0.75, 0.85, 0.92, 0.97], dtype=t.dtype, device=t.device)
        grid = -torch.quantile(t, qs, dim=1).t()
    R, Em = [], []
    for j in range(grid.shape[1]):
        d = grid[:, j:j + 1]
        if act == ACT_SIGMOID:
            fa, ft = torch.sigmoid(a + d), torch.sigmoid(t + d)
            wsum = (ft * (1 - ft)).sum(1)
        else:
            fa, ft = (a + d).clamp_min(0), (t + d).clamp_min(0)
            wsum = (ft > 0).double().sum(1)
        den = ft.sum(1).clamp_min(1e-300)
        R.append(wsum / den)
        Em.append((fa - ft).abs().sum(1) / den)
    return torch.stack(R, 1), torch.stack(Em, 1)


def ladder_thresholds(z1: torch.Tensor, z2: Optional[torch.Tensor], z3: torch.Tensor, act: int, budget: float, safety: float, rule: str = "cross") -> dict:
    """Thresholds of the precision ladder from the logits of the rungs on the calibration images (see DepthEngine.calibrate): per rung k the largest eps = e / r over
    the points, the `global` threshold budget / (safety eps_max) and the `cross` threshold -- the smallest r at which a calibration point has safety * e > budget (the
    rung's error AT the operating point where it would be left, not its worst anywhere; inf when no point exceeds the budget).  r from rung 1, r3 from rung 2 (no
    second rung: r3 from rung 1).  A sigmoid's r lives in (0, 1): its thresholds are capped at 0.97.  Pure torch: unit-tested on the CPU."""
    if rule not in ("cross", "global"):
        raise HipExtError(f"ladder_thresholds: rule={rule!r} (cross | global)")

    def one(zk):
        R, Em = ladder_curve(zk, z3, act)
        eps_max = float((Em / R.clamp_min(1e-300)).max())
        bad = Em * safety > budget
        return eps_max, budget / (safety * max(eps_max, 1e-30)), (float(R[bad].min()) if bool(bad.any()) else float("inf"))
    cap = (lambda v, lo: min(max(v, lo), 0.97)) if act == ACT_SIGMOID else (lambda v, lo: max(v, lo))
    pick = (lambda g, c: c if rule == "cross" else g)
    e1, g1, c1 = one(z1)
    res = dict(budget=budget, safety=safety, rule=rule, eps1=e1, r_global=g1, r_cross=c1, act={ACT_SIGMOID: "sigmoid", ACT_RELU: "relu", ACT_NONE: "none"}[act])
    if z2 is not None:
        e2, g2, c2 = one(z2)
        r = cap(pick(g1, c1), 0.02)
        res.update(eps2=e2, r3_global=g2, r3_cross=c2, r=r, r3=cap(pick(g2, c2), r))
    else:
        res.update(r3=cap(pick(g1, c1), 0.0))
    return res


class DepthEngine:
    """Runs one forward.  ``final_act``: 'sigmoid' | 'relu' | 'none'."""

    def __init__(self, weights: PackedWeights, final_act: str, normalise_input: bool, ladder: Optional[dict] = None):
        self.w = weights
        self.final_act = {"sigmoid": ACT_SIGMOID, "relu": ACT_RELU, "none": ACT_NONE}[final_act]
        self.normalise_input = normalise_input
        # Precision ladder of the sigmoid heads (see _escalate): dict(r=<threshold on sum s(1-s) / sum s>, make=<callable -> head-only PackedWeights
        # in split precision>) or None.  The first rung -- this engine's own weights -- must keep its taps [hi | lo] (PackedWeights.tap_split).
        # Heads without a sigmoid keep the third rung's token-diversity trigger only: dict(div=..., make3=...) (DA2/dpt.py::_flat_input_rung).
        self.ladder = ladder
        self._warned = False          # the one-time notice that images are being re-run
        self.second_rung_first_calls = 0     # calls that ran the second rung's head first (diagnostic)
        self.third_rung_first_calls = 0      # ... the third-rung engine first
        self._start_rung = 1          # which head runs first (_run_ladder): 2 after a call most of whose images left the first rung
        if self.ladder is not None and "make" in self.ladder and not weights.tap_split and "proj" not in weights.split:
            raise HipExtError("precision ladder: the engine's weights must be packed with tap_split=True")
        self._w_hi: Optional[PackedWeights] = None
        self._ws_hi: "OrderedDict[tuple, Workspace]" = OrderedDict()
        self.block_probe = None       # diagnostic hook: callable(block index, workspace) behind every transformer block (never set on the product path)
        self._eng3: Optional["DepthEngine"] = None       # third rung: an engine with every encoder block and the whole head in split precision
        self.escalated = 0            # images the ladder has re-run so far (second + third rung)
        self.escalated3 = 0           # ... of which on the third rung
        self.tap_f8 = bool(weights.tap_f8)   # the taps (and with them the second rung's products) carry fp8 correction terms
        self.last_ratio = None        # per-image sum s(1-s) / sum s of the most recent call (CPU tensor), None when the ladder is off
        self.last_diversity = None    # per-image token diversity of the last tap (see _escalate)
        self.last_input_diversity = None   # ... and of the patchified input
        self._ws: "OrderedDict[tuple, Workspace]" = OrderedDict()
        self._graphs: "OrderedDict[tuple, object]" = OrderedDict()   # key -> _GraphedForward | False (capture refused) | int (sightings)
        self._lock = threading.Lock()
        self._streams: Dict[str, list] = {}      # device -> the three side streams of the head's forked section

    def max_batch(self, H: int, W: int) -> int:
        if H * W > MAX_ROWS:
            raise HipExtError(f"a {H}x{W} image has more than 2^24 pixels, the row limit of one kernel launch: "
                              "use hip_ext.tiling.tiled_amodal_forward / tiled_raw_forward")
        return max(1, MAX_ROWS // (H * W))

    def workspace(self, B, H, W, device) -> Workspace:
        key = (B, H, W, str(device))
        ws = self._ws.get(key)
        if ws is None:
            while len(self._ws) >= max(1, MAX_WORKSPACES):
                self._ws.popitem(last=False)      # least recently used shape; a captured graph holds its own reference (_GraphedForward.ws)
            ws = Workspace(self.w, B, H, W, device)
            self._ws[key] = ws
        else:
            self._ws.move_to_end(key)
        return ws

    def _side_streams(self, device):
        key = str(device)
        st = self._streams.get(key)
        if st is None:
            st = self._streams[key] = [torch.cuda.Stream(device=device) for _ in range(3)]
        return st

    def saturation_report(self, x: torch.Tensor, guide: Optional[torch.Tensor]):
        """Diagnostic forward, not the hot path: runs ``_forward`` with a probe behind every launch that writes an operand-typed tensor and
        returns ``(output, report)``; ``report`` maps "<kernel> #<launch index> [rows x cols]" to the number of elements that sit at the fp16
        clamp (+-65504: to_op saturates instead of overflowing to inf) or are non-finite.  An empty report = nothing saturated.  Meant for
        real checkpoints with massive activations (the synthetic fills never get near the clamp)."""
        global _probe
        _probe = []
        try:
            out = self._forward(x, guide)
            torch.cuda.synchronize(x.device)
            rep = {label: int(c.item()) for label, c in _probe if int(c.item())}
        finally:
            _probe = None
        return out, rep

    # ---- small helpers over igemm ---------------------------------------------------------
    @staticmethod
    def _kdup(a_width: int, w: torch.Tensor, taps: int = 1, a_seg: int = 0) -> dict:
        """K / lda / a_dup_seg of a contraction whose A rows are ``a_width`` wide and whose packed weights are ``w`` [N, K]: either a plain
        operand (K == taps * a_width) or a split one -- [hi | lo] activations against [w_hi | w_hi | w_lo] weights, 2 K == 3 taps a_width,
        the third k segment re-reading the first (ada_igemm a_dup_seg)."""
        K = int(w.shape[1])
        word = getattr(w, "f8_scales", 0)
        if word:      # [w_hi | w_hi8 | w_lo8] per tap against [hi | lo8 | hi8] rows: a straight walk whose second half goes to the fp8 pipe (ada_igemm_args.f8_from)
            seg = abs(a_seg) or a_width // 2
            if K != 2 * taps * seg or (a_seg > 0):
                raise HipExtError(f"fp8-form weights with K={K} do not fit an operand with segments of {a_seg or seg} ({taps} tap(s))")
            return dict(K=K, lda=a_width, f8_from=seg, f8_mid=seg + seg // 2, f8_scales=word)
        if a_seg < 0:
            if K == 3 * -a_seg:
                raise HipExtError("three-term fp16 weights against a [hi | lo8 | hi8] operand")
            a_seg = -a_seg
        if a_seg:     # the A rows ARE [hi | lo] segments of width a_seg (a tap of an engine that keeps them so): the weights decide what is walked
            if K == 3 * a_seg:
                return dict(K=K, lda=a_width, a_dup_seg=a_seg)
            if K == a_seg:                       # single-precision contraction: the hi half only
                return dict(K=K, lda=a_width, a_dup_seg=0)
            if K == 2 * a_seg:                   # [w_hi | w_lo] against the hi half walked twice
                return dict(K=K, lda=a_width, a_dup_seg=0, a_wrap=a_seg)
            raise HipExtError(f"packed weights with K={K} do not fit a [hi | lo] operand with segments of {a_seg}")
        if K == taps * a_width:
            return dict(K=K, lda=a_width, a_dup_seg=0)
        if 2 * K == 3 * taps * a_width:
            return dict(K=K, lda=a_width, a_dup_seg=a_width // 2)
        if taps == 1 and K == 2 * a_width:     # [w_hi | w_lo] against a plain operand walked twice (weight-only split)
            return dict(K=K, lda=a_width, a_dup_seg=0, a_wrap=a_width)
        raise HipExtError(f"packed weights with K={K} do not fit an operand of width {a_width} ({taps} tap(s))")

    @classmethod
    def _conv3(cls, src_pad, w, M, N, grid, stride=1, cin=None, k_alg=None, **kw):
        """3x3 conv over a zero-bordered NHWC tensor (plain or split input, see _kdup)."""
        B, Hp, Wp, Cp = src_pad.shape
        kd = cls._kdup(Cp, w, taps=9)
        k_igemm(M=M, N=N, k_alg=k_alg if k_alg is not None else 9 * (cin or kd.get("a_dup_seg") or kd.get("f8_from") or Cp), A=src_pad, W=w, a_mode=A_CONV3,
                conv=(grid[0], grid[1], Hp, Wp, stride), **kd, **kw)

    def forward(self, x: torch.Tensor, guide: Optional[torch.Tensor], normalise: Optional[bool] = None) -> torch.Tensor:
        """One forward.  ``normalise`` overrides the engine's default for the fused ImageNet normalisation of the input (the
        patchify kernel applies (x - mean) / std on the fly: callers holding a [0, 1] image need no separate normalise pass).  The smallest problems are bound by the host's launch rate (~230 launches per image): for those the launch
        sequence is captured once per input shape into a HIP graph and replayed, bit-identically (``ADA_GRAPH`` = auto | 1 | 0)."""
        if not x.is_cuda:
            raise HipExtError("DepthEngine.forward: input must live on a HIP device (no CPU fallback in the product path)")
        norm = self.normalise_input if normalise is None else bool(normalise)
        mode = GRAPH_MODE
        # (round 6: with the head's branches on side streams the plain launches of a single ViT-B / ViT-L image are FASTER than the replay -- 2.68 / 5.27 ms against
        #  3.00 / 5.84 ms, profiles/r06_g_* -- so "auto" replays only the model that is bound by the host's launch rate: ViT-S)
        use = mode == "1" or (mode == "auto" and self.w.encoder in GRAPH_AUTO_ENCODERS and x.shape[0] * x.shape[-2] * x.shape[-1] <= GRAPH_AUTO_PIXELS)
        # replay bypasses the Python wrappers: with a KernelTimer or a tile log attached the launches must be issued one by one
        if not use or instrumented() or torch.cuda.is_current_stream_capturing():
            return self._run_ladder(x, guide, norm)
        # the kernel variant / tile override / fused-tail switch in force at capture time are baked into the graph
        key = (tuple(x.shape), None if guide is None else tuple(guide.shape), str(x.device), debug_epoch(), FUSED_TAIL, SUBPIXEL, OC1_COMMUTE, norm)
        with self._lock:
            g = self._graphs.get(key)
            if g is not None:
                self._graphs.move_to_end(key)
            if g is None or isinstance(g, int):
                seen = (g or 0) + 1
                if mode != "1" and seen < 2:       # first sighting of this shape: plain launches, capture when it comes back
                    self._graphs[key] = seen
                    g = False
                else:
                    try:
                        g = _GraphedForward(self, x, guide, norm)
                    except RuntimeError as e:   # capture refused (e.g. an enclosing capture in another mode): same kernels, launched one by one
                        import warnings
                        warnings.warn(f"HIP-graph capture of the forward failed ({e}); this shape runs as plain launches")
                        g = False
                    self._graphs[key] = g
                while len(self._graphs) > max(1, MAX_GRAPHS):
                    self._graphs.popitem(last=False)
        if g is False:
            return self._run_ladder(x, guide, norm)
        return g(x, guide, (lambda ws_, out_, x_, g_: self._escalate(ws_, out_, x_, g_, norm)) if self.ladder is not None else None)

    def _escalate(self, ws: Optional[Workspace], out: torch.Tensor, x: Optional[torch.Tensor] = None, guide: Optional[torch.Tensor] = None,
                  norm: Optional[bool] = None) -> torch.Tensor:
        """Second rung of the precision ladder (sigmoid heads).  The default policy runs the DPT head on single fp16 operands and
        counts on the sigmoid to compress the logit error it leaves (~1.2e-3 mean absolute): in the north-star metric mean|a - b| / mean|b| that error
        arrives multiplied by  r = sum s (1 - s) / sum s  of the image -- 0.3-0.5 for maps that span (0, 1), -> 1 for maps concentrated near 0, where the
        single-precision head exceeds the 1e-3 bar (reference fixtures vitl_518_m10, vitb_518_zeros, ...).  r is computed from the output itself
        (ada_depth_stats_fwd, one deterministic reduction); for the images whose r exceeds the threshold the HEAD ONLY (27 % of the FLOPs) is re-run
        in split precision from the four taps, which this engine keeps [hi | lo] for the purpose, and its result replaces theirs.  The decision is a
        pure function of the image's own first-rung results: deterministic, independent of the rest of the batch.  Second trigger: the TOKEN DIVERSITY
        of the last tap (ada_token_diversity_fwd: sum of the feature columns' variances over the image's patch tokens / sum of their mean squares) --
        0.23-0.53 on noise and image-like inputs, 0.02 on constant and checkerboard images, where every patch token is the same up to its position
        and the head's rounding errors add coherently over positions (ViT-B, all-zero input at output mean 0.50: 1.29e-3 on the first rung, 3.6e-4 on
        the second; profiles/r05_h_*) -- such an image now goes to the THIRD rung, below.  Costs one host read of ~50 floats per image (the forward's only synchronisation); off under stream capture
        (a caller's own HIP graph cannot hold a data-dependent branch).
        THIRD RUNG (round 5, held-out draws of the output-range sweep): at the very bottom of the range -- maps averaging 0.03-0.06, r > 0.75 -- nothing is
        left of the sigmoid's compression and the second rung's own logit error (0.8e-3 ... 1.2e-3, the ENCODER's fp16 operand rounding: ViT-B on a constant
        image at mean 0.03: 1.1e-3 with the whole head in split precision) is what shows.  Images with r above lad["r3"] are therefore run again as a
        whole, encoder blocks and head in split precision (lad["make3"]: an engine of its own, built on first use; 1.6e-4 ... 3.7e-4 on the cases that
        exposed it, profiles/r05_v_third_rung_probe.txt), instead of taking the second rung."""
        lad = self.ladder
        if lad is None:
            return out
        if torch.cuda.is_current_stream_capturing():
            return out
        B, H, W = out.shape[0], out.shape[-2], out.shape[-1]
        if ws is None:
            ws = self.workspace(B, H, W, out.device)
        D = self.w.dim
        # r: the sensitivity of the metric to a logit error, from the output itself -- sum s(1-s) / sum s behind a sigmoid, (positive outputs) / sum out behind a
        # ReLU (round 6: a clipped pixel carries no error; a map that is mostly clipped with the rest just above the kink has a small denominator), N / sum |out|
        # for bare logits.  Thresholds: lad["r"] (second rung: head re-run; sigmoid policy) and lad["r3"] (third rung), both CALIBRATED per checkpoint (calibrate()).
        has_r = "r" in lad or "r3" in lad
        has_div = lad.get("div", 0.0) > 0.0
        if has_r:
            k_depth_stats(out, ws.stat_sums, self.final_act)
        if has_div:
            k_token_diversity(ws.taps[3], ws.taps[3].shape[1], B, ws.ph * ws.pw, D, ws.stat_div)
        if ws.stat_in is not None:
            k_token_diversity(ws.a_pe, ws.a_pe.shape[1], B, ws.ph * ws.pw, self.w.pe_seg, ws.stat_in)
        host = ws.stat_buf.cpu().double()                       # the forward's one synchronisation: (8 + D / 64 + patch width / 64) x 2 floats per image
        nst, ndv = B * STAT_CHUNKS * 2, B * ((D + 63) // 64) * 2
        st, dv = host[:nst].view(B, STAT_CHUNKS, 2).sum(1), host[nst:nst + ndv].view(B, -1, 2).sum(1)
        ratio = st[:, 1] / st[:, 0].clamp_min(1e-300) if has_r else torch.zeros(B, dtype=torch.float64)
        self.last_ratio = ratio if has_r else None
        flat = torch.zeros(B, dtype=torch.bool)
        self.last_diversity = None
        if has_div:
            self.last_diversity = dv[:, 0] / dv[:, 1].clamp_min(1e-300)
            flat = self.last_diversity < lad["div"]
        if ws.stat_in is not None:
            # input-side trigger: the patches of the image are all alike (variance over patches below 1e-4 of their mean square) -- the raw models' last tap does not
            # tell (synthetic raw ViT-B at 518^2: token diversity 0.18 on an all-zero image, 0.11 on a noise image; profiles/r05_aa_*), the input does
            di = host[nst + ndv:].view(B, -1, 2).sum(1)
            self.last_input_diversity = di[:, 0] / di[:, 1].clamp_min(1e-300)
            flat = flat | (self.last_input_diversity < lad.get("div_in", 1e-4))
        trigger = ((ratio > lad["r"]) if ("r" in lad and "make" in lad) else torch.zeros_like(flat)) | (flat if "make" in lad else torch.zeros_like(flat))
        # constant / checkerboard inputs take the third rung too: their rounding errors add coherently in the ENCODER as well (ViT-B, all-zero image at 126 x 154:
        # 8.5e-4 with the head in split precision, 2.3e-4 with everything), and what a degenerate input costs does not matter
        top = (((ratio > lad["r3"]) if "r3" in lad else torch.zeros_like(flat)) | flat) if ("make3" in lad and x is not None) else torch.zeros_like(trigger)
        if not self._warned and bool((trigger | top).any()):
            self._warned = True
            import warnings
            warnings.warn(f"precision ladder: {int((trigger | top).sum())} of {B} image(s) re-run in split precision (second rung: head only; third: whole forward) -- "
                          f"r up to {float(ratio.max()):.3g} against thresholds {lad.get('r')} / {lad.get('r3')}; counted in DepthEngine.escalated / escalated3 "
                          "(this message appears once per engine; module.precision_ladder = False switches the ladder off)")
        # which rung the NEXT call starts on (_run_ladder): the one most of this call's images ended on
        self._start_rung = 3 if 2 * int(top.sum()) > B else (2 if ("make" in lad and 2 * int((trigger & ~top).sum()) > B) else 1)
        idx3 = torch.nonzero(top).flatten()
        if idx3.numel() > 0:      # third rung: the whole forward in split precision for these images, straight from the inputs
            if self._eng3 is None:
                self._eng3 = lad["make3"]()
            if idx3.numel() == B:
                self.escalated += B
                self.escalated3 += B
                return self._eng3.forward(x, guide, norm)
            sel3 = idx3.to(out.device)
            out3 = self._eng3.forward(x.index_select(0, sel3), None if guide is None else guide.index_select(0, sel3), norm)
            out.index_copy_(0, sel3, out3)
            self.escalated += int(idx3.numel())
            self.escalated3 += int(idx3.numel())
        idx = torch.nonzero(trigger & ~top).flatten()
        if idx.numel() == 0:
            return out
        if self._w_hi is None:
            self._w_hi = lad["make"]()
        Be = int(idx.numel())
        hi = self._head_for(ws, B, None if Be == B else idx, self._w_hi)
        if Be == B:
            out = hi
        else:
            out.index_copy_(0, idx.to(out.device), hi)
        self.escalated += Be
        return out

    def _head_for(self, ws: Workspace, B: int, idx: Optional[torch.Tensor], w: PackedWeights) -> torch.Tensor:
        """The DPT head with the weights ``w`` (this engine's own = first rung, or the second rung's) for the images ``idx`` (CPU index tensor; None = all B) of the
        taps in ``ws``, on a head-only workspace of its own (cached per shape)."""
        if idx is None and w is self.w:
            return self._head(ws, B)
        Be = B if idx is None else int(idx.numel())
        dev = ws.taps[0].device
        key = (w is self.w, Be, ws.H, ws.W, str(dev))
        ws2 = self._ws_hi.get(key)
        if ws2 is None:
            while len(self._ws_hi) >= max(1, MAX_WORKSPACES):
                self._ws_hi.popitem(last=False)
            ws2 = Workspace(w, Be, ws.H, ws.W, dev, head_only=True)
            self._ws_hi[key] = ws2
        else:
            self._ws_hi.move_to_end(key)
        own = ws2.taps
        own_cls = getattr(ws2, "cls_op", None)      # use_clstoken models: the read-out also reads the final-LayerNorm'd class tokens of the first rung
        try:
            if idx is None:
                ws2.taps = ws.taps        # every image of the batch: the head reads the encoder's taps in place
                if own_cls is not None:
                    ws2.cls_op = ws.cls_op
            else:
                sel = idx.to(dev)
                for t in range(4):
                    torch.index_select(ws.taps[t].view(B, -1), 0, sel, out=own[t].view(Be, -1))
                    if own_cls is not None:
                        torch.index_select(ws.cls_op[t], 0, sel, out=own_cls[t])
            return self._head(ws2, Be, w)
        finally:
            ws2.taps = own
            if own_cls is not None:
                ws2.cls_op = own_cls

    def _run_ladder(self, x: torch.Tensor, guide: Optional[torch.Tensor], norm: Optional[bool]) -> torch.Tensor:
        """First rung + the ladder's check -- or, when most images of this engine's PREVIOUS call left the first rung (maps near 0: a stream of such images pays
        12 ms of first-rung head per bs=32 step for nothing), the second rung first (_second_rung_first).  Which head runs first changes what a call costs, never
        what it returns: every image carries exactly the output of the rung the first-rung-first order assigns it."""
        lad = self.ladder
        if lad is None or not LADDER_STICKY or self._start_rung == 1 or torch.cuda.is_current_stream_capturing():
            return self._escalate(None, self._forward(x, guide, norm), x, guide, norm)
        if self._start_rung == 3 and "make3" in lad and lad.get("r3", float("inf")) < float("inf"):
            return self._third_rung_first(x, guide, norm)
        if self._start_rung == 2 and "make" in lad:
            return self._second_rung_first(x, guide, norm)
        return self._escalate(None, self._forward(x, guide, norm), x, guide, norm)

    def _third_rung_first(self, x: torch.Tensor, guide: Optional[torch.Tensor], norm: Optional[bool]) -> torch.Tensor:
        """Most images of the previous call ended on the THIRD rung (a checkpoint / operating point whose first rung never suffices: the bf16 build, raw maps that are
        mostly clipped): the everything-split engine runs first for the whole batch and r is read from ITS output.  Images with r3 > threshold (1 + g), or flat
        inputs, keep it -- exactly what the first-rung-first order returns for them; every other image goes through the ordinary order (first rung, then whatever its
        own r1 says).  A third-rung stream costs the third rung, not the sum of the rungs in front of it."""
        lad = self.ladder
        if self._eng3 is None:
            self._eng3 = lad["make3"]()
        eng3 = self._eng3
        out = eng3.forward(x, guide, norm)
        B, H, W = out.shape[0], out.shape[-2], out.shape[-1]
        ws3 = eng3.workspace(B, H, W, out.device)
        r3v = self._ratio_of(out)
        D = self.w.dim
        flat = torch.zeros(B, dtype=torch.bool)
        stat = torch.zeros(B * max((D + 63) // 64, ws3.a_pe.shape[1] // 128 + 1) * 2, dtype=torch.float32, device=out.device)
        if lad.get("div", 0.0) > 0.0:
            G = (D + 63) // 64
            k_token_diversity(ws3.taps[3], ws3.taps[3].shape[1], B, ws3.ph * ws3.pw, D, stat[:B * G * 2].view(B, G, 2))
            dv = stat[:B * G * 2].view(B, G, 2).cpu().double().sum(1)
            self.last_diversity = dv[:, 0] / dv[:, 1].clamp_min(1e-300)
            flat = self.last_diversity < lad["div"]
        Gi = eng3.w.pe_seg // 64
        k_token_diversity(ws3.a_pe, ws3.a_pe.shape[1], B, ws3.ph * ws3.pw, eng3.w.pe_seg, stat[:B * Gi * 2].view(B, Gi, 2))
        di = stat[:B * Gi * 2].view(B, Gi, 2).cpu().double().sum(1)
        self.last_input_diversity = di[:, 0] / di[:, 1].clamp_min(1e-300)
        flat = flat | (self.last_input_diversity < lad.get("div_in", 1e-4))
        sure = flat | (r3v > lad["r3"] * (1.0 + LADDER_GUARD))
        ratio = r3v.clone()
        n3 = int(sure.sum())
        rest = torch.nonzero(~sure).flatten()
        rung_rest = None
        if rest.numel() > 0:
            sel = rest.to(out.device)
            xs, gs = x.index_select(0, sel), (None if guide is None else guide.index_select(0, sel))
            e0, e3 = self.escalated, self.escalated3
            sub = self._escalate(None, self._forward(xs, gs, norm), xs, gs, norm)
            out.index_copy_(0, sel, sub)
            ratio[rest] = self.last_ratio if self.last_ratio is not None else r3v[rest]
            rung_rest = (self.escalated - e0, self.escalated3 - e3)
        self.last_ratio = ratio
        self.escalated += n3
        self.escalated3 += n3
        self.third_rung_first_calls += 1
        on3 = n3 + (rung_rest[1] if rung_rest else 0)
        on2 = (rung_rest[0] - rung_rest[1]) if rung_rest else 0
        self._start_rung = 3 if 2 * on3 > B else (2 if ("make" in lad and 2 * on2 > B) else 1)
        return out

    def _ratio_of(self, out: torch.Tensor) -> torch.Tensor:
        """r of every image of ``out`` (see _escalate) -- one reduction + one host read."""
        sums = torch.empty(out.shape[0], STAT_CHUNKS, 2, dtype=torch.float32, device=out.device)
        k_depth_stats(out, sums, self.final_act)
        st = sums.cpu().double().sum(1)
        return st[:, 1] / st[:, 0].clamp_min(1e-300)

    def _second_rung_first(self, x: torch.Tensor, guide: Optional[torch.Tensor], norm: Optional[bool]) -> torch.Tensor:
        """Encoder -> SECOND-rung head for the whole batch -> r from its output.  The two rungs' maps differ by ~1e-3 of their logits, so r2 decides for every image
        that is not within LADDER_GUARD of a threshold: r2 > r (1 + g) keeps the second rung's result (what the first-rung-first order would have computed for it,
        bit for bit -- the same head over the same taps), r2 > r3 (1 + g) or a flat input goes to the third rung.  Every other image -- below the threshold, or inside a
        guard band -- gets the FIRST-rung head run for it and is decided by its r1 exactly as _escalate decides: the output is a pure function of the image either way."""
        lad = self.ladder
        ws = self._forward(x, guide, norm, head=False)
        B, H, W = ws.B, ws.H, ws.W
        D = self.w.dim
        if self._w_hi is None:
            self._w_hi = lad["make"]()
        out = self._head_for(ws, B, None, self._w_hi)
        has_div = lad.get("div", 0.0) > 0.0
        k_depth_stats(out, ws.stat_sums, self.final_act)
        if has_div:
            k_token_diversity(ws.taps[3], ws.taps[3].shape[1], B, ws.ph * ws.pw, D, ws.stat_div)
        if ws.stat_in is not None:
            k_token_diversity(ws.a_pe, ws.a_pe.shape[1], B, ws.ph * ws.pw, self.w.pe_seg, ws.stat_in)
        host = ws.stat_buf.cpu().double()
        nst, ndv = B * STAT_CHUNKS * 2, B * ((D + 63) // 64) * 2
        st, dv = host[:nst].view(B, STAT_CHUNKS, 2).sum(1), host[nst:nst + ndv].view(B, -1, 2).sum(1)
        r2 = st[:, 1] / st[:, 0].clamp_min(1e-300)
        flat = torch.zeros(B, dtype=torch.bool)
        self.last_diversity = None
        if has_div:
            self.last_diversity = dv[:, 0] / dv[:, 1].clamp_min(1e-300)
            flat = self.last_diversity < lad["div"]
        if ws.stat_in is not None:
            di = host[nst + ndv:].view(B, -1, 2).sum(1)
            self.last_input_diversity = di[:, 0] / di[:, 1].clamp_min(1e-300)
            flat = flat | (self.last_input_diversity < lad.get("div_in", 1e-4))
        g = LADDER_GUARD
        thr, thr3 = lad["r"], lad.get("r3", float("inf"))
        has3 = "make3" in lad
        near3 = ((r2 / thr3 - 1.0).abs() <= g) if (has3 and thr3 < float("inf")) else torch.zeros_like(flat)
        need1 = ~flat & ((r2 <= thr * (1.0 + g)) | near3)          # the first rung decides (and, below the threshold, IS the result)
        rung = torch.full((B,), 2, dtype=torch.int64)
        if has3:
            rung[flat | (r2 > thr3 * (1.0 + g))] = 3
        ratio = r2.clone()
        idx1 = torch.nonzero(need1).flatten()
        out1 = None
        if idx1.numel() > 0:
            out1 = self._head_for(ws, B, None if idx1.numel() == B else idx1, self.w)
            r1 = self._ratio_of(out1)
            ratio[idx1] = r1
            rung[idx1] = torch.where((r1 > thr3) & has3, torch.tensor(3), torch.where(r1 > thr, torch.tensor(2), torch.tensor(1)))
        self.last_ratio = ratio
        sel1 = torch.nonzero(rung == 1).flatten()
        if sel1.numel() > 0:      # the first rung's maps go where the first rung stands
            pos = {int(v): j for j, v in enumerate(idx1.tolist())}
            src = torch.tensor([pos[int(v)] for v in sel1.tolist()], device=out.device)
            out.index_copy_(0, sel1.to(out.device), out1.index_select(0, src))
        sel3 = torch.nonzero(rung == 3).flatten()
        if sel3.numel() > 0:
            if self._eng3 is None:
                self._eng3 = lad["make3"]()
            s3 = sel3.to(out.device)
            out3 = self._eng3.forward(x.index_select(0, s3), None if guide is None else guide.index_select(0, s3), norm)
            out.index_copy_(0, s3, out3)
        n2 = int((rung >= 2).sum())
        self.escalated += n2
        self.escalated3 += int(sel3.numel())
        self.second_rung_first_calls += 1
        self._start_rung = 3 if 2 * int((rung == 3).sum()) > B else (2 if 2 * int((rung == 2).sum()) > B else 1)
        if not self._warned and n2:
            self._warned = True
            import warnings
            warnings.warn(f"precision ladder: {n2} of {B} image(s) on the second / third rung (r up to {float(ratio.max()):.3g} against thresholds {thr} / {lad.get('r3')}); "
                          "counted in DepthEngine.escalated / escalated3 (this message appears once per engine)")
        return out

    def calibrate(self, x: torch.Tensor, guide: Optional[torch.Tensor], budget: float = 9e-4, safety: float = 1.1, flat_index: Optional[int] = None,
                  rule: str = "cross") -> dict:
        """Self-calibration of the ladder's thresholds for THIS checkpoint, on the device, with no oracle (round 6; round 5's thresholds were ~12 constants fitted
        to synthetic weights).  The calibration images ``x`` ([0, 1] RGB; ``guide`` with the model's guide channels) run through the first rung, through the second
        (head re-run from the first rung's taps, where the policy has one) and through the third-rung engine -- every encoder block and the whole head in split
        precision, 1.6-3.7e-4 from the reference where it was measured -- with the final activation switched OFF, i.e. as logits z1, z2, z3.  For a grid of bias
        shifts d (the final bias is added in fp32: moving it moves the map's operating point and nothing else) the metric of rung k against the third,
        e = mean|f(z_k + d) - f(z_3 + d)| / mean f(z_3 + d), is known together with the image's r (see _escalate); e / r is the rung's sensitivity-normalised
        logit error eps_k.  Rule "cross" (default): a rung is left at the smallest r at which any calibration point has safety * e > budget; rule "global":
        at budget / (safety * max eps_k) -- the worst eps anywhere on the grid (round 5's hand-fitted constants, 0.42 / 0.45, are what "global" returns for the
        synthetic fills they were fitted on at budget 9e-4).  r from rung 1, r3 from rung 2 (heads without a second rung: r3 from rung 1).  ``flat_index``: a
        constant image of the batch -- left out of eps, used to place the tap-diversity threshold between it and the other images (geometric mean), or to switch that
        trigger off where the last tap does not tell them apart (outlier-dominated tokens).  Returns the numbers; the caller installs them (DA2/dpt.py)."""
        lad = self.ladder
        if lad is None or "make3" not in lad:
            raise HipExtError("calibrate: this engine has no precision ladder")
        B, H, W = x.shape[0], x.shape[-2], x.shape[-1]
        if self._eng3 is None:
            self._eng3 = lad["make3"]()
        eng3 = self._eng3
        act = self.final_act
        keep = (self.final_act, eng3.final_act)
        z2 = None
        try:
            self.final_act = eng3.final_act = ACT_NONE
            z1 = self._forward(x, guide, True)
            ws = self.workspace(B, H, W, x.device)
            k_token_diversity(ws.taps[3], ws.taps[3].shape[1], B, ws.ph * ws.pw, self.w.dim, ws.stat_div)
            dv = ws.stat_div.double().sum(1)
            tap_div = (dv[:, 0] / dv[:, 1].clamp_min(1e-300)).cpu()
            if "make" in lad:
                if self._w_hi is None:
                    self._w_hi = lad["make"]()
                ws2 = Workspace(self._w_hi, B, H, W, x.device, head_only=True)
                ws2.taps = ws.taps
                if getattr(ws2, "cls_op", None) is not None:
                    ws2.cls_op = ws.cls_op
                z2 = self._head(ws2, B, self._w_hi)
            z3 = eng3._forward(x, guide, True)
        finally:
            self.final_act, eng3.final_act = keep
        use = [i for i in range(B) if i != flat_index]
        sel = torch.tensor(use, device=x.device)
        res = ladder_thresholds(z1.index_select(0, sel), None if z2 is None else z2.index_select(0, sel), z3.index_select(0, sel), act, budget, safety, rule)
        res.update(images=len(use), size=(H, W))
        if flat_index is not None:
            dmin, dflat = float(tap_div[use].min()), float(tap_div[flat_index])
            res.update(tap_diversity_images_min=dmin, tap_diversity_flat=dflat, div=math.sqrt(dmin * dflat) if dflat < 0.25 * dmin else 0.0)
        return res

    def _forward(self, x: torch.Tensor, guide: Optional[torch.Tensor], norm: Optional[bool] = None, head: bool = True):
        """The first rung: encoder + DPT head -> depth map.  ``head=False``: the encoder only -- returns the Workspace holding the four taps (_second_rung_first)."""
        w = self.w
        if not x.is_cuda:
            raise HipExtError("DepthEngine.forward: input must live on a HIP device (no CPU fallback in the product path)")
        B, C, H, W = x.shape
        assert C == 3
        assert H % PATCH == 0, f"Input image height {H} is not a multiple of patch height {PATCH}"
        assert W % PATCH == 0, f"Input image width {W} is not a multiple of patch width: {PATCH}"
        if w.guided:
            if guide is None or guide.shape[1] != w.guide_channels:
                raise HipExtError(f"guide tensor with {w.guide_channels} channels required")
            guide = guide.contiguous().float()
        x = x.contiguous().float()
        ws = self.workspace(B, H, W, x.device)
        D, heads = w.dim, w.heads
        ph, pw = ws.ph, ws.pw
        Np = ph * pw
        N = Np + 1
        T, P = B * N, B * Np

        # ---- tokens: patchify (+normalise) -> one GEMM over [rgb | guide] -> + bias + pos, cls row ----------
        norm = self.normalise_input if norm is None else norm
        mean = (0.485, 0.456, 0.406) if norm else None
        inv_std = (1 / 0.229, 1 / 0.224, 1 / 0.225) if norm else None
        k_patchify(x, guide if w.guided else None, B, w.guide_channels, H, W, mean, inv_std, ws.a_pe, 2 * w.pe_seg, split=True)
        pos = w.pos_embed(ph, pw)
        k_igemm(M=P, N=D, K=w.pe_k, k_alg=(3 + w.guide_channels) * 196, A=ws.a_pe, lda=2 * w.pe_seg, a_dup_seg=w.pe_seg, W=w.pe_w, bias=w.pe_b, res=pos, ldr=D, res_row_mod=Np, res_row_off=1,
                flags=EP_BIAS | EP_RESIDUAL, out_f32=ws.x, ldo_f32=D, map_f32=MAP_TOKEN, map_h=Np)
        k_write_cls(ws.x, B, N, D, w.cls, pos)

        # ---- transformer blocks ------------------------------------------------------------------
        taps = TAPS[w.encoder]
        ldy = ws.y.shape[1]

        def a_ln(blk_, wname):      # A-operand arguments of a linear layer that reads the LayerNorm output ws.y: plain, or [hi | lo] / [hi | lo8 | hi8] in a split block
            if not blk_["esplit"]:
                return dict(K=D, lda=ldy)
            if w.enc_f8:
                return dict(K=2 * D, lda=ldy, f8_from=D, f8_mid=D + D // 2, f8_scales=blk_[wname.replace("_w", "_f8")])
            return dict(K=3 * D, lda=ldy, a_dup_seg=D)

        ldo, ldh = ws.o.shape[1], ws.hd.shape[1]

        def a_act(blk_, kin, ld, wname):       # ... that reads the attention output / the MLP hidden (width kin, row stride ld): plain, or the split forms its producer wrote
            if not blk_["esplit"]:
                return dict(K=kin, lda=ld)
            if w.enc_f8:
                return dict(K=2 * kin, lda=ld, f8_from=kin, f8_mid=kin + kin // 2, f8_scales=blk_[wname.replace("_w", "_f8")])
            return dict(K=3 * kin, lda=ld, a_dup_seg=kin)

        def seg_act(blk_, kin):     # split_seg of the producer of such an activation
            return (-kin if w.enc_f8 else kin) if blk_["esplit"] else 0

        def seg(blk_):              # split_seg of the LayerNorm that feeds blk_'s linear layers
            return (-D if w.enc_f8 else D) if blk_["esplit"] else 0
        ln1_done = False    # block i's norm1 output already sits in ws.y (emitted by the tap LayerNorm of block i - 1, see below)
        for i, blk in enumerate(w.blocks):
            last = i == len(w.blocks) - 1
            if ln1_done:
                ln1_done = False
                k_igemm(M=T, N=3 * D, k_alg=D, A=ws.y, W=blk["qkv_w"], bias=blk["qkv_b"], flags=EP_BIAS, out_op=ws.qkv, ldo_op=3 * D, **a_ln(blk, "qkv_w"))
            else:
                k_layernorm(ws.x, D, T, D, blk["ln1_w"], blk["ln1_b"], LN_EPS, out_op=ws.y, ld_op=ldy, split_seg=seg(blk))
                k_igemm(M=T, N=3 * D, k_alg=D, A=ws.y, W=blk["qkv_w"], bias=blk["qkv_b"], flags=EP_BIAS, out_op=ws.qkv, ldo_op=3 * D, **a_ln(blk, "qkv_w"))
            k_attention(ws.qkv, ws.o, B, N, heads, ld_out=ldo if ldo != D else 0, split_seg=seg_act(blk, D))
            hid = blk["hidden"]
            k_igemm(M=T, N=D, k_alg=D, A=ws.o, W=blk["proj_w"], bias=blk["proj_b"], gamma=blk["ls1"], res=ws.x, ldr=D,
                    flags=EP_BIAS | EP_GAMMA | EP_RESIDUAL, out_f32=ws.x, ldo_f32=D, **a_act(blk, D, ldo, "proj_w"))
            k_layernorm(ws.x, D, T, D, blk["ln2_w"], blk["ln2_b"], LN_EPS, out_op=ws.y, ld_op=ldy, split_seg=seg(blk))
            if w.ffn == "mlp":
                k_igemm(M=T, N=hid, k_alg=D, A=ws.y, W=blk["fc1_w"], bias=blk["fc1_b"], flags=EP_BIAS | EP_GELU,
                        out_op=ws.hd, ldo_op=ldh, split_seg=seg_act(blk, hid), **a_ln(blk, "fc1_w"))
                k_igemm(M=T, N=D, k_alg=hid, A=ws.hd, W=blk["fc2_w"], bias=blk["fc2_b"], gamma=blk["ls2"], res=ws.x, ldr=D,
                        flags=EP_BIAS | EP_GAMMA | EP_RESIDUAL, out_f32=ws.x, ldo_f32=D, **a_act(blk, hid, ldh, "fc2_w"))
            else:
                k_igemm(M=T, N=2 * hid, k_alg=D, A=ws.y, W=blk["w12_w"], bias=blk["w12_b"], flags=EP_BIAS | EP_SWIGLU,
                        out_op=ws.hd, ldo_op=ldh, split_seg=seg_act(blk, hid), **a_ln(blk, "w12_w"))
                k_igemm(M=T, N=D, k_alg=hid, A=ws.hd, W=blk["w3_w"], bias=blk["w3_b"], gamma=blk["ls2"], res=ws.x, ldr=D,
                        flags=EP_BIAS | EP_GAMMA | EP_RESIDUAL, out_f32=ws.x, ldo_f32=D, **a_act(blk, hid, ldh, "w3_w"))
            if self.block_probe is not None:      # diagnostic (tools/stage_errors.py): the residual stream behind block i
                self.block_probe(i, ws)
            if i in taps:  # shared final LayerNorm on the tap, cls row dropped (dinov2.py:337-340)
                tap = ws.taps[taps.index(i)]
                if not last:
                    # the next block's norm1 (block.py:84) reads the very rows this LayerNorm reads: one pass over x, two outputs with the
                    # same statistics -- norm1(x) for all T rows into ws.y, norm(x) without the cls rows into the tap
                    nb = w.blocks[i + 1]
                    k_layernorm(ws.x, D, T, D, nb["ln1_w"], nb["ln1_b"], LN_EPS, out_op=ws.y, ld_op=ldy, split_seg=seg(nb), weight2=w.norm_w, bias2=w.norm_b,
                                out2_op=tap, ld2_op=tap.shape[1], out2_group=N, out2_skip=1, split_seg2=ws.tap_seg)
                    ln1_done = True
                else:
                    k_layernorm(ws.x, D, P, D, w.norm_w, w.norm_b, LN_EPS, group_in=N, skip=1, out_op=tap, ld_op=tap.shape[1],
                                split_seg=ws.tap_seg)
                if w.readout:   # the class token of every image (row b * N of the token matrix): input row stride N * D
                    j = taps.index(i)
                    k_layernorm(ws.x, N * D, B, D, w.norm_w, w.norm_b, LN_EPS, out_op=ws.cls_op[j], ld_op=ws.cls_op[j].shape[1],
                                split_seg=ws.tap_seg)

        return self._head(ws, B) if head else ws

    def _head(self, ws: Workspace, B: int, w: Optional[PackedWeights] = None) -> torch.Tensor:
        w = self.w if w is None else w
        D = w.dim
        ph, pw = ws.ph, ws.pw
        P = B * ph * pw
        oc = w.oc
        Fch = w.features
        grid = ws.grid
        rows = [B * g[0] * g[1] for g in grid]

        ocp, Fp = ws.ocp, ws.Fp
        first = "ip" if w.amodal_head else "rn"     # the family of the contraction that reads the reassembled maps L[i]

        def S(group, seg):   # split_seg argument of a producer whose CONSUMER (a contraction of `group`) reads [hi | lo] segments of width seg
            return (-seg if group in w.f8_groups else seg) if group in w.split else 0     # < 0: [hi | lo8 | hi8]

        taps_in = ws.taps
        if w.readout:   # x = GELU(W_x x + (W_cls cls_b + b))  per image (DA2/dpt.py:164-167)
            KDr = ws.taps[0].shape[1]
            Np = ph * pw
            for i in range(4):
                k_igemm(M=B, N=D, k_alg=D, A=ws.cls_op[i], W=w.ro_wc[i], bias=w.ro_b[i], flags=EP_BIAS, out_f32=ws.cls_bias[i], ldo_f32=D, **self._kdup(KDr, w.ro_wc[i], a_seg=ws.tap_seg))
                # one launch for the whole batch: the bias vector of row m is cls_bias[m // Np] (ada_igemm_args.bias_row_mod)
                k_igemm(M=B * Np, N=D, k_alg=D, A=ws.taps[i], W=w.ro_wx[i], bias=ws.cls_bias[i], bias_row_mod=Np, **self._kdup(KDr, w.ro_wx[i], a_seg=ws.tap_seg),
                        flags=EP_BIAS | EP_GELU, out_op=ws.taps_ro[i], ldo_op=ws.taps_ro[i].shape[1], split_seg=S("proj", D))
            taps_in = ws.taps_ro
        # ---- per level i: reassemble (1x1 project + resize, dpt.py:171-173) -> [amodal: input_projection = conv3x3 -> channels-first LN -> ReLU,
        #      dpt.py:153-159,178-179] -> layerN_rn (blocks.py:20-24): fp32 copy for the residual adds + ReLU'd operand copy for conv1.  The four chains are
        #      independent (chain(i) below touches level-i buffers only) -----
        KD = ws.taps[0].shape[1]

        def reassemble(i):
            if i < 2:
                s_ = 4 if i == 0 else 2
                if i in w.sp:     # 1x1 project -> zero-bordered patch-grid tensor; the transposed conv runs inside the sub-pixel convolution below
                    k_igemm(M=P, N=oc[i], k_alg=D, A=taps_in[i], W=w.proj_w[i], **self._kdup(KD, w.proj_w[i], a_seg=ws.tap_seg), bias=w.proj_b[i], flags=EP_BIAS,
                            out_op=ws.tp[i], ldo_op=ws.tp[i].shape[3], map_op=MAP_PAD, map_h=ph, map_w=pw, split_seg=-ocp[i] if w.sp[i].get("split") else 0)
                    return
                t = ws.t0 if i == 0 else ws.t1
                rs_w, rs_b = (w.rs0_w, w.rs0_b) if i == 0 else (w.rs1_w, w.rs1_b)
                k_igemm(M=P, N=oc[i], k_alg=D, A=taps_in[i], W=w.proj_w[i], **self._kdup(KD, w.proj_w[i], a_seg=ws.tap_seg), bias=w.proj_b[i], flags=EP_BIAS, out_op=t, ldo_op=t.shape[1], split_seg=S(f"rs{i}", ocp[i]))
                k_igemm(M=P, N=s_ * s_ * oc[i], k_alg=oc[i], A=t, W=rs_w, bias=rs_b, flags=EP_BIAS, **self._kdup(t.shape[1], rs_w),
                        out_op=ws.L[i], ldo_op=ws.L[i].shape[3], map_op=MAP_SHUFFLE, map_h=ph, map_w=pw, shuffle_s=s_, shuffle_c=oc[i], split_seg=S(f"{first}{i}", ocp[i]))
            elif i == 2:
                k_igemm(M=P, N=oc[2], k_alg=D, A=taps_in[2], W=w.proj_w[2], **self._kdup(KD, w.proj_w[2], a_seg=ws.tap_seg), bias=w.proj_b[2], flags=EP_BIAS,
                        out_op=ws.L[2], ldo_op=ws.L[2].shape[3], map_op=MAP_PAD, map_h=ph, map_w=pw, split_seg=S(first + "2", ocp[2]))
            else:
                k_igemm(M=P, N=oc[3], k_alg=D, A=taps_in[3], W=w.proj_w[3], **self._kdup(KD, w.proj_w[3], a_seg=ws.tap_seg), bias=w.proj_b[3], flags=EP_BIAS,
                        out_op=ws.pre3, ldo_op=ws.pre3.shape[3], map_op=MAP_PAD, map_h=ph, map_w=pw, split_seg=S("rs3", ocp[3]))
                self._conv3(ws.pre3, w.rs3_w, rows[3], oc[3], grid[3], stride=2, cin=oc[3], bias=w.rs3_b, flags=EP_BIAS,
                            out_op=ws.L[3], ldo_op=ws.L[3].shape[3], map_op=MAP_PAD, map_h=grid[3][0], map_w=grid[3][1], split_seg=S(first + "3", ocp[3]))

        def input_projection(i):      # amodal only
            if i in w.sp:
                # resize_layers[i] + input_projection[i][0] as one sub-pixel convolution over the patch grid: [P, s*s*oc] fp32, column block
                # (py*s + px) = output phase; the LayerNorm reads it in fine-pixel order and takes the transposed conv's bias back out where
                # a tap falls into the zero padding (the outermost ring of the fine grid)
                sp = w.sp[i]
                ncol = sp["s"] * sp["s"] * oc[i]
                self._conv3(ws.tp[i], sp["w"], P, ncol, (ph, pw), k_alg=sp["taps_per_col"] * oc[i], bias=sp["b"], flags=EP_BIAS,
                            out_f32=ws.ipf[i], ldo_f32=ncol, tap_cols=oc[i], tap_mask=sp["masks"])
                k_layernorm(ws.ipf[i], ncol, rows[i], oc[i], w.ip_ln_w[i], w.ip_ln_b[i], LN_EPS, out_op=ws.L2[i], ld_op=ws.L2[i].shape[3],
                            map_op=MAP_PAD, map_h=grid[i][0], map_w=grid[i][1], relu=True, split_seg=S(f"rn{i}", ocp[i]),
                            unshuffle_s=sp["s"], tap_bias=sp["tapb"])
                return
            self._conv3(ws.L[i], w.ip_w[i], rows[i], oc[i], grid[i], cin=oc[i], bias=w.ip_b[i], flags=EP_BIAS, out_f32=ws.ipf[i], ldo_f32=oc[i])
            k_layernorm(ws.ipf[i], oc[i], rows[i], oc[i], w.ip_ln_w[i], w.ip_ln_b[i], LN_EPS, out_op=ws.L2[i],
                        ld_op=ws.L2[i].shape[3], map_op=MAP_PAD, map_h=grid[i][0], map_w=grid[i][1], relu=True, split_seg=S(f"rn{i}", ocp[i]))

        def layer_rn(i):
            if not w.amodal_head and i in w.sp:
                # raw head: resize_layers[i] + layer{i+1}_rn as one sub-pixel convolution over the patch grid, then ONE re-layout pass: fine-pixel
                # order, the transposed conv's bias taken out on the outermost ring, fp32 copy for the residual adds + ReLU'd operand copy
                sp = w.sp[i]
                ncol = sp["s"] * sp["s"] * Fch
                self._conv3(ws.tp[i], sp["w"], P, ncol, (ph, pw), k_alg=sp["taps_per_col"] * oc[i], bias=sp["b"], flags=EP_BIAS,
                            out_f32=ws.spf[i], ldo_f32=ncol, tap_cols=Fch, tap_mask=sp["masks"])
                k_layernorm(ws.spf[i], ncol, rows[i], Fch, None, None, LN_EPS, identity=True, relu=2, out_f32=ws.rnx[i], ld_f32=Fch, out_op=ws.rnr[i],
                            ld_op=ws.rnr[i].shape[3], map_op=MAP_PAD, map_h=grid[i][0], map_w=grid[i][1], split_seg=S(f"rcu{i}", Fp),
                            unshuffle_s=sp["s"], tap_bias=sp["tapb"])
                return
            src = ws.L2[i] if w.amodal_head else ws.L[i]
            self._conv3(src, w.rn_w[i], rows[i], Fch, grid[i], cin=oc[i], flags=EP_RELU_OP, out_f32=ws.rnx[i], ldo_f32=Fch,
                        out_op=ws.rnr[i], ldo_op=ws.rnr[i].shape[3], map_op=MAP_PAD, map_h=grid[i][0], map_w=grid[i][1], split_seg=S(f"rcu{i}", Fp))

        def rcu(i, fw, unit, src_relu_pad, src_f32, **out):
            """ResidualConvUnit (blocks.py:57-80) at grid i: conv2(relu(conv1(relu(x)))) + x."""
            g = grid[i]
            self._conv3(src_relu_pad, fw[f"u{unit}c1_w"], rows[i], Fch, g, cin=Fch, bias=fw[f"u{unit}c1_b"], flags=EP_BIAS | EP_RELU_OP,
                        out_op=ws.tmpa[i], ldo_op=ws.tmpa[i].shape[3], map_op=MAP_PAD, map_h=g[0], map_w=g[1], split_seg=S(f"rcu{i}", Fp))
            self._conv3(ws.tmpa[i], fw[f"u{unit}c2_w"], rows[i], Fch, g, cin=Fch, bias=fw[f"u{unit}c2_b"], res=src_f32, ldr=Fch,
                        flags=EP_BIAS | EP_RESIDUAL, **out)

        def chain(i):     # everything of level i that does not depend on another level; RCU 1 of levels 0-2 (blocks.py:131-133) reads rnr / rnx of its own level only
            reassemble(i)
            if w.amodal_head:
                input_projection(i)
            layer_rn(i)
            if i < 3:
                rcu(i, w.fuse[i], 1, ws.rnr[i], ws.rnx[i], out_f32=ws.r[i], ldo_f32=Fch)

        # (r[i] lives in ipf[i] where that is large enough and zf[i] in rnx[i] -- Workspace: both are dead by the time they are overwritten INSIDE chain i / after the join)
        forked = HEAD_STREAMS == "1" or (HEAD_STREAMS == "auto" and P <= HEAD_STREAMS_ROWS)
        if forked and not instrumented():
            cur = torch.cuda.current_stream(ws.oc1.device)
            side = self._side_streams(ws.oc1.device)
            for st_ in side:
                st_.wait_stream(cur)
            chain(0)                      # the finest level -- most of the work -- stays on the caller's stream
            for i, st_ in zip((1, 2, 3), side):
                with torch.cuda.stream(st_):
                    chain(i)
            for st_ in side:
                cur.wait_stream(st_)
        else:
            for i in range(4):
                chain(i)

        # ---- refinenet4..1 (blocks.py:123-148).  out_conv is applied BEFORE the bilinear resize: both are linear and
        #      the align_corners weights sum to one, so conv1x1(resize(x)) == resize(conv1x1(x)) at a quarter of the MACs.
        s_f32, s_pad = ws.rnx[3], ws.rnr[3]
        for i in (3, 2, 1, 0):
            fw = w.fuse[i]
            rcu(i, fw, 2, s_pad, s_f32, out_op=ws.u[i], ldo_op=ws.u[i].shape[1], split_seg=S(f"out{i}", Fp))
            if i == 0 and w.oc1c is not None:
                break     # refinenet1.out_conv is part of output_conv1's tap maps (below)
            k_igemm(M=rows[i], N=Fch, k_alg=Fch, A=ws.u[i], W=fw["out_w"], bias=fw["out_b"], flags=EP_BIAS, **self._kdup(ws.u[i].shape[1], fw["out_w"]),
                    out_f32=ws.zf[i], ldo_f32=Fch)
            if i > 0:
                j = i - 1
                k_bilinear(ws.zf[i], Fch, B, grid[i][0], grid[i][1], grid[j][0], grid[j][1], Fch, add=ws.r[j], ld_add=Fch,
                           out_f32=ws.s[j], ld_f32=Fch, out_op=ws.sr[j], ld_op=ws.sr[j].shape[3], map_op=MAP_PAD, relu=True, split_seg=S(f"rcu{j}", Fp))
                s_f32, s_pad = ws.s[j], ws.sr[j]
        g2 = ws.g296
        # ---- resize x2 -> output_conv1 (dpt.py:192-193) -> resize to (14 ph, 14 pw) -> output_conv2 (3x3, ReLU, 1x1, activation) (:194-195) ----
        if w.oc1c is not None:
            ntap = 9 * ws.half
            if "f8" in w.oc1c:      # u[0] is [hi | lo8 | hi8] (split group "out0"): a full split product with fp8 correction terms
                k_igemm(M=rows[0], N=ntap, K=2 * Fp, k_alg=Fch, A=ws.u[0], lda=ws.u[0].shape[1], W=w.oc1c["w"], f8_from=Fp, f8_mid=Fp + Fp // 2, f8_scales=w.oc1c["f8"],
                        bias=w.oc1c["b"], flags=EP_BIAS, out_op=ws.tmaps, ldo_op=ntap)
            else:
                k_igemm(M=rows[0], N=ntap, K=2 * Fp, a_wrap=Fp, k_alg=Fch, A=ws.u[0], lda=Fp, W=w.oc1c["w"], bias=w.oc1c["b"], flags=EP_BIAS, out_op=ws.tmaps, ldo_op=ntap)
            k_tapsum_resize(ws.tmaps, ntap, B, grid[0][0], grid[0][1], g2[0], g2[1], ws.half, w.oc1_b, ws.oc1, ws.half)
        else:
            k_bilinear(ws.zf[0], Fch, B, grid[0][0], grid[0][1], g2[0], g2[1], Fch, out_op=ws.p1, ld_op=ws.p1.shape[3], map_op=MAP_PAD, split_seg=S("oc1", Fp))
            self._conv3(ws.p1, w.oc1_w, B * g2[0] * g2[1], ws.half, g2, cin=Fch, bias=w.oc1_b, flags=EP_BIAS, out_f32=ws.oc1, ldo_f32=ws.half)
        out = torch.empty(B, 1, ws.H, ws.W, dtype=torch.float32, device=ws.oc1.device)
        if ws.fused_tail:
            k_dpt_tail(ws.oc1, ws.half, B, g2[0], g2[1], ws.H, ws.W, ws.halfp, w.oc2_w, w.oc2_b, w.tail_w, w.tail_b, self.final_act, out)
            return out
        k_bilinear(ws.oc1, ws.half, B, g2[0], g2[1], ws.H, ws.W, ws.half, out_op=ws.fin, ld_op=ws.fin.shape[3], map_op=MAP_PAD, split_seg=S("oc2", ws.halfp))
        self._conv3(ws.fin, w.oc2_w, B * ws.H * ws.W, w.oc2_w.shape[0], (ws.H, ws.W), cin=ws.half, bias=w.oc2_b, flags=EP_BIAS | EP_TAIL,
                    out_f32=out, ldo_f32=1, tail_w=w.tail_w, tail_b=w.tail_b, tail_act=self.final_act)
        return out
