"""Stand-alone functional forms of the reference's L1 operators on top of libada_hip.

These back the ``forward`` methods of the individual nn.Modules (Attention, Mlp, PatchEmbed, DPTHead ...), so each
piece of the reference's module surface works on its own, computing in the same HIP kernels as the fused engine.
They take/return the reference's tensor layouts (fp32, [B,N,D] tokens, NCHW feature maps) and therefore pay for
layout changes and operand packing on every call -- torch is used for that plumbing (reshape / permute / pad / dtype
cast) only; every contraction, normalisation, softmax and resample runs in a HIP kernel.  The whole-model path
(``hip_ext.engine``) packs weights once and keeps activations in kernel-native layouts instead.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import (A_CONV3, ACT_NONE, ACT_RELU, ACT_SIGMOID, EP_BIAS, EP_GELU, EP_RELU_OP, EP_RESIDUAL, EP_SWIGLU, EP_TAIL,
               MAP_PAD, MAP_SHUFFLE, HipExtError)
from . import attention as k_attention
from . import bilinear as k_bilinear
from . import igemm as k_igemm
from . import layernorm as k_layernorm
from . import operand_dtype
from . import patchify as k_patchify


def _r64(c):
    return (c + 63) // 64 * 64


def _need_cuda(t, name):
    if not t.is_cuda:
        raise HipExtError(f"{name}: expected a tensor on a HIP device (the HIP path has no CPU fallback)")


def _as_operand_rows(x2d):
    """[M, K] (fp32 or operand type) -> contiguous operand-typed [M, r64(K)]."""
    op = operand_dtype()
    M, K = x2d.shape
    if x2d.dtype == op and K % 64 == 0 and x2d.is_contiguous():
        return x2d
    out = torch.zeros(M, _r64(K), dtype=op, device=x2d.device)
    out[:, :K] = x2d
    return out


def _pack_rows(w2d):
    op = operand_dtype()
    N, K = w2d.shape
    out = torch.zeros(N, _r64(K), dtype=op, device=w2d.device)
    out[:, :K] = w2d
    return out


def linear(x, weight, bias=None, gelu=False, out_operand=False):
    """nn.Linear (+ optional exact GELU).  x: [..., K] fp32 or operand-typed; returns fp32 unless out_operand."""
    _need_cuda(x, "linear input")
    lead, K = x.shape[:-1], x.shape[-1]
    N = weight.shape[0]
    A = _as_operand_rows(x.reshape(-1, K))
    W = _pack_rows(weight.detach().float())
    M = A.shape[0]
    flags = (EP_BIAS if bias is not None else 0) | (EP_GELU if gelu else 0)
    b = None if bias is None else bias.detach().float().contiguous()
    if N % 4:
        raise HipExtError("linear: out_features must be a multiple of 4")
    if out_operand:
        out = torch.empty(M, N, dtype=operand_dtype(), device=x.device)
        k_igemm(M=M, N=N, K=A.shape[1], A=A, lda=A.shape[1], W=W, bias=b, flags=flags, out_op=out, ldo_op=N)
    else:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
        k_igemm(M=M, N=N, K=A.shape[1], A=A, lda=A.shape[1], W=W, bias=b, flags=flags, out_f32=out, ldo_f32=N)
    return out.reshape(*lead, N)


def swiglu_linear(x, w12, b12):
    """SwiGLU first half: silu(x1) * x2 with [x1, x2] = w12(x) (reference swiglu_ffn.py:30-32); operand-typed result."""
    _need_cuda(x, "swiglu input")
    lead, K = x.shape[:-1], x.shape[-1]
    hid = w12.shape[0] // 2
    idx = torch.arange(hid, device=w12.device).reshape(-1, 32)
    order = torch.stack([idx, idx + hid], dim=1).reshape(-1)
    A = _as_operand_rows(x.reshape(-1, K))
    W = _pack_rows(w12.detach().float()[order])
    b = None if b12 is None else b12.detach().float()[order].contiguous()
    M = A.shape[0]
    out = torch.empty(M, hid, dtype=operand_dtype(), device=x.device)
    k_igemm(M=M, N=2 * hid, K=A.shape[1], A=A, lda=A.shape[1], W=W, bias=b, flags=(EP_BIAS if b is not None else 0) | EP_SWIGLU,
            out_op=out, ldo_op=hid)
    return out.reshape(*lead, hid)


def layer_norm(x, weight, bias, eps):
    """LayerNorm over the last dim, fp32 in / fp32 out."""
    _need_cuda(x, "layer_norm input")
    D = x.shape[-1]
    x2 = x.reshape(-1, D).float().contiguous()
    out = torch.empty_like(x2)
    k_layernorm(x2, D, x2.shape[0], D, weight.detach().float().contiguous(), bias.detach().float().contiguous(), float(eps),
                out_f32=out, ld_f32=D)
    return out.reshape(x.shape)


def scale_channels(x, gamma):
    """LayerScale: x * gamma (a broadcast multiply; inside the engine it is fused into the GEMM epilogue)."""
    return x * gamma


def self_attention(x, qkv_w, qkv_b, proj_w, proj_b, num_heads):
    """Attention.forward (reference attention.py:49-62): qkv linear -> fused softmax(q k^T / sqrt(d)) v -> proj."""
    _need_cuda(x, "attention input")
    B, N, C = x.shape
    if C // num_heads != 64:
        raise HipExtError("self_attention: the fused kernel is built for head_dim 64 (every DINOv2 size)")
    from .engine import Q_PRESCALE
    w = qkv_w.detach().float().clone()
    w[:C] *= Q_PRESCALE
    b = None
    if qkv_b is not None:
        b = qkv_b.detach().float().clone()
        b[:C] *= Q_PRESCALE
    qkv = linear(x, w, b, out_operand=True).reshape(B * N, 3 * C)
    o = torch.empty(B * N, C, dtype=operand_dtype(), device=x.device)
    k_attention(qkv, o, B, N, num_heads)
    return linear(o.reshape(B, N, C), proj_w, proj_b)


def patch_embed(x, weight, bias):
    """PatchEmbed conv (k = stride = 14) as patchify + GEMM; x: [B,C,H,W] fp32 -> [B, Np, D] fp32."""
    _need_cuda(x, "patch_embed input")
    B, C, H, W = x.shape
    if weight.shape[-1] != 14 or weight.shape[-2] != 14:
        raise HipExtError("patch_embed: the patchify kernel is built for 14x14 patches")
    D = weight.shape[0]
    x = x.float().contiguous()
    K = C * 196
    ld = _r64(K)
    P = B * (H // 14) * (W // 14)
    A = torch.empty(P, ld, dtype=operand_dtype(), device=x.device)
    # the kernel's first three channels are "rgb", the rest "guide": split any channel count accordingly
    if C >= 3:
        k_patchify(x[:, :3].contiguous(), x[:, 3:].contiguous() if C > 3 else None, B, C - 3, H, W, None, None, A, ld)
    else:
        xp = torch.cat([x, torch.zeros(B, 3 - C, H, W, device=x.device)], 1)
        A3 = torch.empty(P, _r64(588), dtype=operand_dtype(), device=x.device)
        k_patchify(xp, None, B, 0, H, W, None, None, A3, A3.shape[1])
        A.zero_()
        A[:, :K] = A3[:, :K]
    out = torch.empty(P, D, dtype=torch.float32, device=x.device)
    k_igemm(M=P, N=D, K=ld, A=A, lda=ld, W=_pack_rows(weight.detach().float().reshape(D, K)),
            bias=None if bias is None else bias.detach().float().contiguous(), flags=EP_BIAS if bias is not None else 0,
            out_f32=out, ldo_f32=D)
    return out.reshape(B, -1, D)


def _to_padded_nhwc(x):
    B, C, H, W = x.shape
    cp = _r64(C)
    buf = torch.zeros(B, H + 2, W + 2, cp, dtype=operand_dtype(), device=x.device)
    buf[:, 1:-1, 1:-1, :C] = x.permute(0, 2, 3, 1)
    return buf


def _pack_conv3(w):
    co, ci = w.shape[:2]
    cp = _r64(ci)
    p = torch.zeros(co, 3, 3, cp, dtype=operand_dtype(), device=w.device)
    p[..., :ci] = w.permute(0, 2, 3, 1)
    return p.reshape(co, 9 * cp)


def conv2d(x, weight, bias=None, stride=1, padding=0, relu_input=False):
    """nn.Conv2d for the two shapes the DPT head uses: 1x1/s1/p0 and 3x3/(s1|s2)/p1.  NCHW fp32 in/out."""
    _need_cuda(x, "conv2d input")
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    if relu_input:
        x = torch.relu(x)
    b = None if bias is None else bias.detach().float().contiguous()
    flags = EP_BIAS if b is not None else 0
    if Co % 4:
        raise HipExtError("conv2d: out_channels must be a multiple of 4 (use conv_tail for the 32->1 head)")
    if (kh, kw) == (1, 1) and stride == 1 and padding == 0:
        A = _as_operand_rows(x.permute(0, 2, 3, 1).reshape(-1, C))
        out = torch.empty(A.shape[0], Co, dtype=torch.float32, device=x.device)
        k_igemm(M=A.shape[0], N=Co, K=A.shape[1], A=A, lda=A.shape[1], W=_pack_rows(weight.detach().float().reshape(Co, C)), bias=b,
                flags=flags, out_f32=out, ldo_f32=Co)
        return out.reshape(B, H, W, Co).permute(0, 3, 1, 2).contiguous()
    if (kh, kw) == (3, 3) and padding == 1 and stride in (1, 2):
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        xin = _to_padded_nhwc(x.float())
        cp = xin.shape[3]
        out = torch.empty(B * Ho * Wo, Co, dtype=torch.float32, device=x.device)
        k_igemm(M=B * Ho * Wo, N=Co, K=9 * cp, A=xin, lda=cp, W=_pack_conv3(weight.detach().float()), a_mode=A_CONV3,
                conv=(Ho, Wo, H + 2, W + 2, stride), bias=b, flags=flags, out_f32=out, ldo_f32=Co)
        return out.reshape(B, Ho, Wo, Co).permute(0, 3, 1, 2).contiguous()
    raise HipExtError(f"conv2d: unsupported geometry k={kh}x{kw} stride={stride} padding={padding}")


def conv_transpose2d(x, weight, bias, stride):
    """nn.ConvTranspose2d with kernel == stride (non-overlapping): GEMM + pixel shuffle.  NCHW fp32 in/out."""
    _need_cuda(x, "conv_transpose2d input")
    B, Ci, H, W = x.shape
    Co, s = weight.shape[1], int(stride)
    if weight.shape[2] != s or weight.shape[3] != s or Co % 8:
        raise HipExtError("conv_transpose2d: kernel must equal stride and out_channels be a multiple of 8")
    A = _as_operand_rows(x.permute(0, 2, 3, 1).reshape(-1, Ci))
    Wt = _pack_rows(weight.detach().float().permute(2, 3, 1, 0).reshape(s * s * Co, Ci))
    out = torch.zeros(B, s * H + 2, s * W + 2, Co, dtype=operand_dtype(), device=x.device)
    bb = torch.zeros(Co, device=x.device) if bias is None else bias.detach().float()
    k_igemm(M=B * H * W, N=s * s * Co, K=A.shape[1], A=A, lda=A.shape[1], W=Wt, bias=bb.repeat(s * s).contiguous(), flags=EP_BIAS,
            out_op=out, ldo_op=Co, map_op=MAP_SHUFFLE, map_h=H, map_w=W, shuffle_s=s, shuffle_c=Co)
    return out[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float().contiguous()


def fold_batchnorm(w, b, bn_weight, bn_bias, running_mean, running_var, eps=1e-5):
    """Inference BatchNorm2d behind a conv (reference blocks.py:70-76 with bn=True) as that conv's weight and bias:
    bn(conv(x)) = conv(x) * s + (bn_bias - running_mean * s), s = bn_weight / sqrt(running_var + eps)  -- a parameter transform, done once."""
    s = bn_weight.detach().float() / torch.sqrt(running_var.detach().float() + eps)
    w2 = w.detach().float() * s.view(-1, 1, 1, 1)
    b0 = b.detach().float() if b is not None else torch.zeros_like(s)
    return w2, (b0 - running_mean.detach().float()) * s + bn_bias.detach().float()


def subpixel_merge(wt, bt, w3, b3, s):
    """Composes ``conv3x3(conv_transpose2d(x, wt, bt, stride=s), w3, b3, padding=1)`` -- nothing in between, reference
    DA2/dpt.py:88-100,173 (resize_layers[0/1]) followed by :153-159,178-179 (input_projection[i][0]) or util/blocks.py:20-24 (layerN_rn) --
    into ONE 3x3 convolution over the COARSE grid whose s*s*Co output columns are the s x s output phases of every coarse pixel
    (a sub-pixel convolution).  wt [Ci, Cm, s, s], bt [Cm] or None, w3 [Co, Cm, 3, 3], b3 [Co] or None, all fp32; composed in fp64.

    Fine pixel (s y + py, s x + px) reads the fine pixels (s y + py + ty - 1, s x + px + tx - 1), ty, tx in 0..2; with v = py + ty - 1, that
    pixel belongs to coarse row y + dy, dy = floor(v / s), at inner phase v - s dy.  Hence

        Wm[(py, px, co), (dy, dx), ci] = sum over (ty -> dy, tx -> dx) of  W3[co, :, ty, tx] @ Wt[ci, :, qy, qx]^T

    which is non-zero for 1, 2 or 4 of the 9 coarse taps only: 36 Ci Co MACs per coarse pixel instead of 16 Ci Cm + 144 Cm Co for s = 4.
    Returns (Wm [s*s*Co, 9, Ci], bias [s*s*Co] for interior pixels, tap_bias [s*s*Co, 9], masks [s*s]): tap_bias[n, t] is the part of the
    bias that arrives through coarse tap t (the transposed conv's bias seen through the 3x3 filter) -- where tap t falls into the zero
    padding the fine pixels it stands for do not exist and that part must be left out (ada_layernorm_ex tap_bias); masks[p] has bit t set
    iff phase p touches tap t (ada_igemm_args.tap_mask)."""
    Ci, Cm = wt.shape[:2]
    Co = w3.shape[0]
    dev = wt.device
    # every product W3[:, :, ty, tx] @ Wt[:, :, qy, qx]^T in ONE launch: rows (ty, tx, co), columns (qy, qx, ci) -- compose_f32 (ada_igemm in split
    # precision: ~1e-7 relative, far below the fp16 rounding the packed result gets); the sums over the fine taps that share a coarse tap are adds
    P = compose_f32(w3.permute(2, 3, 0, 1).reshape(9 * Co, Cm), wt.permute(2, 3, 0, 1).reshape(s * s * Ci, Cm)).view(3, 3, Co, s, s, Ci)
    Wm = torch.zeros(s, s, Co, 9, Ci, dtype=torch.float32, device=dev)
    tb = torch.zeros(s, s, Co, 9, dtype=torch.float32, device=dev)
    masks = [0] * (s * s)
    tbias = None if bt is None else (w3.double() * bt.double().view(1, Cm, 1, 1)).sum(1).float()     # [Co, 3, 3]: W3[:, :, ty, tx] @ bt (a reduction, not a GEMM)
    for py in range(s):
        for px in range(s):
            for ty in range(3):
                for tx in range(3):
                    vy, vx = py + ty - 1, px + tx - 1
                    dy, dx = vy // s, vx // s              # floor: -1 -> -1, s -> 1
                    qy, qx = vy - s * dy, vx - s * dx
                    t = (dy + 1) * 3 + (dx + 1)
                    Wm[py, px, :, t, :] += P[ty, tx, :, qy, qx, :]
                    if tbias is not None:
                        tb[py, px, :, t] += tbias[:, ty, tx]
                    masks[py * s + px] |= 1 << t
    bias = tb.sum(-1)
    if b3 is not None:
        bias = bias + b3.float().view(1, 1, Co)
    return (Wm.reshape(s * s * Co, 9, Ci).contiguous(), bias.reshape(-1).contiguous(), tb.reshape(s * s * Co, 9).contiguous(), masks)


def compose_f32(a, b):
    """a [M, K] @ b [N, K]^T -> [M, N] fp32 for WEIGHT composition at pack time (sub-pixel merges, output_conv1 o out_conv), on the library's own GEMM:
    both factors are carried as hi + lo operand pairs -- a as [hi | lo] rows, b packed [hi | hi | lo] -- and one ada_igemm over the three k segments
    evaluates a_hi b_hi + a_lo b_hi + a_hi b_lo with fp32 accumulation (error ~ 2^-22 |a||b| per term; the result is rounded to the operand type, or
    to a hi / lo pair, right afterwards).  No vendor BLAS anywhere in the package, pack time included (round 4 composed with torch fp64 matmuls)."""
    _need_cuda(a, "compose_f32 input")
    op = operand_dtype()
    M, K = a.shape
    N = b.shape[0]
    kp, np_ = _r64(K), (N + 3) // 4 * 4
    a32 = F.pad(a.detach().float(), (0, kp - K))
    b32 = F.pad(b.detach().float(), (0, kp - K, 0, np_ - N))
    a_hi = a32.to(op)
    b_hi = b32.to(op)
    A = torch.cat([a_hi, (a32 - a_hi.float()).to(op)], dim=1).contiguous()
    Wp = torch.cat([b_hi, b_hi, (b32 - b_hi.float()).to(op)], dim=1).contiguous()
    out = torch.empty(M, np_, dtype=torch.float32, device=a.device)
    k_igemm(M=M, N=np_, K=3 * kp, A=A, lda=2 * kp, a_dup_seg=kp, W=Wp, flags=0, out_f32=out, ldo_f32=np_)
    return out[:, :N] if np_ != N else out


def residual_conv_unit(x, w1, b1, w2, b2):
    """ResidualConvUnit (reference blocks.py:57-80) with ReLU / residual fused into the conv epilogues."""
    _need_cuda(x, "residual_conv_unit input")
    B, C, H, W = x.shape
    x = x.float()
    cp = _r64(C)
    xr = _to_padded_nhwc(torch.relu(x))
    mid = torch.zeros(B, H + 2, W + 2, cp, dtype=operand_dtype(), device=x.device)
    geom = (H, W, H + 2, W + 2, 1)
    k_igemm(M=B * H * W, N=C, K=9 * cp, A=xr, lda=cp, W=_pack_conv3(w1.detach().float()), a_mode=A_CONV3, conv=geom,
            bias=b1.detach().float().contiguous(), flags=EP_BIAS | EP_RELU_OP, out_op=mid, ldo_op=cp, map_op=MAP_PAD, map_h=H, map_w=W)
    res = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous()
    out = torch.empty_like(res)
    k_igemm(M=B * H * W, N=C, K=9 * cp, A=mid, lda=cp, W=_pack_conv3(w2.detach().float()), a_mode=A_CONV3, conv=geom,
            bias=b2.detach().float().contiguous(), res=res, ldr=C, flags=EP_BIAS | EP_RESIDUAL, out_f32=out, ldo_f32=C)
    return out.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()


def interpolate_bilinear_ac(x, size):
    """F.interpolate(mode='bilinear', align_corners=True) on NCHW fp32."""
    _need_cuda(x, "interpolate input")
    B, C, H, W = x.shape
    ho, wo = int(size[0]), int(size[1])
    if C % 4:
        raise HipExtError("interpolate_bilinear_ac: channels must be a multiple of 4")
    src = x.float().permute(0, 2, 3, 1).reshape(-1, C).contiguous()
    out = torch.empty(B * ho * wo, C, dtype=torch.float32, device=x.device)
    k_bilinear(src, C, B, H, W, ho, wo, C, out_f32=out, ld_f32=C)
    return out.reshape(B, ho, wo, C).permute(0, 3, 1, 2).contiguous()


def conv_tail(x, w0, b0, w2, b2, final_act):
    """output_conv2: conv3x3(C->32) + ReLU + conv1x1(32->1) + Sigmoid/ReLU/identity in one launch
    (reference DA2/dpt.py:146-151, RAW dpt.py:109-115).  NCHW fp32 in, [B,1,H,W] fp32 out."""
    _need_cuda(x, "conv_tail input")
    B, C, H, W = x.shape
    xin = _to_padded_nhwc(x.float())
    cp = xin.shape[3]
    out = torch.empty(B, 1, H, W, dtype=torch.float32, device=x.device)
    act = {"sigmoid": ACT_SIGMOID, "relu": ACT_RELU, "none": ACT_NONE}[final_act]
    k_igemm(M=B * H * W, N=w0.shape[0], K=9 * cp, A=xin, lda=cp, W=_pack_conv3(w0.detach().float()), a_mode=A_CONV3,
            conv=(H, W, H + 2, W + 2, 1), bias=b0.detach().float().contiguous(), flags=EP_BIAS | EP_TAIL, out_f32=out, ldo_f32=1,
            tail_w=w2.detach().float().reshape(-1).contiguous(), tail_b=float(b2.detach().float().reshape(-1)[0].item()), tail_act=act)
    return out
