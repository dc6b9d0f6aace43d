"""hip_ext -- ctypes binding of libada_hip.so (the C ABI declared in include/ada_hip.h).

PyTorch is used here for device memory and streams only: every wrapper hands raw device pointers,
sizes and the current HIP stream to the library.  There is NO fallback: if the shared library is
missing or a tensor is not on a HIP device the call raises -- the product path never computes on
the CPU (the fp32 CPU oracle under /oracle is test infrastructure and is never imported here).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint16, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(os.path.dirname(_HERE), "csrc")

# --- constants mirrored from include/ada_hip.h ------------------------------------------------
ABI_VERSION = 8
DT_F32, DT_F16, DT_BF16 = 0, 1, 2
A_PLAIN, A_CONV3 = 0, 1
MAP_PLAIN, MAP_PAD, MAP_TOKEN, MAP_SHUFFLE = 0, 1, 2, 3
EP_BIAS, EP_GELU, EP_GAMMA, EP_RESIDUAL, EP_RELU_OP, EP_SWIGLU, EP_TAIL, EP_RELU_F32 = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40, 0x80
ACT_NONE, ACT_SIGMOID, ACT_RELU = 0, 1, 2

EXPORTS = (
    "ada_abi_version", "ada_operand_dtype", "ada_last_error", "ada_igemm", "ada_attention_fwd", "ada_attention_ex",
    "ada_pos_embed_resize", "ada_layernorm_fwd", "ada_layernorm_ex", "ada_patchify", "ada_write_cls", "ada_bilinear_fwd", "ada_selftest",
    "ada_minmax_fwd", "ada_depth_stats_fwd", "ada_token_diversity_fwd", "ada_normalize_fwd", "ada_blend_fwd", "ada_depth_eval_fwd", "ada_tile_blend_fwd", "ada_dpt_tail_fwd", "ada_tapsum_resize_fwd",
    "ada_debug_set_tile", "ada_debug_set_variant", "ada_debug_set_group", "ada_debug_last_tile",
    "ada_debug_set_timestamps", "ada_debug_set_attention_variant", "ada_debug_count_saturated",
)

# indices into the per-image sums of ada_depth_eval_fwd (ADA_EVAL_* in include/ada_hip.h)
EVAL_N, EVAL_SUM_P, EVAL_SUM_G, EVAL_SUM_PP, EVAL_SUM_PG, EVAL_ABS_REL, EVAL_SQ_REL, EVAL_SQ, EVAL_LOG_SQ, EVAL_LOG, \
    EVAL_LOG10_ABS, EVAL_D1, EVAL_D2, EVAL_D3, EVAL_INV_SQ = range(15)
EVAL_NSUM = 16


class IgemmArgs(ctypes.Structure):
    """struct ada_igemm_args (include/ada_hip.h) -- field order and types must match exactly."""
    _fields_ = [
        ("M", c_int32), ("N", c_int32), ("K", c_int32), ("a_mode", c_int32),
        ("A", c_void_p), ("lda", c_int64),
        ("Ho", c_int32), ("Wo", c_int32), ("Hp", c_int32), ("Wp", c_int32), ("stride", c_int32),
        ("W", c_void_p), ("bias", c_void_p), ("gamma", c_void_p), ("res", c_void_p), ("ldr", c_int64),
        ("res_row_mod", c_int32), ("res_row_off", c_int32), ("flags", c_int32),
        ("out_f32", c_void_p), ("ldo_f32", c_int64), ("map_f32", c_int32),
        ("out_op", c_void_p), ("ldo_op", c_int64), ("map_op", c_int32),
        ("map_h", c_int32), ("map_w", c_int32), ("shuffle_s", c_int32), ("shuffle_c", c_int32),
        ("tail_w", c_void_p), ("tail_b", c_float), ("tail_act", c_int32),
        ("split_seg", c_int32), ("a_dup_seg", c_int32),
        ("tap_cols", c_int32), ("tap_mask", c_uint16 * 16), ("a_wrap", c_int32), ("bias_row_mod", c_int32),
        ("f8_from", c_int32), ("f8_mid", c_int32), ("f8_scales", ctypes.c_uint32),
    ]


class LayerNormArgs(ctypes.Structure):
    """struct ada_layernorm_args (include/ada_hip.h)."""
    _fields_ = [
        ("in_", c_void_p), ("ld_in", c_int64), ("rows_out", c_int32), ("dim", c_int32), ("group_in", c_int32), ("skip", c_int32),
        ("weight", c_void_p), ("bias", c_void_p), ("eps", c_float),
        ("out_op", c_void_p), ("ld_op", c_int64), ("map_op", c_int32), ("map_h", c_int32), ("map_w", c_int32), ("relu", c_int32),
        ("out_f32", c_void_p), ("ld_f32", c_int64), ("split_seg", c_int32),
        ("weight2", c_void_p), ("bias2", c_void_p), ("out2_op", c_void_p), ("ld2_op", c_int64),
        ("out2_group", c_int32), ("out2_skip", c_int32), ("split_seg2", c_int32), ("unshuffle_s", c_int32), ("tap_bias", c_void_p), ("identity", c_int32),
    ]


class HipExtError(RuntimeError):
    pass


_lib = None
_lib_path = None


def library_path(bf16: bool = False) -> str:
    return os.path.join(_CSRC, "libada_hip_bf16.so" if bf16 else "libada_hip.so")


def load(path: Optional[str] = None):
    """Loads (once) and returns the ctypes handle.  Raises HipExtError when the library is absent."""
    global _lib, _lib_path
    if _lib is not None and (path is None or path == _lib_path):
        return _lib
    path = path or os.environ.get("ADA_HIP_LIB") or library_path()
    if not os.path.exists(path):
        raise HipExtError(
            f"libada_hip.so not found at {path}: build it with `python {os.path.join(_CSRC, 'build.py')}` "
            "(there is no CPU fallback for the HIP path)")
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise HipExtError(f"cannot load {path}: {e}") from e
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise HipExtError(f"{path} does not export {name}")
    lib.ada_abi_version.restype = c_int
    lib.ada_operand_dtype.restype = c_int
    lib.ada_last_error.restype = c_char_p
    lib.ada_igemm.argtypes = [ctypes.POINTER(IgemmArgs), c_void_p]
    lib.ada_igemm.restype = c_int
    lib.ada_pos_embed_resize.argtypes = [c_void_p, c_int32, c_int32, c_int32, c_int32, c_double, c_double, c_void_p, c_void_p]
    lib.ada_pos_embed_resize.restype = c_int
    lib.ada_attention_fwd.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]
    lib.ada_attention_fwd.restype = c_int
    lib.ada_attention_ex.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_int32, c_void_p]
    lib.ada_attention_ex.restype = c_int
    lib.ada_layernorm_fwd.argtypes = [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_float,
                                      c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_int32, c_void_p]
    lib.ada_layernorm_fwd.restype = c_int
    lib.ada_layernorm_ex.argtypes = [ctypes.POINTER(LayerNormArgs), c_void_p]
    lib.ada_layernorm_ex.restype = c_int
    lib.ada_patchify.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                 ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_void_p, c_int64, c_int32, c_void_p]
    lib.ada_patchify.restype = c_int
    lib.ada_write_cls.argtypes = [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]
    lib.ada_write_cls.restype = c_int
    lib.ada_bilinear_fwd.argtypes = [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64,
                                     c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]
    lib.ada_bilinear_fwd.restype = c_int
    lib.ada_selftest.argtypes = [c_void_p, c_int64, c_void_p]
    lib.ada_selftest.restype = c_int
    lib.ada_minmax_fwd.argtypes = [c_void_p, c_int32, c_int64, c_void_p, c_void_p]
    lib.ada_minmax_fwd.restype = c_int
    lib.ada_depth_stats_fwd.argtypes = [c_void_p, c_int32, c_int64, c_int32, c_int32, c_void_p, c_void_p]
    lib.ada_depth_stats_fwd.restype = c_int
    lib.ada_token_diversity_fwd.argtypes = [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p]
    lib.ada_token_diversity_fwd.restype = c_int
    lib.ada_normalize_fwd.argtypes = [c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p]
    lib.ada_normalize_fwd.restype = c_int
    lib.ada_blend_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]
    lib.ada_blend_fwd.restype = c_int
    lib.ada_tile_blend_fwd.argtypes = [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]
    lib.ada_tile_blend_fwd.restype = c_int
    lib.ada_dpt_tail_fwd.argtypes = [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                     c_float, c_int32, c_void_p, c_void_p]
    lib.ada_dpt_tail_fwd.restype = c_int
    lib.ada_tapsum_resize_fwd.argtypes = [c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p]
    lib.ada_tapsum_resize_fwd.restype = c_int
    lib.ada_depth_eval_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_float, c_float, c_void_p, c_void_p]
    lib.ada_depth_eval_fwd.restype = c_int
    for name in ("ada_debug_set_tile", "ada_debug_set_variant", "ada_debug_set_group", "ada_debug_set_attention_variant"):
        getattr(lib, name).argtypes = [c_int]
        getattr(lib, name).restype = None
    for name in ("ada_debug_set_timestamps",):
        getattr(lib, name).argtypes = [c_void_p]
        getattr(lib, name).restype = None
    lib.ada_debug_count_saturated.argtypes = [c_void_p, c_int64, c_void_p, c_void_p]
    lib.ada_debug_count_saturated.restype = c_int
    lib.ada_debug_last_tile.argtypes = []
    lib.ada_debug_last_tile.restype = c_int
    if lib.ada_abi_version() != ABI_VERSION:
        raise HipExtError(f"{path}: ABI version {lib.ada_abi_version()} != binding version {ABI_VERSION}")
    _lib, _lib_path = lib, path
    return lib


def operand_dtype() -> torch.dtype:
    """torch dtype of the contraction operands the loaded library was built for."""
    return torch.float16 if load().ada_operand_dtype() == DT_F16 else torch.bfloat16


def _check(rc: int, what: str):
    if rc != 0:
        msg = load().ada_last_error().decode(errors="replace")
        raise HipExtError(f"{what} failed (rc={rc}): {msg}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str, dtype=None) -> int:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HipExtError(f"{name}: expected a tensor on a HIP device (the HIP path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise HipExtError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], name: str, dtype=None):
    return None if t is None else _dev(t, name, dtype)


class KernelTimer:
    """Optional per-kernel timing with HIP events recorded on the stream the kernels are launched on
    (bench.py uses it to compute the roofline fraction of the dominant kernels inside the timed region)."""

    def __init__(self):
        self.records = {}
        self.active = True     # bench.py brackets the launches of every 4th timed step only: the event records cost ~2 us of stream time each

    def start(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def stop(self, name, ev0, work):
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record(torch.cuda.current_stream())
        self.records.setdefault(name, []).append((ev0, ev1, float(work)))

    def summary(self):
        """name -> dict(launches, ms_total, ms_avg, work_total)  (call after a device synchronize)"""
        out = {}
        for name, recs in self.records.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in recs)
            out[name] = dict(launches=len(recs), ms_total=ms, ms_avg=ms / len(recs), work_total=sum(w for _, _, w in recs))
        return out


_timer: Optional[KernelTimer] = None


def set_timer(t: Optional[KernelTimer]):
    global _timer
    _timer = t


# ------------------------------------------------------------------------------------------------
# thin wrappers (argument meaning = include/ada_hip.h)
# ------------------------------------------------------------------------------------------------
def igemm(*, M, N, K, A, lda, W, k_alg=None, a_mode=A_PLAIN, conv=None, bias=None, gamma=None, res=None, ldr=0,
          res_row_mod=0, res_row_off=0, flags=0, out_f32=None, ldo_f32=0, map_f32=MAP_PLAIN, out_op=None, ldo_op=0,
          map_op=MAP_PLAIN, map_h=0, map_w=0, shuffle_s=0, shuffle_c=0, tail_w=None, tail_b=0.0, tail_act=ACT_NONE, split_seg=0,
          a_dup_seg=0, tap_cols=0, tap_mask=None, a_wrap=0, bias_row_mod=0,
          f8_from=0, f8_mid=0, f8_scales=0):
    op = operand_dtype()
    a = IgemmArgs()
    a.M, a.N, a.K, a.a_mode = M, N, K, a_mode
    a.A, a.lda = _dev(A, "A", op), lda
    if conv is not None:
        a.Ho, a.Wo, a.Hp, a.Wp, a.stride = conv
    a.W = _dev(W, "W", op)
    a.bias = _opt(bias, "bias", torch.float32)
    a.gamma = _opt(gamma, "gamma", torch.float32)
    a.res, a.ldr = _opt(res, "res", torch.float32), ldr
    a.res_row_mod, a.res_row_off, a.flags = res_row_mod, res_row_off, flags
    a.out_f32, a.ldo_f32, a.map_f32 = _opt(out_f32, "out_f32", torch.float32), ldo_f32, map_f32
    a.out_op, a.ldo_op, a.map_op = _opt(out_op, "out_op", op), ldo_op, map_op
    a.map_h, a.map_w, a.shuffle_s, a.shuffle_c = map_h, map_w, shuffle_s, shuffle_c
    a.tail_w, a.tail_b, a.tail_act = _opt(tail_w, "tail_w", torch.float32), tail_b, tail_act
    a.split_seg = split_seg
    a.a_dup_seg = a_dup_seg
    a.a_wrap = a_wrap
    a.bias_row_mod = bias_row_mod
    a.f8_from, a.f8_mid, a.f8_scales = f8_from, f8_mid, f8_scales
    if tap_cols:
        a.tap_cols = tap_cols
        for i, m in enumerate(tap_mask):
            a.tap_mask[i] = int(m)
    if _timer is not None and _timer.active:
        ev = _timer.start()
        _check(load().ada_igemm(ctypes.byref(a), _stream()), "ada_igemm")
        _timer.stop("igemm", ev, 2.0 * M * N * (k_alg if k_alg is not None else K))  # algorithmic FLOP (MAC = 2)
    else:
        _check(load().ada_igemm(ctypes.byref(a), _stream()), "ada_igemm")
    if _tile_log is not None:
        _tile_log.append((M, N, K, load().ada_debug_last_tile()))


def attention(qkv: torch.Tensor, out: torch.Tensor, batch: int, n_tokens: int, heads: int, ld_out: int = 0, split_seg: int = 0):
    """ada_attention_fwd / ada_attention_ex: ``ld_out`` = row stride of ``out`` (0: heads * 64), ``split_seg`` != 0: the split-precision forms of the output row."""
    op = operand_dtype()
    ev = _timer.start() if (_timer is not None and _timer.active) else None
    if ld_out or split_seg:
        _check(load().ada_attention_ex(_dev(qkv, "qkv", op), _dev(out, "out", op), batch, n_tokens, heads, ld_out, split_seg, _stream()), "ada_attention_ex")
    else:
        _check(load().ada_attention_fwd(_dev(qkv, "qkv", op), _dev(out, "out", op), batch, n_tokens, heads, _stream()),
               "ada_attention_fwd")
    if ev is not None:
        _timer.stop("attention", ev, 4.0 * batch * heads * 64 * float(n_tokens) ** 2)  # QK^T + PV, MAC = 2


def layernorm(inp, ld_in, rows_out, dim, weight, bias, eps, *, group_in=0, skip=0, out_op=None, ld_op=0, map_op=MAP_PLAIN,
              map_h=0, map_w=0, relu=False, out_f32=None, ld_f32=0, split_seg=0, weight2=None, bias2=None, out2_op=None, ld2_op=0,
              out2_group=0, out2_skip=0, split_seg2=0, unshuffle_s=0, tap_bias=None, identity=False):
    """ada_layernorm_ex (include/ada_hip.h): LayerNorm rows -> op-typed / fp32 output; optionally a second op-typed output with its own gain /
    bias (out2_*), optionally reading a sub-pixel convolution's [coarse pixel, s*s*dim] output in fine-pixel order (unshuffle_s, tap_bias)."""
    op = operand_dtype()
    a = LayerNormArgs()
    a.in_, a.ld_in, a.rows_out, a.dim, a.group_in, a.skip = _dev(inp, "in", torch.float32), ld_in, rows_out, dim, group_in, skip
    a.weight, a.bias, a.eps = _opt(weight, "weight", torch.float32), _opt(bias, "bias", torch.float32), eps
    a.identity = int(identity)
    a.out_op, a.ld_op, a.map_op, a.map_h, a.map_w, a.relu = _opt(out_op, "out_op", op), ld_op, map_op, map_h, map_w, int(relu)
    a.out_f32, a.ld_f32, a.split_seg = _opt(out_f32, "out_f32", torch.float32), ld_f32, split_seg
    a.weight2, a.bias2 = _opt(weight2, "weight2", torch.float32), _opt(bias2, "bias2", torch.float32)
    a.out2_op, a.ld2_op, a.out2_group, a.out2_skip, a.split_seg2 = _opt(out2_op, "out2_op", op), ld2_op, out2_group, out2_skip, split_seg2
    a.unshuffle_s, a.tap_bias = unshuffle_s, _opt(tap_bias, "tap_bias", torch.float32)
    _check(load().ada_layernorm_ex(ctypes.byref(a), _stream()), "ada_layernorm_ex")


def patchify(x, guide, batch, cg, height, width, mean, inv_std, out, ld, split=False):
    op = operand_dtype()
    if mean is not None:
        m = (c_float * 3)(*mean)
        s = (c_float * 3)(*inv_std)
    else:
        m = s = None
    _check(load().ada_patchify(_dev(x, "x", torch.float32), _opt(guide, "guide", torch.float32), batch, cg, height, width,
                               m, s, _dev(out, "out", op), ld, int(split), _stream()), "ada_patchify")


def pos_embed_resize(pos, sq, dim, ph, pw, scale_h, scale_w, out):
    """pos fp32 [1 + sq*sq, dim] -> out fp32 [1 + ph*pw, dim] (ada_pos_embed_resize: bicubic, ATen semantics)."""
    _check(load().ada_pos_embed_resize(_dev(pos, "pos", torch.float32), sq, dim, ph, pw, float(scale_h), float(scale_w),
                                       _dev(out, "out", torch.float32), _stream()), "ada_pos_embed_resize")


def write_cls(tokens, batch, n_tokens, dim, cls, pos0):
    _check(load().ada_write_cls(_dev(tokens, "tokens", torch.float32), batch, n_tokens, dim,
                                _dev(cls, "cls", torch.float32), _dev(pos0, "pos0", torch.float32), _stream()), "ada_write_cls")


def bilinear(inp, ld_in, batch, hi, wi, ho, wo, channels, *, add=None, ld_add=0, out_f32=None, ld_f32=0, out_op=None,
             ld_op=0, map_op=MAP_PLAIN, relu=False, split_seg=0):
    op = operand_dtype()
    _check(load().ada_bilinear_fwd(_dev(inp, "in", torch.float32), ld_in, batch, hi, wi, ho, wo, channels,
                                   _opt(add, "add", torch.float32), ld_add, _opt(out_f32, "out_f32", torch.float32), ld_f32,
                                   _opt(out_op, "out_op", op), ld_op, map_op, int(relu), split_seg, _stream()), "ada_bilinear_fwd")


def dpt_tail(inp, ld_in, batch, hi, wi, ho, wo, cp, w, bias, tail_w, tail_b, tail_act, out):
    """Fused bilinear resize + output_conv2 (3x3 -> ReLU -> 1x1 -> activation), ada_dpt_tail_fwd."""
    op = operand_dtype()
    ev = _timer.start() if (_timer is not None and _timer.active) else None
    _check(load().ada_dpt_tail_fwd(_dev(inp, "in", torch.float32), ld_in, batch, hi, wi, ho, wo, cp, _dev(w, "w", op),
                                   _dev(bias, "bias", torch.float32), _dev(tail_w, "tail_w", torch.float32), tail_b, tail_act,
                                   _dev(out, "out", torch.float32), _stream()), "ada_dpt_tail_fwd")
    if ev is not None:
        _timer.stop("dpt_tail", ev, 2.0 * batch * ho * wo * 32 * 9 * cp)


def tapsum_resize(inp, ld_in, batch, hi, wi, ho, wo, channels, bias, out, ld_out):
    """conv3x3 of an align-corners up-sampling from the nine coarse tap maps (ada_tapsum_resize_fwd)."""
    ev = _timer.start() if (_timer is not None and _timer.active) else None
    if inp.dtype not in (torch.float32, operand_dtype()):
        raise HipExtError(f"tapsum_resize: tap maps must be fp32 or {operand_dtype()}, got {inp.dtype}")
    code = DT_F32 if inp.dtype == torch.float32 else (DT_F16 if inp.dtype == torch.float16 else DT_BF16)
    _check(load().ada_tapsum_resize_fwd(_dev(inp, "in"), code, ld_in, batch, hi, wi, ho, wo, channels, _opt(bias, "bias", torch.float32),
                                        _dev(out, "out", torch.float32), ld_out, _stream()), "ada_tapsum_resize_fwd")
    if ev is not None:
        _timer.stop("tapsum_resize", ev, 2.0 * 36 * batch * ho * wo * channels)


def minmax(inp, minmax_out):
    """inp: fp32 [B, ...] contiguous -> minmax_out fp32 [B, 2]."""
    B = inp.shape[0]
    _check(load().ada_minmax_fwd(_dev(inp, "in", torch.float32), B, inp.numel() // B, _dev(minmax_out, "minmax", torch.float32), _stream()),
           "ada_minmax_fwd")


def depth_stats(inp, sums, act=ACT_SIGMOID):
    """inp: fp32 [B, ...] contiguous depth maps of a head that ends in `act` -> sums fp32 [B, chunks, 2], per chunk (sum s, sum s (1 - s)) for a sigmoid,
    (sum out, number of positive outputs) for a ReLU, (sum |out|, number of outputs) for none (ada_depth_stats_fwd)."""
    B = inp.shape[0]
    _check(load().ada_depth_stats_fwd(_dev(inp, "in", torch.float32), B, inp.numel() // B, sums.shape[1], int(act), _dev(sums, "sums", torch.float32), _stream()),
           "ada_depth_stats_fwd")


def token_diversity(tap, ld, batch, rows_per_image, dim, sums):
    """tap: operand-typed [batch * rows_per_image, ld] -> sums fp32 [batch, ceil(dim / 64), 2] = per column chunk (sum Var_rows, sum E_rows[t^2]) (ada_token_diversity_fwd)."""
    _check(load().ada_token_diversity_fwd(_dev(tap, "tap", operand_dtype()), ld, batch, rows_per_image, dim, _dev(sums, "sums", torch.float32), _stream()),
           "ada_token_diversity_fwd")


def normalize(inp, minmax_in, norm=None, obs=None):
    B = inp.shape[0]
    _check(load().ada_normalize_fwd(_dev(inp, "in", torch.float32), _dev(minmax_in, "minmax", torch.float32), B, inp.numel() // B,
                                    _opt(norm, "norm", torch.float32), _opt(obs, "obs", torch.float32), _stream()), "ada_normalize_fwd")


def blend(amodal, base, mask, out):
    B, H, W = amodal.shape[0], amodal.shape[-2], amodal.shape[-1]
    _check(load().ada_blend_fwd(_dev(amodal, "amodal", torch.float32), _dev(base, "base", torch.float32), _dev(mask, "mask", torch.float32),
                                B, H, W, _dev(out, "out", torch.float32), _stream()), "ada_blend_fwd")


def tile_blend(tiles, origin_y, origin_x, height, width, ramp, out):
    """tiles fp32 [B, T, th, tw]; origin_y / origin_x int32 [T] on the device; out fp32 [B, height, width] (ada_tile_blend_fwd)."""
    B, T, th, tw = tiles.shape
    _check(load().ada_tile_blend_fwd(_dev(tiles, "tiles", torch.float32), B, T, th, tw, _dev(origin_y, "origin_y", torch.int32),
                                     _dev(origin_x, "origin_x", torch.int32), height, width, ramp, _dev(out, "out", torch.float32), _stream()),
           "ada_tile_blend_fwd")


def depth_eval(pred, gt, mask=None, scale_shift=None, clip=None) -> torch.Tensor:
    """Per-image masked evaluation sums, fp64 [B, EVAL_NSUM] (ada_depth_eval_fwd).  pred / gt: fp32 [B, ...]; mask: uint8/bool
    of the same shape or None; scale_shift: fp32 [B, 2] or None; clip: (lo, hi) or None."""
    B = pred.shape[0]
    n = pred[0].numel()
    if gt.shape != pred.shape or (mask is not None and mask.shape != pred.shape):
        raise HipExtError(f"depth_eval: shape mismatch pred {tuple(pred.shape)} gt {tuple(gt.shape)}")
    if mask is not None and mask.dtype == torch.bool:
        mask = mask.view(torch.uint8) if mask.is_contiguous() else mask.to(torch.uint8)
    sums = torch.empty(B, EVAL_NSUM, dtype=torch.float64, device=pred.device)
    lo, hi = (float(clip[0]), float(clip[1])) if clip is not None else (0.0, 0.0)
    _check(load().ada_depth_eval_fwd(_dev(pred, "pred", torch.float32), _dev(gt, "gt", torch.float32),
                                     _dev(mask, "mask", torch.uint8) if mask is not None else None, B, n,
                                     _dev(scale_shift, "scale_shift", torch.float32) if scale_shift is not None else None,
                                     lo, hi, _dev(sums, "sums", torch.float64), _stream()), "ada_depth_eval_fwd")
    return sums


# --- tuning / diagnostic hooks (include/ada_hip.h, last section) ---------------------------------
TILE_NAMES = {0: "256x32", 1: "128x64", 2: "256x128", 3: "256x256", 4: "128x128"}


_tile_log = None
# bumped by every debug_set_* call: captured HIP graphs (hip_ext/engine.py) bake the kernel variant in, so they are keyed by this epoch
_debug_epoch = 0


def debug_epoch() -> int:
    return _debug_epoch


def instrumented() -> bool:
    """True while a KernelTimer or a tile log is attached: launches must then go through the Python wrappers (no graph replay)."""
    return _timer is not None or _tile_log is not None


def _bump_epoch():
    global _debug_epoch
    _debug_epoch += 1


def set_tile_log(log):
    """log: a list that receives (M, N, K, tile code) for every ada_igemm launch, or None to stop recording."""
    global _tile_log
    _tile_log = log


def debug_set_tile(cfg: int = -1):
    _bump_epoch()
    load().ada_debug_set_tile(int(cfg))


def debug_set_variant(v: int = 0):
    _bump_epoch()
    load().ada_debug_set_variant(int(v))


def debug_set_group(g: int = 0):
    _bump_epoch()
    load().ada_debug_set_group(int(g))


def debug_last_tile() -> int:
    return int(load().ada_debug_last_tile())


def debug_set_attention_variant(v: int = 5):
    _bump_epoch()
    load().ada_debug_set_attention_variant(int(v))


def count_saturated(buf: torch.Tensor, counter: torch.Tensor):
    """Adds to ``counter`` (int64 [1] on the device) the number of elements of the operand-typed tensor ``buf`` that sit at the fp16 clamp
    (+-65504) or are inf / NaN (ada_debug_count_saturated)."""
    if not buf.is_contiguous():
        raise HipExtError("count_saturated: contiguous tensor required")
    _check(load().ada_debug_count_saturated(_dev(buf, "buf", operand_dtype()), buf.numel(), _dev(counter, "counter", torch.int64), _stream()),
           "ada_debug_count_saturated")


def selftest() -> int:
    scratch = torch.zeros(1 << 18, dtype=torch.int32, device="cuda")
    return load().ada_selftest(scratch.data_ptr(), scratch.numel() * 4, _stream())
