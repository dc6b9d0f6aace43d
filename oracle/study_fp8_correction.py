"""TEST INFRASTRUCTURE / numerics study (CPU, not imported by the product): can the CORRECTION terms of a split-precision product run on the fp8 matrix
pipe?  Split precision evaluates x w ~ x_hi w_hi + x_lo w_hi + x_hi w_lo with hi = fp16(v), lo = fp16(v - hi): three fp16 products (two for a weight-only
split).  The two correction products are ~2^-11 of the main one, so they only need a few bits: here they are evaluated with BOTH factors rounded to
fp8 e4m3 (3 mantissa bits; gfx950's v_mfma_f32_16x16x128_f8f6f4 runs at twice the fp16 rate) after a power-of-two scale per row (activations) /
per output channel (weights) -- 2x the fp16 MACs of a single-precision product instead of 3x, 1.5x instead of 2x for the weight-only form.

    python oracle/study_fp8_correction.py raw_vitg_224_w1 [fixture ...]

Modes, every contraction of the forward in the same mode, relative L1 against the fp32 forward:
    fp16        x_hi w_hi                                               (single precision: the default of the sigmoid models)
    split       x_hi w_hi + x_lo w_hi + x_hi w_lo in fp16               (what the product runs today where the policy asks for it)
    fp8corr     x_hi w_hi in fp16 + e4m3(x_lo) e4m3(w_hi) + e4m3(x_hi) e4m3(w_lo)
    bf8corr     the same with e5m2 (2 mantissa bits)
    fp8corr-pt  e4m3 with ONE scale per tensor instead of per row / channel
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "amodal-depth-anything_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import dav2_oracle as O  # noqa: E402
from _cases import case_inputs, load_golden, schema_state_dict  # noqa: E402


def _pow2_scale(t, dim, top):
    """power-of-two scale s (per slice along `dim`, or per tensor for dim=None) with max|t * s| <= top"""
    a = t.abs().amax() if dim is None else t.abs().amax(dim=dim, keepdim=True)
    a = a.clamp_min(1e-30)
    return torch.exp2(torch.floor(torch.log2(top / a)))


def _q8(t, dt, dim):
    top = 448.0 if dt == torch.float8_e4m3fn else 57344.0
    s = _pow2_scale(t, dim, top)
    return (t * s).to(dt).to(torch.float32) / s


class Corr(O._Numerics):
    """x w = x_hi w_hi (fp16) + corrections in `mode`."""

    def __init__(self, mode):
        super().__init__(torch.float16)
        self.mode = mode

    def _parts(self, t, dims):
        hi = t.to(torch.float16).to(torch.float32)
        lo = t - hi
        if self.mode == "fp16":
            return hi, None, None
        if self.mode == "split":
            return hi, lo.to(torch.float16).to(torch.float32), hi
        dt = torch.float8_e5m2 if self.mode.startswith("bf8") else torch.float8_e4m3fn
        dim = None if self.mode.endswith("-pt") else dims
        return hi, _q8(lo, dt, dim), _q8(hi, dt, dim)

    def _apply(self, fn, x, w, xdims, wdims):
        xh, xl, xh8 = self._parts(x, xdims)
        wh, wl, wh8 = self._parts(w, wdims)
        y = fn(xh, wh, True)
        if xl is not None:
            y = y + fn(xl, wh8 if self.mode != "split" else wh, False) + fn(xh8 if self.mode != "split" else xh, wl, False)
        return y

    def linear(self, x, w, b=None):
        return self._apply(lambda a, c, bias: F.linear(a, c, b if bias else None), x, w, -1, 1)

    def conv(self, x, w, b=None, stride=1, padding=0):
        # activations: one scale per pixel (the GEMM row of an implicit-GEMM conv) ; weights: per output channel
        return self._apply(lambda a, c, bias: F.conv2d(a, c, b if bias else None, stride=stride, padding=padding), x, w, 1, (1, 2, 3))

    def convT(self, x, w, b, stride):
        return self._apply(lambda a, c, bias: F.conv_transpose2d(a, c, b if bias else None, stride=stride), x, w, 1, (0, 2, 3))

    def matmul(self, a, b):      # attention matmuls: fp16 single (negligible share)
        return self.q(a) @ self.q(b)


def forward(case, sd, inputs, nm):
    x, grgb, mask, obs = inputs
    real = O._Numerics
    O._Numerics = lambda dt=None: nm
    try:
        if case["kind"] == "raw":
            return O.raw_forward(sd, case["encoder"], x)
        return O.amodal_forward(sd, case["encoder"], case["guide_type"], case["loss"], x, grgb, mask, obs)
    finally:
        O._Numerics = real


def main():
    for name in (sys.argv[1:] or ["raw_vitg_224_w1"]):
        _, meta = load_golden(name)
        case = meta["case"]
        sd = schema_state_dict(case, meta)
        inputs = case_inputs(case)
        with torch.no_grad():
            ref = forward(case, sd, inputs, O._Numerics(torch.float32))
            line = f"{name}:"
            for mode in ("fp16", "split", "fp8corr", "bf8corr", "fp8corr-pt"):
                out = forward(case, sd, inputs, Corr(mode))
                line += f"  {mode} {O.rel_l1(out, ref):.3e}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
