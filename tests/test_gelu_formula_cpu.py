"""CPU: the constants of gelu_erf() in csrc/ada_igemm.hip, evaluated in fp32 numpy exactly as the kernel does (clamp, Horner fma chain,
exp2, final fma), reproduce the exact-erf GELU of the reference (mlp.py:23, nn.GELU()) to fp32-roundoff class everywhere."""
import os
import re

import numpy as np
import torch

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "amodal-depth-anything_amd", "csrc", "ada_igemm.hip")


def _kernel_constants():
    body = open(SRC).read()
    body = body[body.index("ADA_DEV float gelu_erf(float x)"):]
    body = body[:body.index("\n}\n")]
    clamp = float(re.search(r"__builtin_fminf\(__builtin_fabsf\(x\), ([0-9.eE+-]+)f\)", body).group(1))
    lead = float(re.search(r"float q = ([0-9.eE+-]+)f;", body).group(1))
    rest = [float(v) for v in re.findall(r"__builtin_fmaf\(q, a, ([0-9.eE+-]+)f\)", body)]
    return clamp, [lead] + rest


def test_gelu_formula_matches_exact_erf_gelu():
    clamp, coef = _kernel_constants()
    assert len(coef) == 7 and clamp == 12.0
    x = np.concatenate([np.linspace(-40, 40, 1600001), np.array([0.0, -0.0, 1e-30, -1e-30, 65504.0, -65504.0])]).astype(np.float32)
    a = np.minimum(np.abs(x), np.float32(clamp))
    q = np.full_like(a, np.float32(coef[0]))
    for c in coef[1:]:
        q = (q * a + np.float32(c)).astype(np.float32)
    got = (np.maximum(x, 0) - a * np.exp2(-q).astype(np.float32)).astype(np.float32)
    want = torch.nn.functional.gelu(torch.from_numpy(x).double()).numpy()
    err = np.abs(got - want)
    assert np.isfinite(got).all()
    assert float(err[np.abs(x) <= 12].max()) <= 4e-7, float(err[np.abs(x) <= 12].max())
    assert float((err / np.maximum(np.abs(want), 1.0)).max()) <= 4e-7      # |x| > 12: identity / zero to fp32 relative roundoff
