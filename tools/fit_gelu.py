#!/usr/bin/env python
"""Fit behind gelu_erf() in csrc/ada_igemm.hip:  gelu(x) = max(x, 0) - |x| * Phi(-|x|),  Phi(-a) = exp2(-q(a)).
q(a) = -log2(0.5 * erfc(a / sqrt 2)) is fitted by a polynomial on a in [0, 6], iteratively re-weighted towards the minimax error of the
product a * exp2(-q) (the GELU error), then checked in fp32 Horner arithmetic over x in [-12, 12].  Prints degree, max |error|, coefficients."""
import numpy as np
from scipy.special import erf, erfc


def fit(n, amax=6.0):
    a = (np.cos(np.linspace(0, np.pi, 3000)) * 0.5 + 0.5) * amax
    f = -np.log2(0.5 * erfc(a / np.sqrt(2)))
    A = np.stack([a ** k for k in range(n + 1)], 1)
    w0 = np.maximum(a * 0.5 * erfc(a / np.sqrt(2)), 1e-4)     # d gelu = a * tail * ln2 * dq
    w = w0.copy()
    for _ in range(200):
        coef = np.linalg.lstsq(A * w[:, None], f * w, rcond=None)[0]
        err = np.abs(A @ coef - f) * w0
        w = w * (1 + 2 * err / err.max())
        w /= w.mean()
    return coef


def check(coef):
    x = np.linspace(-12, 12, 2000001).astype(np.float32)
    a = np.minimum(np.abs(x), np.float32(12))
    q = np.zeros_like(a)
    for c in coef[::-1].astype(np.float32):
        q = (q * a + c).astype(np.float32)
    g = (np.maximum(x, 0) - a * np.exp2(-q).astype(np.float32)).astype(np.float32)
    xd = x.astype(np.float64)
    return float(np.abs(g - 0.5 * xd * (1 + erf(xd / np.sqrt(2)))).max())


if __name__ == "__main__":
    for n in (4, 5, 6, 7):
        c = fit(n)
        print(n, f"{check(c):.3e}", " ".join(f"{v:.10e}" for v in c))
