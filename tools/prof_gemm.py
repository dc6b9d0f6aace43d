#!/usr/bin/env python
"""Interval anatomy of the phased 256x256 GEMM main loop (ada_debug_set_timestamps)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hip_ext as H  # noqa: E402

op = H.operand_dtype()
T = 43840
for name, M, N, K in (("qkv", T, 3072, 1024), ("fc2-like f16 out", T, 1024, 4096)):
    A = torch.randn(M, K, device="cuda").to(op)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(op)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, dtype=op, device="cuda")
    nblk = ((M + 255) // 256) * ((N + 255) // 256)
    buf = torch.zeros(nblk * 2 * 8, dtype=torch.int64, device="cuda")
    for _ in range(2):
        H.igemm(M=M, N=N, K=K, A=A, lda=K, W=W, bias=bias, flags=H.EP_BIAS, out_op=out, ldo_op=N)
    H.load().ada_debug_set_timestamps(buf.data_ptr())
    H.igemm(M=M, N=N, K=K, A=A, lda=K, W=W, bias=bias, flags=H.EP_BIAS, out_op=out, ldo_op=N)
    torch.cuda.synchronize()
    H.load().ada_debug_set_timestamps(None)
    t = buf.cpu().reshape(nblk, 2, 8).double()
    nk = t[0, 0, 3].item()
    print(f"{name}: M={M} N={N} K={K}  ({nk:.0f} k-tiles, 4 phases each); s_memtime ticks per PHASE, mean over {nblk} workgroups")
    for g in (0, 1):
        lb, m, b2 = (t[:, g, k].mean().item() / (4 * nk) for k in range(3))
        print(f"   group {g}:  L + barrier + read wait {lb:7.1f}   16 MFMAs {m:7.1f}   barrier {b2:7.1f}   sum {lb + m + b2:7.1f}  (k-tile {4 * (lb + m + b2):7.1f})")
