#!/usr/bin/env python
"""Column-group width sweep (ada_debug_set_group) for the encoder GEMM shapes at ViT-L bs=32, bias-only and fused epilogues: is the fc1 gap to
the library (1136 vs 955 TF/s at equal epilogue, profiles/r03_a_vs_vendor_libs.txt) a tile-order / L2-residency effect?  The 8 MB fc1 weight
matrix does not fit one XCD's 4 MiB L2; the group width decides how many 512 KB weight slabs a run of tiles keeps resident.
    python tools/group_sweep.py [--reps 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hip_ext as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rows", type=int, default=43840)
args = ap.parse_args()
op = H.operand_dtype()
dev = "cuda"
torch.manual_seed(0)
T = args.rows
shapes = [("qkv bias", T, 3072, 1024, H.EP_BIAS, "op"), ("fc1 bias", T, 4096, 1024, H.EP_BIAS, "op"), ("fc1 bias+gelu", T, 4096, 1024, H.EP_BIAS | H.EP_GELU, "op"),
          ("fc2 bias", T, 1024, 4096, H.EP_BIAS, "op"), ("fc2 ls+res", T, 1024, 4096, H.EP_BIAS | H.EP_GAMMA | H.EP_RESIDUAL, "res"),
          ("proj bias", T, 1024, 1024, H.EP_BIAS, "op"), ("proj ls+res", T, 1024, 1024, H.EP_BIAS | H.EP_GAMMA | H.EP_RESIDUAL, "res")]
print(f"rows = {T}; columns: group width (0 = the launch heuristic); cells: us per launch (TFLOP/s)")
for name, M, N, K, flags, mode in shapes:
    A = torch.randn(M, K, device=dev).to(op)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(op)
    bias, gamma = torch.randn(N, device=dev), torch.rand(N, device=dev)
    kw = dict(M=M, N=N, K=K, A=A, lda=K, W=W, bias=bias, flags=flags)
    if mode == "op":
        out = torch.empty(M, N, dtype=op, device=dev)
        kw.update(out_op=out, ldo_op=N)
    else:
        x = torch.randn(M, N, device=dev)
        kw.update(gamma=gamma, res=x, ldr=N, out_f32=x, ldo_f32=N)
    cells = []
    for g in (0, 1, 2, 3, 4, 6, 8, 16):
        if g > N // 256:
            continue
        H.debug_set_group(g)
        for _ in range(3):
            H.igemm(**kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            H.igemm(**kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        cells.append(f"g={g}: {ms * 1e3:6.1f} ({2.0 * M * N * K / ms / 1e9:5.0f})")
    H.debug_set_group(0)
    print(f"{name:14s} " + "  ".join(cells), flush=True)
