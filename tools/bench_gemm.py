#!/usr/bin/env python
"""Isolated igemm timing at the ViT-L bs=32 GEMM shapes (random operands).  Env ADA_IGEMM_SCHED / ADA_IGEMM_TILE select
kernel variants (read once per process), so A/B runs are separate processes: see tools/ab_gemm.sh."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hip_ext as H  # noqa: E402

op = H.operand_dtype()
dev = "cuda"
torch.manual_seed(0)
T = 43840
shapes = [
    ("qkv   bias->op", T, 3072, 1024, dict(flags=H.EP_BIAS), "op"),
    ("proj  ls+res f32", T, 1024, 1024, dict(flags=H.EP_BIAS | H.EP_GAMMA | H.EP_RESIDUAL), "f32res"),
    ("fc1   gelu->op", T, 4096, 1024, dict(flags=H.EP_BIAS | H.EP_GELU), "op"),
    ("fc2   ls+res f32", T, 1024, 4096, dict(flags=H.EP_BIAS | H.EP_GAMMA | H.EP_RESIDUAL), "f32res"),
    ("plain K=8192 f32", 8192, 8192, 8192, dict(flags=0), "f32"),
]
reps = int(os.environ.get("REPS", "10"))
only = os.environ.get("ONLY")
if only:
    shapes = [s for s in shapes if only in s[0]]
print(f"SCHED={os.environ.get('ADA_IGEMM_SCHED', '0')} TILE={os.environ.get('ADA_IGEMM_TILE', '-')}")
for name, M, N, K, kw, mode in shapes:
    A = (torch.randn(M, K, device=dev) * 1.0).to(op)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(op)
    if os.environ.get("ZERO"):
        A.zero_(); W.zero_()
    bias = torch.randn(N, device=dev)
    gamma = torch.rand(N, device=dev)
    args = dict(M=M, N=N, K=K, A=A, lda=K, W=W, bias=bias, **kw)
    if mode == "op":
        out = torch.empty(M, N, dtype=op, device=dev)
        args.update(out_op=out, ldo_op=N)
    elif mode == "f32res":
        x = torch.randn(M, N, device=dev)
        args.update(gamma=gamma, res=x, ldr=N, out_f32=x, ldo_f32=N)
    else:
        out = torch.empty(M, N, device=dev)
        args.update(out_f32=out, ldo_f32=N)
        args.pop("bias")
    for _ in range(2):
        H.igemm(**args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        H.igemm(**args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"  {name:18s} M={M} N={N} K={K}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s")
