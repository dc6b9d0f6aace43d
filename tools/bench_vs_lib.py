#!/usr/bin/env python
"""Yardstick, not product: the vendor libraries (hipBLASLt through torch.matmul / F.linear, torch's SDPA) timed on the same box at the
ViT-L bs=32 shapes, next to this build's igemm / attention kernels on the same random operands.  Nothing here is on the
product path; it only tells how far the hand-written kernels sit from what the library reaches under the same clocks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import hip_ext as H  # noqa: E402
from hip_ext.engine import Q_PRESCALE  # noqa: E402

dev = "cuda"
op = H.operand_dtype()
torch.manual_seed(0)
reps = int(os.environ.get("REPS", "10"))


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


T = 43840
print(f"operand dtype {op}; {reps} reps each")
for name, M, N, K in [("qkv", T, 3072, 1024), ("proj", T, 1024, 1024), ("fc1", T, 4096, 1024), ("fc2", T, 1024, 4096),
                      ("square", 8192, 8192, 8192)]:
    A = torch.randn(M, K, device=dev).to(op)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(op)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=op, device=dev)
    ms_lib = timeit(lambda: torch.matmul(A, W.t(), out=out))
    biash = bias.to(op)
    ms_lin = timeit(lambda: F.linear(A, W, biash))
    ms_ada = timeit(lambda: H.igemm(M=M, N=N, K=K, A=A, lda=K, W=W, bias=bias, flags=H.EP_BIAS, out_op=out, ldo_op=N))
    fl = 2.0 * M * N * K / 1e9
    print(f"  gemm {name:6s} M={M} N={N} K={K}: matmul {ms_lib * 1e3:7.1f} us {fl / ms_lib:7.1f} TF | F.linear+bias {ms_lin * 1e3:7.1f} us "
          f"{fl / ms_lin:7.1f} TF | ada_igemm+bias {ms_ada * 1e3:7.1f} us {fl / ms_ada:7.1f} TF")

B, N, h, d = 32, 1370, 16, 64
qkv = torch.randn(B * N, 3 * h * d, device=dev)
qkv[:, :h * d] *= Q_PRESCALE
qkv = qkv.to(op)
o = torch.empty(B * N, h * d, dtype=op, device=dev)
q, k, v = (t.reshape(B, N, h, d).transpose(1, 2).contiguous() for t in torch.randn(B * N, 3 * h * d, device=dev).to(op).chunk(3, dim=1))
fl = 4.0 * B * h * N * N * d / 1e9
ms_ada = timeit(lambda: H.attention(qkv, o, B, N, h))
print(f"  attention B={B} N={N} h={h}: ada_attention_fwd {ms_ada * 1e3:7.1f} us {fl / ms_ada:7.1f} TF")
for backend in ("FLASH_ATTENTION", "EFFICIENT_ATTENTION", "MATH"):
    try:
        from torch.nn.attention import SDPBackend, sdpa_kernel
        with sdpa_kernel(getattr(SDPBackend, backend)):
            ms = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
        print(f"     torch SDPA {backend:20s}: {ms * 1e3:7.1f} us {fl / ms:7.1f} TF (inputs already [B,h,N,d] contiguous)")
    except Exception as e:  # backend not built into this torch
        print(f"     torch SDPA {backend:20s}: unavailable ({str(e).splitlines()[0][:90]})")
