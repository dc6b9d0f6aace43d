#!/usr/bin/env python
"""Isolated timing of ada_attention_fwd at the ViT-L bs=32 shape (random q/k/v)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hip_ext as H  # noqa: E402

op = H.operand_dtype()
B, N, heads = int(os.environ.get("B", 32)), int(os.environ.get("N", 1370)), 16
D = heads * 64
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device="cuda")
qkv[:, :D] *= 0.125 * 1.4426950408889634
qkv = qkv.to(op)
if os.environ.get("ZERO"):
    qkv.zero_()
out = torch.empty(B * N, D, dtype=op, device="cuda")
reps = int(os.environ.get("REPS", 10))
H.debug_set_attention_variant(int(os.environ.get("VARIANT", 5)))
for _ in range(2):
    H.attention(qkv, out, B, N, heads)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    H.attention(qkv, out, B, N, heads)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"variant={os.environ.get('VARIANT', 0)} attention B={B} N={N} heads={heads}: {ms * 1e3:.1f} us  {4.0 * B * heads * 64 * N * N / ms / 1e9:.1f} TFLOP/s")
