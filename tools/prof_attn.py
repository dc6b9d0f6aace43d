#!/usr/bin/env python
"""Interval anatomy of the ping-pong attention kernel (variant 2: s_memtime stamps per wave)."""
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
out = torch.empty(B * N, D, dtype=op, device="cuda")
nblk = ((N + 127) // 128) * B * heads
buf = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device="cuda")
H.load().ada_debug_set_attention_profile(buf.data_ptr())
H.debug_set_attention_variant(2)
for _ in range(3):
    H.attention(qkv, out, B, N, heads)
torch.cuda.synchronize()
t = buf.cpu().reshape(nblk, 8, 8).double()
jm = t[0, 0, 5].item()
names = ["M: copies + 16 MFMA issue", "vmcnt(4) wait", "barrier after M", "V: reads + softmax + lgkm", "barrier after V"]
for half, sl in (("half 0 (waves 0-3)", slice(0, 4)), ("half 1 (waves 4-7)", slice(4, 8))):
    print(half, f"-- s_memtime ticks per interval pair, mean over {nblk} workgroups (jmax = {jm:.0f})")
    tot = 0
    for k in range(5):
        v = t[:, sl, k].mean().item() / jm
        tot += v
        print(f"   {names[k]:32s} {v:8.1f}")
    print(f"   {'sum per tile':32s} {tot:8.1f}")
