#!/usr/bin/env python
"""Isolated timing of the DPT tail at the ViT-L bs=32 shape: fused kernel vs resize + tail GEMM."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hip_ext as H  # noqa: E402

op = H.operand_dtype()
B, C, hi, ho = int(os.environ.get("B", 32)), 128, 296, 518
torch.manual_seed(0)
x = torch.randn(B * hi * hi, C, device="cuda")
w = (torch.randn(32, 9 * C, device="cuda") * (9 * C) ** -0.5).to(op)
b, tw = torch.randn(32, device="cuda"), torch.randn(32, device="cuda")
out = torch.empty(B, ho, ho, device="cuda")
fin = torch.zeros(B, ho + 2, ho + 2, C, dtype=op, device="cuda")
out2 = torch.empty(B, 1, ho, ho, device="cuda")


def fused():
    H.dpt_tail(x, C, B, hi, hi, ho, ho, C, w, b, tw, 0.1, H.ACT_SIGMOID, out)


def two():
    H.bilinear(x, C, B, hi, hi, ho, ho, C, out_op=fin, ld_op=C, map_op=H.MAP_PAD)
    H.igemm(M=B * ho * ho, N=32, K=9 * C, A=fin, lda=C, W=w, a_mode=H.A_CONV3, conv=(ho, ho, ho + 2, ho + 2, 1), bias=b,
            flags=H.EP_BIAS | H.EP_TAIL, out_f32=out2, ldo_f32=1, tail_w=tw, tail_b=0.1, tail_act=H.ACT_SIGMOID)


for name, fn in (("fused", fused), ("resize + tail GEMM", two)):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"ablate={os.environ.get('ADA_TAIL_ABLATE', 0)} {name}: {e0.elapsed_time(e1) / 10:.3f} ms")
