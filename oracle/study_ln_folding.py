"""TEST INFRASTRUCTURE ONLY -- CPU numerics study for a possible next step ("Next"): folding the block
LayerNorms into the GEMMs that follow them,

    LN(x) W^T + b  =  rstd * (x (gamma*W)^T - mu * s) + c,      s_n = sum_k (gamma*W)_nk,   c = W beta + b,

with x itself (not LN(x)) rounded to the fp16 operand type.  Emulates both variants with the oracle's operand rounding and
reports relative L1 of the depth map against the fp32 oracle.      python oracle/study_ln_folding.py [vits|vitb|vitl]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import dav2_oracle as O  # noqa: E402
from src.models import get_model  # noqa: E402
from src.util.synth_weights import fill_state_dict_, make_inputs  # noqa: E402


def folded_linear(nm, x, ln_w, ln_b, W, b):
    mu = x.mean(-1, keepdim=True)
    rstd = (x.var(-1, unbiased=False, keepdim=True) + O.LN_EPS).rsqrt()
    Wp = nm.q(W * ln_w[None, :])
    acc = F.linear(nm.q(x), Wp)
    return rstd * (acc - mu * Wp.sum(1)) + (F.linear(ln_b[None], W)[0] + b)


def folded_block(nm, sd, p, x, heads, kind):
    assert kind == "mlp"
    B, N, C = x.shape
    d = C // heads
    qkv = folded_linear(nm, x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * d ** -0.5, qkv[1], qkv[2]
    attn = nm.matmul(q, k.transpose(-2, -1)).softmax(dim=-1)
    o = nm.matmul(attn, v).transpose(1, 2).reshape(B, N, C)
    x = x + nm.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"]) * sd[p + "ls1.gamma"]
    h = F.gelu(folded_linear(nm, x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + nm.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"]) * sd[p + "ls2.gamma"]


def main(encoder):
    torch.manual_seed(0)
    gt, loss = "mask+observation", "entire_target_object"
    m = get_model("AmodalDAv2", guide_type=gt, loss_stategy=loss, encoder=encoder, pretrained=False)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    fill_state_dict_(sd, 0)
    x, _, mask, obs = make_inputs(1, 518, 518, 0)
    ref = O.amodal_forward(sd, encoder, gt, loss, x, None, mask, obs)
    std = O.amodal_forward(sd, encoder, gt, loss, x, None, mask, obs, operand_dtype=torch.float16)
    orig = O.block
    O.block = folded_block
    try:
        fold = O.amodal_forward(sd, encoder, gt, loss, x, None, mask, obs, operand_dtype=torch.float16)
    finally:
        O.block = orig
    print(f"{encoder}: fp16 operands, LayerNorm output rounded (today): rel-L1 {O.rel_l1(std, ref):.3e}")
    print(f"{encoder}: fp16 operands, LayerNorm folded (x rounded):       rel-L1 {O.rel_l1(fold, ref):.3e}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "vits")
