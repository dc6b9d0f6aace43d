#!/usr/bin/env python
"""Whole model against the oracle (run on the box's host cores) at extreme aspect ratios and sizes the fixtures and the fuzz sweep do not reach:
one patch row / column up to 73 patches long, 2 x 146, a 1 x 1 grid at batch 5.  ViT-S amodal (sigmoid and ssi) and raw.  Measurement tool (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from _cases import build_product_model, case_inputs, oracle_forward, rel_l1, synth_state_dict  # noqa: E402

SIZES = [(1, 14, 1022), (1, 1022, 14), (2, 28, 2044), (1, 2044, 28), (5, 14, 14), (1, 42, 1498), (3, 14, 70)]


def main():
    worst = 0.0
    for i, (B, H, W) in enumerate(SIZES):
        for kind, loss in (("amodal", "entire_target_object"), ("amodal", "invisible_part_ssi"), ("raw", "")):
            if kind == "raw":
                spec = dict(kind="raw", encoder="vits", features=64, out_channels=[48, 96, 192, 384], B=B, H=H, W=W, seed=300 + i)
            else:
                spec = dict(kind="amodal", encoder="vits", guide_type="mask+observation", loss=loss, B=B, H=H, W=W, seed=300 + i)
            model = build_product_model(spec)
            sd = synth_state_dict(model)
            x, grgb, mask, obs = case_inputs(spec)
            tr = {}
            oracle_forward(sd, spec, x, grgb, mask, obs, trace=tr)
            key = ("" if kind == "raw" else "encoder.") + "depth_head.scratch.output_conv2.2.bias"
            sd[key] = sd[key] - float(tr["logits"].mean()) + (1.5 if kind == "raw" else 0.0)
            model.load_state_dict(sd, strict=True)
            ref = oracle_forward(sd, spec, x, grgb, mask, obs)
            model = model.cuda()
            with torch.no_grad():
                out = (model(x.cuda()) if kind == "raw" else model(x.cuda(), guide_rgb=grgb.cuda(), guide_mask=mask.cuda(), observation=obs.cuda())).cpu()
            err = rel_l1(out, ref)
            worst = max(worst, err)
            print(f"{kind:6s} {loss[:14]:14s} B={B} {H:4d} x {W:4d}: rel-L1 vs oracle = {err:.3e}  finite={bool(torch.isfinite(out).all())}", flush=True)
    print(f"worst {worst:.3e}")
    assert worst <= 1e-3


if __name__ == "__main__":
    main()
