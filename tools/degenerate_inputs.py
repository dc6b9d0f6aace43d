#!/usr/bin/env python
"""Whole model against the oracle (run on the box's host cores) on degenerate inputs: constant images, empty / full masks, constant observations,
a per-pixel 0/1 checkerboard.  ViT-S and ViT-B amodal (sigmoid) at 518 x 518 and 126 x 154, the weights' logit offset centred on the noise inputs
of the same seed (so that a constant input does not just saturate the sigmoid).  Measurement tool (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from _cases import build_product_model, case_inputs, oracle_forward, rel_l1, synth_state_dict  # noqa: E402


def variants(x, mask, obs):
    H, W = x.shape[-2:]
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    cb = ((yy + xx) % 2).float()
    yield "all zeros", torch.zeros_like(x), torch.zeros_like(mask), torch.zeros_like(obs)
    yield "all ones", torch.ones_like(x), torch.ones_like(mask), torch.ones_like(obs)
    yield "noise image, full mask, zero observation", x, torch.ones_like(mask), torch.zeros_like(obs)
    yield "noise image, empty mask, noise observation", x, torch.zeros_like(mask), obs
    yield "checkerboard image / mask / observation", cb.expand_as(x).clone(), cb.expand_as(mask).clone(), (1 - cb).expand_as(obs).clone()


def main():
    worst = 0.0
    for enc, H, W in (("vits", 518, 518), ("vitb", 518, 518), ("vitb", 126, 154)):
        spec = dict(kind="amodal", encoder=enc, guide_type="mask+observation", loss="entire_target_object", B=2, H=H, W=W, seed=400)
        model = build_product_model(spec)
        sd = synth_state_dict(model)
        x, grgb, mask, obs = case_inputs(spec)
        tr = {}
        oracle_forward(sd, spec, x, grgb, mask, obs, trace=tr)
        key = "encoder.depth_head.scratch.output_conv2.2.bias"
        sd[key] = sd[key] - float(tr["logits"].mean())
        model.load_state_dict(sd, strict=True)
        model = model.cuda()
        for name, xv, mv, ov in variants(x, mask, obs):
            ref = oracle_forward(sd, spec, xv, grgb, mv, ov)
            with torch.no_grad():
                out = model(xv.cuda(), guide_rgb=None, guide_mask=mv.cuda(), observation=ov.cuda()).cpu()
            err = rel_l1(out, ref)
            worst = max(worst, err)
            print(f"{enc} {H} x {W} {name:48s} rel-L1 vs oracle = {err:.3e}  output mean {float(ref.mean()):.3f} std {float(ref.std()):.3f}", flush=True)
    print(f"worst {worst:.3e}")


if __name__ == "__main__":
    main()
