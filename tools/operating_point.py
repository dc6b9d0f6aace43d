#!/usr/bin/env python
"""How the relative L1 of a sigmoid-head model depends on where its outputs sit in (0, 1): the final bias of a reference fixture's weights is shifted so that
the mean depth moves from ~0.05 to ~0.95, and the HIP path is compared with the oracle (run on the box) at each point -- with the default precision policy
(single-precision head: the sigmoid compresses the logit noise around 0.5) and with the unbounded-head policy (head in split precision, leading encoder
blocks in split precision).  Measurement tool (GPU box):  python tools/operating_point.py [fixture ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from _cases import build_product_model, case_inputs, load_golden, oracle_forward, rel_l1, synth_state_dict  # noqa: E402

SHIFTS = [float(s) for s in os.environ.get("SHIFTS", "-3,-2,-1,0,1,2").split(",")]
POLICIES = [("auto", "auto"), ("out1,out2,out3,projw", "auto"), ("split", "auto"), ("split", "deep")]


def main():
    for name in (sys.argv[1:] or ["vitb_518", "vitl_518"]):
        _, meta = load_golden(name)
        case = meta["case"]
        model = build_product_model(case)
        sd0 = synth_state_dict(model, meta)
        x, grgb, mask, obs = case_inputs(case)
        key = meta["final_bias_key"]
        deep = 8 if case["encoder"] in ("vitl", "vitg") else 4
        for sh in SHIFTS:
            sd = dict(sd0)
            sd[key] = sd0[key] + sh
            ref = oracle_forward(sd, case, x, grgb, mask, obs)
            model.load_state_dict(sd, strict=True)
            model = model.cuda()
            line = f"{name} bias {sh:+.0f}: output mean {float(ref.mean()):.3f} |"
            for head, enc in POLICIES:
                model.encoder.head_precision = head
                model.encoder.encoder_precision = deep if enc == "deep" else "auto"
                with torch.no_grad():
                    out = model(x.cuda(), guide_rgb=grgb.cuda(), guide_mask=mask.cuda(), observation=obs.cuda()).cpu()
                line += f" head={head}, enc={enc}: {rel_l1(out, ref):.2e} (abs {float((out - ref).abs().mean()):.1e}) |"
            print(line, flush=True)
            model = model.cpu()


if __name__ == "__main__":
    main()
