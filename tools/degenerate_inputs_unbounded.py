#!/usr/bin/env python
"""The models WITHOUT a sigmoid (raw ReLU heads, 'ssi' heads) on constant / checkerboard inputs, where every patch token is the same up to its position and
rounding errors add coherently -- the corner that needed the third rung on the sigmoid heads.  Whole model against the CPU oracle run on the box, small sizes.
PROBE_UNCENTRED=1 leaves the raw models' synthetic logits where they fall (most of the ReLU map clipped to zero: the metric's ill-conditioned corner).
Measurement tool (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from _cases import build_product_model, oracle_forward, rel_l1, synth_state_dict  # noqa: E402
from src.util.synth_weights import make_inputs  # noqa: E402

RAW = {"vits": (64, [48, 96, 192, 384]), "vitb": (128, [96, 192, 384, 768]), "vitl": (256, [256, 512, 1024, 1024]), "vitg": (384, [1536] * 4)}


# (f8_terms, encoder_precision): the default policy; its correction terms on the fp16 pipe; every encoder block in split precision, fp8 / fp16 terms
POLICIES = [(None, "auto"), ("none", "auto"), ("both", 99), ("none", 99)]


def main():
    worst = 0.0
    print("# columns: default policy | fp16 correction terms | + every encoder block split (fp8 terms) | the same with fp16 terms")
    specs = [dict(kind="raw", encoder=e, features=RAW[e][0], out_channels=RAW[e][1], B=1, H=126, W=154, seed=900 + i) for i, e in enumerate(RAW)]
    specs += [dict(kind="amodal", encoder=e, guide_type="mask+observation", loss="invisible_part_ssi", B=1, H=126, W=154, seed=910 + i) for i, e in enumerate(("vits", "vitb", "vitl"))]
    specs += [dict(kind="raw", encoder="vitb", features=128, out_channels=RAW["vitb"][1], B=1, H=518, W=518, seed=920),
              dict(kind="amodal", encoder="vitb", guide_type="mask+observation", loss="invisible_part_ssi", B=1, H=518, W=518, seed=921)]
    draws = os.environ.get("PROBE_DRAWS")
    if draws:      # more weight draws on plain inputs (the raw ViT-S draw that 4 split blocks did not hold came from here): no degenerate inputs, two sizes
        specs = [dict(kind="raw", encoder=e, features=RAW[e][0], out_channels=RAW[e][1], B=1, H=H, W=W, seed=940 + i) for i, e in enumerate(("vits", "vitb", "vitl")) for (H, W) in ((126, 154), (266, 322))]
        specs += [dict(kind="amodal", encoder=e, guide_type="mask+observation", loss="invisible_part_ssi", B=1, H=126, W=154, seed=950 + i) for i, e in enumerate(("vits", "vitb", "vitl"))]
    for spec in specs:
        model = build_product_model(spec)
        for wseed in ((2, 3, 4, 5) if draws else (0, 1)):
            sd = synth_state_dict(model, seed=wseed)
            if spec["kind"] == "raw" and not os.environ.get("PROBE_UNCENTRED"):      # as the reference fixtures do (oracle/make_golden.py): the logits of the NOISE input centred at +1.5, most of the map positive
                tr = {}
                oracle_forward(sd, spec, make_inputs(1, spec["H"], spec["W"], spec["seed"], style="noise")[0], None, None, None, trace=tr)
                key = "depth_head.scratch.output_conv2.2.bias"
                sd[key] = sd[key] - (float(tr["logits"].mean()) - 1.5)
            model.load_state_dict(sd, strict=True)
            model = model.cuda()
            for style in (("noise", "structured") if draws else ("noise", "zeros", "checker")):
                x, grgb, mask, obs = make_inputs(1, spec["H"], spec["W"], spec["seed"], style=style)
                holder = model if spec["kind"] == "raw" else model.encoder      # the module that carries the engine's policy attributes
                ref = oracle_forward(sd, spec, x, None, None, None) if spec["kind"] == "raw" else oracle_forward(sd, spec, x, grgb, mask, obs)
                denom = float(ref.abs().mean())
                errs = []
                for f8, encp in POLICIES:
                    holder.f8_terms, holder.encoder_precision = f8, encp
                    with torch.no_grad():
                        out = (model(x.cuda()) if spec["kind"] == "raw" else model(x.cuda(), guide_rgb=grgb.cuda(), guide_mask=mask.cuda(), observation=obs.cuda())).cpu()
                    errs.append(rel_l1(out, ref) if denom > 0 else float("nan"))
                    if (f8, encp) == POLICIES[0]:
                        e_ = holder._engine()
                        dv = float(e_.last_diversity[0]) if e_.last_diversity is not None else float("nan")
                        rung = "re-run" if e_.escalated3 else "as is"
                        dvi = float(e_.last_input_diversity[0]) if e_.last_input_diversity is not None else float("nan")
                holder.f8_terms, holder.encoder_precision = None, "auto"
                worst = max(worst, errs[0] if errs[0] == errs[0] else 0.0)
                print(f"{spec['kind']:6s} {spec['encoder']} {spec.get('loss', 'relu head'):20s} {spec['H']}x{spec['W']} weights {wseed} {style:8s} rel-L1 vs oracle = " + "  ".join(f"{e:.3e}" for e in errs) +
                      f"   token diversity {dv:.3f}, of the input {dvi:.1e} ({rung}); mean |ref| {denom:.3e}, std {float(ref.std()):.3e}, zeros in ref {float((ref == 0).float().mean()):.2f}", flush=True)
            model = model.cpu()
    print(f"worst {worst:.3e}")


if __name__ == "__main__":
    main()
