#!/usr/bin/env python
"""The precision ladder's self-calibration (DepthEngine.calibrate) across models, weight draws, fills and calibration sizes: eps of each rung, the
thresholds it installs and what round 5's fitted constants were.  python tools/calibration_table.py [quick]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from _cases import build_product_model  # noqa: E402
from src.models.amodalsynthdrive.depth_anything_v2 import dpt as P  # noqa: E402
from src.util.synth_weights import fill_state_dict_  # noqa: E402

RAW = {"vits": (64, [48, 96, 192, 384]), "vitb": (128, [96, 192, 384, 768]), "vitl": (256, [256, 512, 1024, 1024]), "vitg": (384, [1536] * 4)}


def main():
    quick = "quick" in sys.argv
    rows = [("amodal", e, "entire_target_object", w, t) for e in ("vitb", "vitl") for (w, t) in ((0, "normal"), (1, "normal"), (2, "normal"), (0, "heavy"))]
    rows += [("amodal", "vits", "entire_target_object", 0, "normal"), ("amodal", "vitb", "ssi_x", 0, "normal"), ("amodal", "vitl", "ssi_x", 0, "heavy")]
    rows += [("raw", e, "", w, t) for e in ("vits", "vitb", "vitl", "vitg") for (w, t) in ((0, "normal"), (1, "normal"))]
    if quick:
        rows = rows[:2] + rows[8:9] + rows[11:13]
    sizes = [(266, 322)] if quick else [(126, 154), (266, 322), (518, 518)]
    print(f"# budget {P._LADDER_BUDGET} safety {P._LADDER_SAFETY} rule {P._LADDER_RULE}")
    print("# kind encoder loss weights fill | calibration size | eps1 eps2 | r: global cross | r3: global cross -> installed r r3 div (tap diversity: images min / flat) | seconds")
    for kind, enc, loss, wseed, tail in rows:
        case = dict(kind=kind, encoder=enc, guide_type="mask+observation", loss=loss, features=RAW[enc][0], out_channels=RAW[enc][1])
        model = build_product_model(case)
        fill_state_dict_(model.state_dict(), wseed, tail=tail)
        model = model.cuda()
        owner = model if kind == "raw" else model.encoder
        for size in sizes:
            if enc == "vitg" and size[0] > 300:
                continue
            P._LADDER_CAL_SIZE = size
            object.__setattr__(owner, "_engine_obj", None)
            object.__setattr__(owner, "_engine_stamp", None)
            object.__setattr__(owner, "_ladder_cal", None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            owner._engine()
            torch.cuda.synchronize()
            c = owner.ladder_calibration or {}
            f = lambda k: ("%.3e" % c[k]) if k in c and c[k] is not None else "-"      # noqa: E731
            print(f"{kind:6s} {enc} {loss or 'relu':22s} w{wseed} {tail:6s} | {size[0]}x{size[1]} | eps1 {f('eps1')} eps2 {f('eps2')} | r {f('r_global')} {f('r_cross')} | r3 {f('r3_global')} {f('r3_cross')} -> r {f('r_installed')} r3 {f('r3_installed')} "
                  f"div {f('div_installed')} ({f('tap_diversity_images_min')} / {f('tap_diversity_flat')}) | {time.perf_counter() - t0:.2f} s", flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
