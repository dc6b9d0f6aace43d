#!/usr/bin/env python
"""Which layer groups of the DPT head need split precision?  For each fixture the product model is run with a list of group subsets
(module.head_precision) and the rel-L1 against the reference golden is printed; with --time the raw ViT-G 8 x 1022^2 step (BASELINE
config 5) is timed per subset as well.  Measurement tool (GPU box): python tools/head_split_sweep.py [--time] [fixture ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from _cases import build_product_model, case_inputs, load_golden, rel_l1, synth_state_dict  # noqa: E402

SUBSETS = ["", "oc2,out", "oc2,out,rn1,rn2,rn3", "oc2,out,rn2,rn3", "oc2,out,rn", "oc2,out,rn1,rn2,rn3,proj", "oc2,out,rn1,rn2,rn3,proj,rs1,rs3",
           "oc1,oc2,out,rn1,rn2,rn3", "oc2,out0,rn1,rn2,rn3", "oc2,out,rn3", "tok,ip,rn,out,oc1,oc2", "tok,ip,rn,rcu,out,oc1,oc2"]
if os.environ.get("SUBSETS"):
    SUBSETS = os.environ["SUBSETS"].split(";")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    do_time = "--time" in sys.argv
    names = args or ["vits_ssi_image_mask", "vits_518", "raw_vits_518", "raw_vitg_224", "raw_vitg_1022"]
    for name in names:
        gold, meta = load_golden(name)
        case = meta["case"]
        model = build_product_model(case)
        model.load_state_dict(synth_state_dict(model, meta), strict=True)
        model = model.cuda()
        x, grgb, mask, obs = (t.cuda() for t in case_inputs(case))
        st = case["stride"]
        x8 = None
        if do_time and name == "raw_vitg_1022":
            x8 = case_inputs(dict(case, B=8, seed=11))[0].cuda()
        for sub in SUBSETS:
            inner = model.encoder if hasattr(model, "encoder") and not isinstance(model.encoder, str) else model
            inner.head_precision = sub if sub else "single"
            object.__setattr__(inner, "_engine_obj", None)       # drop the previous subset's workspace before the new one is built
            object.__setattr__(inner, "_engine_stamp", None)
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            with torch.no_grad():
                out = model(x) if case["kind"] == "raw" else model(x, guide_rgb=grgb, guide_mask=mask, observation=obs)
            err = rel_l1(out[..., ::st, ::st].cpu(), gold)
            line = f"{name:24s} split=[{sub:28s}] rel-L1 {err:.3e}"
            if x8 is not None:
                with torch.no_grad():
                    model(x8)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        model(x8)
                    torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 3
                line += f"   8x1022^2: {8 / dt:6.2f} images/s  {torch.cuda.max_memory_allocated() / 2**30:5.1f} GiB peak"
            print(line, flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
