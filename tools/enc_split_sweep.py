#!/usr/bin/env python
"""Parity against the number of leading transformer blocks whose linear layers run in split precision (module.encoder_precision = K) and the
head policy, on every raw ViT-G fixture (default) or the named fixtures (raw or amodal); --time adds the 8 x 1022^2 step (BASELINE config 5)
per setting.
    python tools/enc_split_sweep.py [--time] [fixture ...]        (KS=0,4,8 HEADS=auto in the environment select the grid)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from _cases import build_product_model, case_inputs, load_golden, rel_l1, synth_state_dict  # noqa: E402

KS = [int(k) for k in os.environ.get("KS", "0,8,12,16,20").split(",")]
HEADS = os.environ.get("HEADS", "auto;oc1,oc2,out,rn1,rn2,rn3,proj,rs1,rs3").split(";")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    do_time = "--time" in sys.argv
    names = args or ["raw_vitg_224", "raw_vitg_224_w1", "raw_vitg_224_w2", "raw_vitg_224_heavy", "raw_vitg_224_struct", "raw_vitg_1022"]
    for name in names:
        gold, meta = load_golden(name)
        case = meta["case"]
        model = build_product_model(case)
        model.load_state_dict(synth_state_dict(model, meta), strict=True)
        model = model.cuda()
        x, grgb, mask, obs = (t.cuda() for t in case_inputs(case))
        eng_owner = model if case["kind"] == "raw" else model.encoder       # the module that carries the engine and its precision attributes
        run = (lambda inp: model(inp)) if case["kind"] == "raw" else (lambda inp: model(inp, guide_rgb=grgb, guide_mask=mask, observation=obs))
        st = case["stride"]
        x8 = case_inputs(dict(case, B=8, seed=11))[0].cuda() if (do_time and name == "raw_vitg_1022") else None
        for head in HEADS:
            for K in KS:
                eng_owner.head_precision = head
                eng_owner.encoder_precision = K
                object.__setattr__(eng_owner, "_engine_obj", None)
                object.__setattr__(eng_owner, "_engine_stamp", None)
                torch.cuda.empty_cache()
                torch.cuda.reset_peak_memory_stats()
                with torch.no_grad():
                    out = run(x)
                line = f"{name:22s} head=[{head:40s}] first {K:2d} blocks split: rel-L1 {rel_l1(out[..., ::st, ::st].cpu(), gold):.3e}"
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
