#!/usr/bin/env python
"""What does the ladder's second rung need?  The low-mean / degenerate reference fixtures of the sigmoid heads with the DPT head's contractions in split
precision group by group (head_precision = list, ladder off): relative L1 against the reference golden and ms per forward of a batch of 8 -- the
cheapest subset that holds the bar is what DepthEngine._escalate should re-run.   python tools/ladder_subsets.py [fixture ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from _cases import case_inputs, fixture_model, load_golden, rel_l1  # noqa: E402

ALL = "proj,rs0,rs1,rs3,ip0,ip1,ip2,ip3,rn0,rn1,rn2,rn3,rcu0,rcu1,rcu2,rcu3,out0,out1,out2,out3,oc1,oc2"
SUBSETS = {
    "rung 1 (default groups)": "auto",
    "all": ALL,
    "all - rcu0": ALL.replace("rcu0,", ""),
    "all - rcu0,1": ALL.replace("rcu0,rcu1,", ""),
    "all - rcu*": ALL.replace("rcu0,rcu1,rcu2,rcu3,", ""),
    "all - rcu* - rn0 - ip0": ALL.replace("rcu0,rcu1,rcu2,rcu3,", "").replace("rn0,", "").replace("ip0,", ""),
    "oc1,oc2,out*,proj": "proj,out0,out1,out2,out3,oc1,oc2",
    "oc2,out*,projw": "projw,out1,out2,out3,oc2",
    "all - oc2": ALL.replace(",oc2", ""),
    "all - oc1 - out0": ALL.replace("out0,", "").replace("oc1,", ""),
}


def main():
    names = sys.argv[1:] or ["vitl_518_m10", "bench_vitl_b32_low", "vitl_518_struct_m20", "vitb_518_zeros", "vitb_126x154_zeros", "vitb_518_struct_m10", "vitb_518_m20"]
    print(f"{'subset':26s} " + " ".join(f"{n[:18]:>18s}" for n in names) + "   ms (batch 8, first fixture's model)")
    data = {}
    for n in names:
        gold, meta = load_golden(n)
        data[n] = (gold, meta, fixture_model(meta), case_inputs(meta["case"]))
    for label, groups in SUBSETS.items():
        cells = []
        for n in names:
            gold, meta, model, (x, grgb, mask, obs) = data[n]
            case = meta["case"]
            with torch.no_grad():
                dict(model.named_parameters())[meta["final_bias_key"]].fill_(meta["final_bias"])
            enc = model.encoder
            enc.precision_ladder = False
            enc.head_precision = groups
            with torch.no_grad():
                out = model(x.cuda(), guide_rgb=grgb.cuda(), guide_mask=mask.cuda(), observation=obs.cuda()).cpu()
            st = case["stride"]
            sub = out[..., ::st, ::st]
            cells.append(max(rel_l1(sub[i], gold[i]) for i in range(gold.shape[0])))
        gold, meta, model, (x, grgb, mask, obs) = data[names[0]]
        xb, mb, ob = (t[:1].expand(8, -1, -1, -1).contiguous().cuda() for t in (x, mask, obs))
        with torch.no_grad():
            for _ in range(2):
                model(xb, guide_rgb=None, guide_mask=mb, observation=ob)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                model(xb, guide_rgb=None, guide_mask=mb, observation=ob)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"{label:26s} " + " ".join(f"{c:18.2e}" for c in cells) + f"   {ms:7.2f}", flush=True)
    for n in names:
        enc = data[n][2].encoder
        enc.head_precision, enc.precision_ladder = "auto", None


if __name__ == "__main__":
    main()
