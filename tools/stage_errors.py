#!/usr/bin/env python
"""Per-stage error of the product against the fp32 oracle, as a function of the number K of leading transformer blocks whose linear layers run in
split precision (module.encoder_precision = K): the residual stream behind EVERY block, the four normalised taps and the depth map.  Written to
root-cause the non-monotone parity of raw ViT-G in K (8 blocks 7.8e-4, 12 blocks 1.14e-3, 16 blocks 8.0e-4, all 40 1.6e-3).
The oracle is the checker here (a measurement tool, like tests/): the product path never imports it.
    KS=0,8,12,16,40 python tools/stage_errors.py [fixture ...]        (default fixtures: raw_vitg_224 raw_vitg_224_w1)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from _cases import build_product_model, case_inputs, load_golden, rel_l1, synth_state_dict  # noqa: E402
from oracle import dav2_oracle as O  # noqa: E402

KS = [int(k) for k in os.environ.get("KS", "0,8,12,16,40").split(",")]
HEAD = os.environ.get("HEAD", "auto")
F8 = os.environ.get("F8")          # none | both ...: module.f8_terms


def rel(a, b):      # mean |a - b| / mean |b|
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().mean() / b.abs().mean())


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["raw_vitg_224", "raw_vitg_224_w1"]
    for name in names:
        gold, meta = load_golden(name)
        case = meta["case"]
        model = build_product_model(case)
        sd = synth_state_dict(model, meta)
        model.load_state_dict(sd, strict=True)
        x, grgb, mask, obs = case_inputs(case)
        trace = {"_every_block": True}
        raw = case["kind"] == "raw"
        with torch.no_grad():
            if raw:
                ref = O.raw_forward({k: v.float() for k, v in sd.items()}, case["encoder"], x, trace=trace, **({"use_clstoken": True} if case.get("use_clstoken") else {}))
            else:
                ref = O.amodal_forward({k: v.float() for k, v in sd.items()}, case["encoder"], case["guide_type"], case["loss"], x, grgb, mask, obs, trace=trace)
        depth = len([k for k in trace if k.startswith("block")])
        model = model.cuda()
        owner = model if raw else model.encoder
        xs = [t.cuda() for t in (x, grgb, mask, obs)]
        run = (lambda: model(xs[0])) if raw else (lambda: model(xs[0], guide_rgb=xs[1], guide_mask=xs[2], observation=xs[3]))
        st = case["stride"]
        table = {}
        for K in KS:
            owner.head_precision, owner.encoder_precision, owner.precision_ladder = HEAD, min(K, depth), False
            if F8:
                owner.f8_terms = F8
            object.__setattr__(owner, "_engine_obj", None)
            object.__setattr__(owner, "_engine_stamp", None)
            eng = owner._engine()
            got = {}
            eng.block_probe = lambda i, ws, _g=got: _g.__setitem__(i, ws.x.clone())
            with torch.no_grad():
                out = run()
            eng.block_probe = None
            ws = next(iter(eng._ws.values()))
            B, N, D = trace["block0"].shape
            row = {f"b{i}": rel(got[i].view(B, N, D), trace[f"block{i}"]) for i in range(depth)}
            for j in range(4):
                t = ws.taps[j][:, :D].float().view(B, N - 1, D)
                row[f"tap{j}"] = rel(t, trace[f"tap{j}"])
            row["out"] = rel_l1(out[..., ::st, ::st].cpu(), gold)
            row["out_full"] = rel(out, ref)
            table[K] = row
        print(f"# {name}: {case['encoder']} {case['H']}x{case['W']} B={case['B']}; head={HEAD} f8={F8 or 'policy'}; rel-L1 (mean|a-b| / mean|b|) of each stage against the fp32 oracle")
        keys = [f"b{i}" for i in range(depth)] + [f"tap{j}" for j in range(4)] + ["out", "out_full"]
        print(f"{'stage':>9s} " + " ".join(f"K={K:<9d}" for K in KS))
        for k in keys:
            print(f"{k:>9s} " + " ".join(f"{table[K][k]:<11.3e}" for K in KS))
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
