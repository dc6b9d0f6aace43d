#!/usr/bin/env python
"""Reference fixtures of the sigmoid heads under the rungs of the precision ladder (GPU box; the goldens are the checker, no oracle run needed):
per fixture the output mean, the sigmoid compression factor r = sum s(1-s) / sum s of the first-rung output (what DepthEngine._escalate thresholds),
and the relative L1 against the reference golden with
    rung 1 only (ladder off)  |  the default policy (ladder on)  |  head in split precision  |  head + leading encoder blocks in split precision.
    python tools/parity_table.py [fixture ...]        (default: every sigmoid ViT-B / ViT-L fixture)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from _cases import case_inputs, fixture_model, golden_names_by_model, load_golden, rel_l1  # noqa: E402


def main():
    names = sys.argv[1:]
    if not names:
        names = []
        for n in golden_names_by_model():
            c = load_golden(n)[1]["case"]
            if c["kind"] == "amodal" and "ssi" not in c["loss"] and c["encoder"] in ("vitb", "vitl"):
                names.append(n)
    print(f"{'fixture':28s} {'mean':>6s} {'r':>6s} {'tokdiv':>6s} | {'rung 1':>9s} {'default':>9s} {'esc':>4s} | {'head split':>10s} {'+enc deep':>10s}")
    for name in names:
        gold, meta = load_golden(name)
        case = meta["case"]
        model = fixture_model(meta)
        enc = model.encoder
        whole = "take" in case and case["B"] <= 8
        x, grgb, mask, obs = case_inputs({k: v for k, v in case.items() if k != "take"} if whole else case)
        st = case["stride"]

        def run():
            with torch.no_grad():
                out = model(x.cuda(), guide_rgb=grgb.cuda(), guide_mask=mask.cuda(), observation=obs.cuda()).cpu()
            if whole:
                out = out[case["take"]]
            sub = out[..., ::st, ::st]
            return rel_l1(sub, gold), max(rel_l1(sub[i], gold[i]) for i in range(gold.shape[0]))
        res = {}
        enc.head_precision, enc.encoder_precision = "auto", "auto"
        enc.precision_ladder = False
        res["r1"] = run()
        enc.precision_ladder = None
        res["def"] = run()
        eng = enc._engine()
        # token diversity of the final tap (torch ops, study only): sum_c Var_p(t) / sum_c E_p[t^2] per image -- ~0 when every patch token is the same (constant inputs)
        ws = next(iter(eng._ws.values()))
        Dm = eng.w.dim
        t3 = ws.taps[3][:, :Dm].float().view(ws.B, -1, Dm)
        div = float((t3.var(dim=1, unbiased=False).sum(-1) / (t3 * t3).mean(dim=1).sum(-1)).min())
        ratio = eng.last_ratio
        rmax = float(ratio.max()) if ratio is not None else float("nan")
        nesc = int((ratio > eng.ladder.get("r", float("inf"))).sum()) if ratio is not None else 0
        enc.head_precision = "split"
        res["hs"] = run()
        enc.encoder_precision = 8 if case["encoder"] in ("vitl", "vitg") else 4
        res["hse"] = run()
        enc.head_precision, enc.encoder_precision = "auto", "auto"
        print(f"{name:28s} {meta['out_mean']:6.3f} {rmax:6.3f} {div:6.3f} | {res['r1'][1]:9.2e} {res['def'][1]:9.2e} {nesc:4d} | {res['hs'][1]:10.2e} {res['hse'][1]:10.2e}", flush=True)


if __name__ == "__main__":
    main()
