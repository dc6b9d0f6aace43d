#!/usr/bin/env python
"""Single-image latency of AmodalDAv2 (the infer.py use case) and small-batch throughput, with and without HIP-graph replay."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch
from hip_ext import engine as E
from src.models import get_model
from src.util.synth_weights import centred_final_bias, fill_state_dict_, make_inputs
for enc in (sys.argv[1:] or ["vitl"]):
    m = get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="x", encoder=enc, pretrained=False).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}; fill_state_dict_(sd, 0)
    cb = centred_final_bias(enc, ROOT)      # centred logits: the default (first-rung) path is what is timed
    if cb: sd[cb[0]] = torch.full_like(sd[cb[0]], cb[1])
    m.load_state_dict(sd); m = m.cuda()
    for B in (1, 2, 4, 8):
        x, _, mask, obs = make_inputs(B, 518, 518, 0, device="cuda")
        res = {}
        for mode in ("0", "1"):
            E.GRAPH_MODE = mode
            with torch.no_grad():
                for _ in range(3): out = m(x, guide_mask=mask, observation=obs)
                torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
                for _ in range(n): out = m(x, guide_mask=mask, observation=obs)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            res[mode] = (dt, out.clone())
        same = torch.equal(res["0"][1], res["1"][1])
        d0, d1 = res["0"][0], res["1"][0]
        eng = m.encoder._engine()
        print(f"{enc} B={B}: [ladder re-ran {eng.escalated} images so far] launches {d0 * 1e3:7.2f} ms ({B / d0:7.1f} images/s)   graph replay {d1 * 1e3:7.2f} ms ({B / d1:7.1f} images/s)   bit-identical: {same}")
