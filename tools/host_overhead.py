#!/usr/bin/env python
"""Host-side cost of enqueuing one forward (no device sync inside the loop) vs the device time per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch
from src.models import get_model
from src.util.synth_weights import fill_state_dict_, make_inputs
m = get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="x", encoder="vitl", pretrained=False).eval()
sd = {k: v.clone() for k, v in m.state_dict().items()}; fill_state_dict_(sd, 0); m.load_state_dict(sd); m = m.cuda()
x, _, mask, obs = make_inputs(32, 518, 518, 0, device="cuda")
with torch.no_grad():
    for _ in range(2): m(x, guide_mask=mask, observation=obs)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 5
    host = []
    for _ in range(n):
        h0 = time.perf_counter(); m(x, guide_mask=mask, observation=obs); host.append(time.perf_counter() - h0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("host enqueue ms per forward:", [round(h * 1e3, 2) for h in host], " wall ms per step:", round(dt / n * 1e3, 2))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
with torch.no_grad():
    m(x, guide_mask=mask, observation=obs)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
