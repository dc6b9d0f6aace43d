#!/usr/bin/env python
"""Throughput of the AmodalDAv2 ViT-L forward at several batch sizes (images/s, ms per image): does a smaller batch -- whose per-block working
set (residual stream, LayerNorm output, qkv, attention output, MLP hidden: ~31 MB per image) fits the 256 MB Infinity Cache -- beat batch 32
in spite of the tile quantisation it costs?     python tools/batch_sweep.py [--batches 4,8,12,16,24,32] [--steps 10]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="4,8,12,16,24,32")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--encoder", default="vitl")
args = ap.parse_args()
from src.models import get_model  # noqa: E402
from src.util.synth_weights import fill_state_dict_, make_inputs  # noqa: E402

m = get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="entire_target_object", encoder=args.encoder, pretrained=False).eval()
sd = {k: v.clone() for k, v in m.state_dict().items()}
fill_state_dict_(sd, 0)
m.load_state_dict(sd)
m = m.cuda()
for B in [int(b) for b in args.batches.split(",")]:
    x, _, mask, obs = make_inputs(B, 518, 518, 0, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            m(x, guide_mask=mask, observation=obs)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                m(x, guide_mask=mask, observation=obs)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / args.steps)
    print(f"B={B:3d}: {1e3 * best:7.2f} ms/step  {1e3 * best / B:6.3f} ms/image  {B / best:7.1f} images/s", flush=True)
    m.encoder._engine()._ws.clear()
    torch.cuda.empty_cache()
