#!/usr/bin/env python
"""Runs the other BASELINE.json configurations once on the GPU (finite-output + throughput sanity; parity for these model
families is covered by tests/test_gpu_model.py at smaller sizes):
   config 2: AmodalDAv2 ViT-B, bs=8, 518x518      config 5: raw Depth-Anything-V2 ViT-G, bs=8, 1022x1022 (1024 is not a
   multiple of 14 -- SURVEY.md 0.4)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch
from src.models import get_model
from src.models.amodalsynthdrive.depth_anything_v2_raw.dpt import DepthAnythingV2 as Raw
from src.util.synth_weights import centred_final_bias, fill_state_dict_, make_inputs


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / n


def synth(m, encoder=None):
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    fill_state_dict_(sd, 0)
    cb = centred_final_bias(encoder, ROOT) if encoder else None     # sigmoid model: centred logits (the default, first-rung path is what is timed)
    if cb:
        sd[cb[0]] = torch.full_like(sd[cb[0]], cb[1])
    m.load_state_dict(sd)
    return m.cuda().eval()


with torch.no_grad():
    m = synth(get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="entire_target_object", encoder="vitb", pretrained=False), "vitb")
    x, _, mask, obs = make_inputs(8, 518, 518, 0, device="cuda")
    out, dt = timeit(lambda: m(x, guide_mask=mask, observation=obs))
    print(f"config 2  ViT-B bs=8 518x518: {8 / dt:.1f} images/s ({dt * 1e3:.1f} ms/step, {8 / dt * 396.26 / 1e3:.0f} TFLOP/s), finite={bool(torch.isfinite(out).all())}, range=({float(out.min()):.3f},{float(out.max()):.3f}), mean {float(out.mean()):.3f}, ladder re-ran {m.encoder._engine().escalated} images")
    del m, out
    torch.cuda.empty_cache()
    r = synth(Raw(encoder="vitg", features=384, out_channels=[1536] * 4))
    xg = torch.randn(8, 3, 1022, 1022, device="cuda")
    # The synthetic logits of the raw model fall where they fall: most of the ReLU map clipped, the rest just above the kink (r = N+ / sum out up to 0.8).  Such
    # images take the ladder's third rung (round 6) -- measured below as the "un-centred twin".  Real base-depth checkpoints predict a disparity that is positive
    # almost everywhere, which is what the reference fixtures model by moving the final bias (oracle/make_golden.py: mean logit -> 1.5): the headline of config 5.
    eng = r._engine()
    out_u, dt_u = timeit(lambda: r(xg), n=2)
    esc_u, r_u = eng.escalated, [float(v) for v in eng.last_ratio]
    from hip_ext import ACT_NONE
    r.precision_ladder = False
    eng0 = r._engine()
    eng0.final_act = ACT_NONE
    mean_logit = float(r(xg[:2]).mean())
    r.precision_ladder = None
    with torch.no_grad():
        r.depth_head.scratch.output_conv2[2].bias.add_(1.5 - mean_logit)
    out, dt = timeit(lambda: r(xg), n=2)
    eng = r._engine()
    print(f"config 5  raw ViT-G bs=8 1022x1022: {8 / dt:.2f} images/s ({dt * 1e3:.1f} ms/step, {8 / dt * 22824.89 / 1e3:.0f} TFLOP/s), finite={bool(torch.isfinite(out).all())}, shape={tuple(out.shape)}, "
          f"mem={torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; final bias moved so that the mean logit is 1.5 (as the reference fixtures do): r = N+ / sum out "
          f"{min(float(v) for v in eng.last_ratio):.3f}..{max(float(v) for v in eng.last_ratio):.3f} against the calibrated {eng.ladder.get('r3'):.3f}, ladder re-ran {eng.escalated} images "
          f"({r.ladder_calibration and {k: r.ladder_calibration[k] for k in ('eps1', 'r_global', 'r_cross')}})")
    print(f"config 5  un-centred twin (the synthetic logits as they fall, {float((out_u == 0).float().mean()):.0%} of the map clipped): {8 / dt_u:.2f} images/s ({dt_u * 1e3:.1f} ms/step); "
          f"r {min(r_u):.3f}..{max(r_u):.3f}: the ladder re-ran {esc_u} images in the timing calls (third rung: every contraction split)")
