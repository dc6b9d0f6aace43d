#!/usr/bin/env python
"""Per-shape micro-benchmark of the libada_hip kernels at the ViT-L / bs=32 / 518x518 shapes of the forward pass.
Each distinct launch is timed alone with HIP events (warm L2/MALL, random operands) -> TFLOP/s or GB/s table.
Usage: python tools/bench_shapes.py [--batch 32] [--reps 5]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hip_ext  # noqa: E402
from hip_ext import engine as E  # noqa: E402

E.GRAPH_MODE = "0"   # the launches are recorded through the Python wrappers: no graph replay here


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--encoder", default="vitl")
    args = ap.parse_args()
    from src.models import get_model
    from src.util.synth_weights import centred_final_bias, fill_state_dict_, make_inputs
    m = get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="x", encoder=args.encoder, pretrained=False).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    fill_state_dict_(sd, 0)
    cb = centred_final_bias(args.encoder, ROOT)
    if cb:
        sd[cb[0]] = torch.full_like(sd[cb[0]], cb[1])
    m.load_state_dict(sd)
    m = m.cuda()
    x, _, mask, obs = make_inputs(args.batch, 518, 518, 0, device="cuda")
    with torch.no_grad():
        m(x, guide_mask=mask, observation=obs)  # warm-up, allocates the workspace
    # record every launch of one forward with its arguments, then replay each distinct one in isolation
    calls = []
    real = {n: getattr(E, "k_" + n) for n in ("igemm", "attention", "layernorm", "patchify", "write_cls", "bilinear", "dpt_tail", "tapsum_resize")}

    def rec(name):
        def f(*a, **k):
            calls.append((name, a, k))
            return real[name](*a, **k)
        return f
    for n in real:
        setattr(E, "k_" + n, rec(n))
    with torch.no_grad():
        m(x, guide_mask=mask, observation=obs)
    for n in real:
        setattr(E, "k_" + n, real[n])
    torch.cuda.synchronize()

    def key(name, a, k):
        if name == "igemm":
            return (name, k["M"], k["N"], k["K"], k.get("a_mode", 0), k.get("flags", 0), k.get("map_op", 0), k.get("out_f32") is not None, k.get("out_op") is not None,
                    (k.get("conv") or (0,) * 5)[4])
        if name == "attention":
            return (name,) + tuple(a[2:])
        if name == "layernorm":
            return (name, a[2], a[3], k.get("map_op", 0), k.get("out_f32") is not None, k.get("out2_op") is not None, k.get("unshuffle_s", 0))
        if name == "bilinear":
            return (name,) + tuple(a[2:8]) + (k.get("add") is not None,)
        return (name,)
    groups = {}
    for name, a, k in calls:
        groups.setdefault(key(name, a, k), []).append((a, k))
    rows = []
    for kk, lst in groups.items():
        name = kk[0]
        a, k = lst[0]
        for _ in range(2):
            real[name](*a, **k)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(args.reps):
            real[name](*a, **k)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / args.reps
        info = dict(kernel=name, count=len(lst), ms=ms, total_ms=ms * len(lst))
        if name == "igemm":
            flop = 2.0 * k["M"] * k["N"] * (k.get("k_alg") or k["K"])
            info.update(M=k["M"], N=k["N"], K=k["K"], conv=k.get("a_mode", 0), flags=hex(k.get("flags", 0)), tflops=flop / ms / 1e9)
        elif name == "attention":
            flop = 4.0 * a[2] * a[4] * 64 * a[3] ** 2
            info.update(B=a[2], N=a[3], heads=a[4], tflops=flop / ms / 1e9)
        elif name == "layernorm":
            byts = a[2] * a[3] * (4 + 2) + (a[2] * a[3] * 2 if k.get("out2_op") is not None else 0)
            info.update(rows=a[2], dim=a[3], gbps=byts / ms / 1e6, second_output=k.get("out2_op") is not None, unshuffle=k.get("unshuffle_s", 0))
        elif name == "bilinear":
            info.update(shape=list(a[2:8]))
        elif name == "dpt_tail":
            info.update(shape=list(a[2:8]), tflops=2.0 * a[2] * a[5] * a[6] * 32 * 9 * a[7] / ms / 1e9)
        elif name == "tapsum_resize":
            info.update(shape=list(a[2:8]), tap_map_dtype=str(a[0].dtype), gbps=(a[0].numel() * a[0].element_size() + a[2] * a[5] * a[6] * a[7] * 4) / ms / 1e6)
        rows.append(info)
    rows.sort(key=lambda r: -r["total_ms"])
    tot = sum(r["total_ms"] for r in rows)
    print(f"sum of isolated kernel times: {tot:.2f} ms per forward (batch {args.batch})")
    for r in rows:
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}))


if __name__ == "__main__":
    main()
