#!/usr/bin/env python
"""Which precision policy holds the 1e-3 bar for THIS checkpoint?  (GPU box; the fp32 CPU oracle is the checker -- a tool, not the product path.)

The default policy (src/models/.../dpt.py: _head_split_policy, _encoder_split_policy, the sigmoid heads' precision ladder) was chosen on synthetic
fills; a real checkpoint can sit elsewhere (massive activations, logits far from 0).  For a state dict this runs, per input kind x resolution,
    * DepthEngine.saturation_report  -- operand-typed activations at the fp16 clamp (should be empty),
    * the product against the oracle on: i.i.d. noise, image-like (smooth RGB / ellipse mask / smooth observation), all-zero and checkerboard inputs,
      and (sigmoid heads) the same inputs with the final bias moved so that the depth map averages 0.10,
at 126 x 154 and 518 x 518, and prints the relative L1 under the default policy and -- where that misses the bar -- under the wider ones
(head in split precision; head + leading encoder blocks), i.e. the row of the policy table the checkpoint is in.

    python tools/qualify_checkpoint.py --encoder vitl [--weights model.safetensors | --synthetic-seed 0 [--tail heavy]] [--raw] [--sizes 126x154,518x518]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from _cases import build_product_model, oracle_forward, rel_l1  # noqa: E402
from src.util.synth_weights import fill_state_dict_, make_inputs  # noqa: E402

BAR = 1e-3
RAW_CFG = {"vits": (64, [48, 96, 192, 384]), "vitb": (128, [96, 192, 384, 768]), "vitl": (256, [256, 512, 1024, 1024]), "vitg": (384, [1536] * 4)}


def load_weights(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return sd.get("state_dict", sd) if isinstance(sd, dict) else sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--encoder", default="vitl")
    ap.add_argument("--raw", action="store_true", help="raw Depth-Anything-V2 (ReLU head) instead of AmodalDAv2")
    ap.add_argument("--guide-type", default="mask+observation")
    ap.add_argument("--loss", default="entire_target_object")
    ap.add_argument("--weights", default=None)
    ap.add_argument("--synthetic-seed", type=int, default=0)
    ap.add_argument("--tail", default="normal")
    ap.add_argument("--sizes", default="126x154,518x518")
    ap.add_argument("--seed", type=int, default=7)
    a = ap.parse_args()
    if a.raw:
        f, oc = RAW_CFG[a.encoder]
        case = dict(kind="raw", encoder=a.encoder, features=f, out_channels=oc)
    else:
        case = dict(kind="amodal", encoder=a.encoder, guide_type=a.guide_type, loss=a.loss)
    model = build_product_model(case)
    if a.weights:
        model.load_state_dict(load_weights(a.weights), strict=True)
        src = a.weights
    else:
        fill_state_dict_(model.state_dict(), a.synthetic_seed, tail=a.tail)
        src = f"synthetic fill seed {a.synthetic_seed} ({a.tail})"
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    enc = model if a.raw else model.encoder
    sigmoid = enc.depth_head.final_act == "sigmoid"
    bias_key = ("" if a.raw else "encoder.") + "depth_head.scratch.output_conv2.2.bias"
    print(f"# {('raw DepthAnythingV2' if a.raw else 'AmodalDAv2')} {a.encoder}, head activation {enc.depth_head.final_act}; weights: {src}; bar {BAR:g}")
    model = model.cuda()
    enc._engine()       # packs the weights and -- round 6 -- calibrates the ladder's thresholds for THIS checkpoint on the device (no oracle involved)
    cal = getattr(enc, "ladder_calibration", None)
    if cal:
        print("# ladder calibration (module.ladder_calibration): " + ", ".join(f"{k} {v:.4g}" if isinstance(v, float) else f"{k} {v}" for k, v in cal.items()))
    worst = {}
    for size in a.sizes.split(","):
        H, W = (int(v) for v in size.split("x"))
        for style in ("noise", "structured", "zeros", "checker"):
            x, grgb, mask, obs = make_inputs(1, H, W, a.seed, style=style)
            if a.raw:
                x = (x - torch.tensor([0.485, 0.456, 0.406]).view(-1, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(-1, 1, 1)
            shifts = [("", 0.0)]
            if sigmoid:   # the same input with the maps moved to an average of 0.10 (where the sigmoid stops compressing the logit error)
                tr = {}
                oracle_forward(sd, case, x, grgb, mask, obs, trace=tr)
                lg = tr["logits"].double()
                lo_, hi_ = -60.0, 60.0
                for _ in range(60):
                    mid = 0.5 * (lo_ + hi_)
                    lo_, hi_ = (mid, hi_) if float(torch.sigmoid(lg - mid).mean()) > 0.10 else (lo_, mid)
                shifts.append((" @mean 0.10", -0.5 * (lo_ + hi_)))
            for tag, sh in shifts:
                sd2 = dict(sd)
                sd2[bias_key] = sd[bias_key] + sh
                with torch.no_grad():
                    dict(model.named_parameters())[bias_key].copy_(sd2[bias_key])
                ref = oracle_forward(sd2, case, x, grgb, mask, obs)
                line = f"{H:4d}x{W:<4d} {style + tag:22s} ref mean {float(ref.mean()):6.3f} |"
                for name, head, encp in (("default", "auto", "auto"), ("head split", "split", "auto"), ("head + encoder split", "split", 8 if a.encoder in ("vitl", "vitg") else 4),
                                         ("head + every encoder block split", "split", 99)):
                    enc.head_precision, enc.encoder_precision = head, encp
                    eng0 = enc._engine()
                    n2, n3 = eng0.escalated, eng0.escalated3
                    with torch.no_grad():
                        if a.raw:
                            out = model(x.cuda()).cpu()
                        else:
                            out = model(x.cuda(), guide_rgb=grgb.cuda(), guide_mask=mask.cuda(), observation=obs.cuda()).cpu()
                    e = rel_l1(out, ref)
                    eng = enc._engine()
                    extra = ""
                    if name == "default":
                        if eng.last_ratio is not None and eng.ladder is not None:
                            rung = ", rung 3" if eng.escalated3 > n3 else (", rung 2" if eng.escalated > n2 else "")
                            extra = f" (r {float(eng.last_ratio.max()):.2f}{rung})"
                        guide = None if a.raw else model.build_guide(grgb.cuda(), mask.cuda(), obs.cuda())
                        _, rep = eng.saturation_report(x.cuda(), guide)
                        if rep:
                            extra += f" SATURATED: {sum(rep.values())} elements in {len(rep)} tensors"
                    line += f" {name} {e:.2e}{extra} |"
                    worst[name] = max(worst.get(name, 0.0), e)
                    if e <= BAR:
                        break
                print(line, flush=True)
        enc.head_precision, enc.encoder_precision = "auto", "auto"
    print("# worst case per policy (a wider policy is only run where the narrower one missed the bar): " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    ok = worst.get("default", 0.0) <= BAR
    print(f"# verdict: the default policy {'holds' if ok else 'does NOT hold'} {BAR:g} on every probe" +
          ("" if ok else "; set model.encoder.head_precision / encoder_precision to the first policy above whose worst case is inside the bar"))


if __name__ == "__main__":
    main()
