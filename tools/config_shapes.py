#!/usr/bin/env python
"""Per-launch table of one forward of any configuration (default: BASELINE config 2, AmodalDAv2 ViT-B, 8 x 518^2): for every distinct ada_igemm launch the
tile the heuristic chose, its tile count and rounds on 256 CUs, us and TFLOP/s timed alone, and -- SWEEP=1 -- the time under every other tile
configuration (ada_debug_set_tile) and the 4-wave main loop; the other kernels (attention, LayerNorm, resizes, tail) with us and GB/s or TFLOP/s.
    ENCODER=vitb B=8 SWEEP=1 python tools/config_shapes.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hip_ext  # noqa: E402
from hip_ext import engine as E  # noqa: E402
from src.models import get_model  # noqa: E402
from src.util.synth_weights import centred_final_bias, fill_state_dict_, make_inputs  # noqa: E402

E.GRAPH_MODE = "0"
ENC, B, SWEEP, REPS = os.environ.get("ENCODER", "vitb"), int(os.environ.get("B", "8")), os.environ.get("SWEEP") == "1", int(os.environ.get("REPS", "10"))
RAW, SIZE = os.environ.get("RAW") == "1", int(os.environ.get("SIZE", "518"))     # RAW=1 SIZE=1022 ENCODER=vitg B=8: BASELINE config 5
TILE = {0: (256, 32, 2), 1: (128, 64, 3), 2: (256, 128, 1), 3: (256, 256, 1), 4: (128, 128, 2)}   # code -> (BM, BN, workgroups per CU)
NAMES = ("igemm", "attention", "layernorm", "patchify", "write_cls", "bilinear", "dpt_tail", "tapsum_resize")


def timeit(fn, reps=REPS):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    if RAW:
        from src.models.amodalsynthdrive.depth_anything_v2_raw.dpt import DepthAnythingV2 as Raw
        feats = {"vits": (64, [48, 96, 192, 384]), "vitb": (128, [96, 192, 384, 768]), "vitl": (256, [256, 512, 1024, 1024]), "vitg": (384, [1536] * 4)}[ENC]
        m = Raw(encoder=ENC, features=feats[0], out_channels=feats[1]).eval()
        fill_state_dict_(m.state_dict(), 0)
        m = m.cuda()
        x = torch.randn(B, 3, SIZE, SIZE, device="cuda")
        run = lambda: m(x)     # noqa: E731
    else:
        m = get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="entire_target_object", encoder=ENC, pretrained=False).eval()
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        fill_state_dict_(sd, 0)
        cb = centred_final_bias(ENC, ROOT)      # centred logits: the default (first-rung) path is what is timed
        if cb:
            sd[cb[0]] = torch.full_like(sd[cb[0]], cb[1])
        m.load_state_dict(sd)
        m = m.cuda()
        x, _, mask, obs = make_inputs(B, SIZE, SIZE, 0, device="cuda")
        run = lambda: m(x, guide_mask=mask, observation=obs)     # noqa: E731
    with torch.no_grad():
        run()
        whole = timeit(run, 5)
    calls = []
    real = {n: getattr(E, "k_" + n) for n in NAMES}
    for n in NAMES:
        setattr(E, "k_" + n, (lambda nn: (lambda *a, **k: (calls.append((nn, a, k)), real[nn](*a, **k))[1]))(n))
    with torch.no_grad():
        run()
    for n in NAMES:
        setattr(E, "k_" + n, real[n])
    torch.cuda.synchronize()
    groups = {}
    for name, a, k in calls:
        if name == "igemm":
            key = (name, k["M"], k["N"], k["K"], k.get("a_mode", 0), k.get("flags", 0), k.get("map_op", 0), k.get("out_f32") is not None, k.get("out_op") is not None,
                   k.get("a_wrap", 0), k.get("a_dup_seg", 0), k.get("tap_cols", 0))
        elif name == "layernorm":
            key = (name, a[2], a[3], k.get("out2_op") is not None, k.get("unshuffle_s", 0))
        elif name == "bilinear":
            key = (name,) + tuple(a[2:8])
        else:
            key = (name,)
        groups.setdefault(key, []).append((a, k))
    eng = (m if RAW else m.encoder)._engine()
    print(f"# precision ladder re-ran {eng.escalated} images during the warm-up / timing calls")
    print(f"# {'raw DepthAnythingV2' if RAW else 'AmodalDAv2'} {ENC}, {B} x {SIZE} x {SIZE}: whole forward {whole:.0f} us = {B / whole * 1e6:.1f} images/s; {len(calls)} launches, {len(groups)} distinct; launches timed alone ({REPS} reps, warm caches)")
    rows = []
    for key, lst in groups.items():
        name = key[0]
        a, k = lst[0]
        us = timeit(lambda: real[name](*a, **k))
        r = dict(kernel=name, n=len(lst), us=round(us, 1), total_us=round(us * len(lst), 1))
        if name == "igemm":
            code = hip_ext.debug_last_tile()
            bm, bn, occ = TILE[code % 100]
            tiles = -(-k["M"] // bm) * -(-k["N"] // bn)
            flop = 2.0 * k["M"] * k["N"] * (k.get("k_alg") or k["K"])
            r.update(M=k["M"], N=k["N"], K=k["K"], conv=k.get("a_mode", 0), flags=hex(k.get("flags", 0)), tile=f"{bm}x{bn}" + ("/4w" if code >= 200 else ""), tiles=tiles,
                     rounds=round(tiles / (256.0 * occ), 2), tflops=round(flop / us / 1e6, 1))
            if SWEEP:
                alt = {}
                for cfg in (0, 1, 2, 3, 4):
                    if (k.get("flags", 0) & hip_ext.EP_TAIL) or (cfg == 0 and k["N"] > 256):
                        continue
                    hip_ext.debug_set_tile(cfg)
                    try:
                        alt[f"{TILE[cfg][0]}x{TILE[cfg][1]}"] = round(timeit(lambda: real[name](*a, **k)), 1)
                    except Exception:
                        pass
                hip_ext.debug_set_tile(3)
                hip_ext.debug_set_variant(16)
                try:
                    alt["256x256/4w"] = round(timeit(lambda: real[name](*a, **k)), 1)
                except Exception:
                    pass
                hip_ext.debug_set_tile(-1)
                hip_ext.debug_set_variant(0)
                best = min(alt, key=alt.get)
                r.update(sweep=alt, best=best, gain_pct=round((us / alt[best] - 1) * 100, 1))
        elif name == "attention":
            r.update(B=a[2], N=a[3], heads=a[4], tflops=round(4.0 * a[2] * a[4] * 64 * a[3] ** 2 / us / 1e6, 1))
        elif name == "layernorm":
            r.update(rows=a[2], dim=a[3], gbps=round(a[2] * a[3] * (6 + (2 if key[3] else 0)) / us / 1e3, 0))
        rows.append(r)
    rows.sort(key=lambda r: -r["total_us"])
    tot = sum(r["total_us"] for r in rows)
    print(f"# sum of isolated launches {tot:.0f} us; igemm {sum(r['total_us'] for r in rows if r['kernel'] == 'igemm'):.0f}, attention {sum(r['total_us'] for r in rows if r['kernel'] == 'attention'):.0f}, "
          f"layernorm {sum(r['total_us'] for r in rows if r['kernel'] == 'layernorm'):.0f}")
    if SWEEP:
        print(f"# igemm with the best tile per shape: {sum(r['n'] * min(r['sweep'].values()) for r in rows if 'sweep' in r):.0f} us against {sum(r['total_us'] for r in rows if 'sweep' in r):.0f} with the heuristic")
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
