#!/usr/bin/env python
"""For every distinct igemm launch of one ViT-L bs=32 forward, time each applicable tile configuration in isolation
(debug hook ada_debug_set_tile) and print the best one next to the heuristic's choice."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch
import hip_ext
from hip_ext import engine as E
from src.models import get_model
from src.util.synth_weights import fill_state_dict_, make_inputs
B = int(os.environ.get("B", 32))
lib = hip_ext.load()
m = get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="x", encoder="vitl", pretrained=False).eval()
sd = {k: v.clone() for k, v in m.state_dict().items()}; fill_state_dict_(sd, 0); m.load_state_dict(sd); m = m.cuda()
x, _, mask, obs = make_inputs(B, 518, 518, 0, device="cuda")
with torch.no_grad():
    m(x, guide_mask=mask, observation=obs)
calls = []
real = E.k_igemm
E.k_igemm = lambda **k: (calls.append(k), real(**k))[1]
with torch.no_grad():
    m(x, guide_mask=mask, observation=obs)
E.k_igemm = real
torch.cuda.synchronize()
seen = {}
for k in calls:
    key = (k["M"], k["N"], k["K"], k.get("a_mode", 0), k.get("flags", 0), k.get("map_op", 0), k.get("out_f32") is not None, k.get("out_op") is not None)
    seen.setdefault(key, [0, k])[0] += 1


def timeit(k, reps=5):
    for _ in range(2): real(**k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): real(**k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot_h = tot_b = 0.0
for key, (cnt, k) in sorted(seen.items(), key=lambda kv: -kv[1][0] * kv[0][0] * kv[0][1] * kv[0][2]):
    if k["N"] <= 128 or (k.get("flags", 0) & hip_ext.EP_TAIL):
        continue
    lib.ada_debug_set_tile(-1)
    th = timeit(k)
    res = {}
    for cfg in (3, 2, 4, 5, 1):
        lib.ada_debug_set_tile(cfg)
        try:
            res[cfg] = timeit(k)
        except Exception as e:
            res[cfg] = float("nan")
    lib.ada_debug_set_tile(-1)
    best = min(res, key=lambda c: res[c])
    tot_h += cnt * th; tot_b += cnt * res[best]
    print(f"x{cnt:3d} M={k['M']:8d} N={k['N']:5d} K={k['K']:5d} conv={k.get('a_mode', 0)} flags={k.get('flags', 0):#x}: heuristic {th * 1e3:7.1f} us | " +
          " ".join(f"c{c}:{t * 1e3:7.1f}" for c, t in res.items()) + f" | best c{best} ({(th / res[best] - 1) * 100:+.0f}%)")
print(f"total over N>128 igemm launches: heuristic {tot_h:.2f} ms, per-shape best {tot_b:.2f} ms")
