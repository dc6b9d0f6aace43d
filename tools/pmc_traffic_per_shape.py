#!/usr/bin/env python
"""Fabric traffic of every ada_igemm launch of one ViT-L bs=32 forward against its ALGORITHMIC bytes (operands once + outputs once + residual once): where
the 1.6x of `roofline.traffic` comes from.  Two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE; separate, --kernel-trace only) over a child that
runs the forward a few times; the parent (never touches the GPU before the children are done) then records the launch list of one forward and matches the
igemm dispatches of the LAST forward of each pass to it by order.

    python tools/pmc_traffic_per_shape.py            (parent)        python tools/pmc_traffic_per_shape.py --child N   (what the profiler runs)"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
ENC, B = os.environ.get("ENCODER", "vitl"), int(os.environ.get("B", "32"))


def build():
    import torch
    from src.models import get_model
    from src.util.synth_weights import centred_final_bias, fill_state_dict_, make_inputs
    m = get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="entire_target_object", encoder=ENC, pretrained=False).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    fill_state_dict_(sd, 0)
    cb = centred_final_bias(ENC, ROOT)
    if cb:
        sd[cb[0]] = torch.full_like(sd[cb[0]], cb[1])
    m.load_state_dict(sd)
    m = m.cuda()
    x, _, mask, obs = make_inputs(B, 518, 518, 0, device="cuda")
    return m, (lambda: m(x, guide_mask=mask, observation=obs))


def child(n):
    import torch
    m, run = build()
    with torch.no_grad():
        for _ in range(n):
            run()
    torch.cuda.synchronize()


def igemm_dispatches(path, counter):
    disp = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "igemm_kernel" not in r["Kernel_Name"]:
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), 0.0)
        disp[int(r["Dispatch_Id"])] = d + float(r["Counter_Value"])
    return [disp[k] for k in sorted(disp)]


def main():
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tmp = tempfile.mkdtemp(prefix="ada_pmc_shape_")
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = os.path.join(tmp, counter)
        subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "run", "--", sys.executable, os.path.abspath(__file__), "--child", "2"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), check=True)
        vals[counter] = igemm_dispatches(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0], counter)
    shutil.rmtree(tmp, ignore_errors=True)
    import torch
    from hip_ext import engine as E
    E.GRAPH_MODE = "0"
    m, run = build()
    with torch.no_grad():
        run()
    calls = []
    real = E.k_igemm
    E.k_igemm = lambda **k: (calls.append(k), real(**k))[1]
    with torch.no_grad():
        run()
    E.k_igemm = real
    n = len(calls)
    f, w = vals["FETCH_SIZE"][-n:], vals["WRITE_SIZE"][-n:]
    assert len(f) == n and len(w) == n, (len(vals["FETCH_SIZE"]), n)
    groups = {}
    for k, fk, wk in zip(calls, f, w):
        conv = k.get("a_mode", 0)
        a_rows = k["A"].numel() * 2 if conv else k["M"] * min(k["lda"], k["K"]) * 2            # conv: the padded NHWC operand tensor once
        alg = a_rows + k["W"].numel() * 2
        if k.get("out_f32") is not None:
            alg += k["M"] * k["N"] * 4
        if k.get("out_op") is not None:
            alg += k["M"] * k["N"] * 2 * (2 if k.get("split_seg") else 1)
        if k.get("res") is not None:
            alg += k["M"] * k["N"] * 4
        key = (k["M"], k["N"], k["K"], conv, hex(k.get("flags", 0)), k.get("out_f32") is not None, k.get("out_op") is not None)
        g = groups.setdefault(key, [0, 0.0, 0.0, alg])
        g[0] += 1
        g[1] += 2.0 * fk * 1024.0        # KiB; gfx950 tallies 128-byte read requests as 64 B (MI355X_MICROARCH.md)
        g[2] += wk * 1024.0
    rows = sorted(groups.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))
    tot_m = sum(g[1] + g[2] for _, g in rows)
    tot_a = sum(g[0] * g[3] for _, g in rows)
    print(f"# AmodalDAv2 {ENC} bs={B}: {n} igemm launches per forward, fabric traffic {tot_m / 1e9:.1f} GB = {tot_m / n / 1e6:.0f} MB per launch against {tot_a / 1e9:.1f} GB algorithmic ({tot_m / tot_a:.2f}x)")
    print(f"{'n':>3s} {'M':>7s} {'N':>5s} {'K':>6s} conv {'flags':>6s} | {'fetch MB':>9s} {'write MB':>9s} {'alg MB':>8s} {'ratio':>6s} {'excess MB x n':>13s}")
    for (M, N, K, conv, flags, of, oo), (cnt, fe, wr, alg) in rows:
        print(f"{cnt:3d} {M:7d} {N:5d} {K:6d} {conv:4d} {flags:>6s} | {fe / cnt / 1e6:9.0f} {wr / cnt / 1e6:9.0f} {alg / 1e6:8.0f} {(fe + wr) / cnt / alg:6.2f} {(fe + wr - cnt * alg) / 1e6:13.0f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        main()
