#!/usr/bin/env python
"""Board power and shader clock WHILE one kernel loops: each of the ViT-L bs=32 GEMM shapes and the attention kernel is launched back to back for
--seconds, and `rocm-smi --showpower --showclocks` is sampled once a second from a helper thread during the loop (not after it).
    python tools/power_loops.py [--seconds 12] [--forward]      (--forward: the whole ViT-L bs=32 forward of bench.py instead of single kernels)"""
import argparse
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hip_ext as H  # noqa: E402


def sample(stop, rows):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            pw = re.search(r"Power \(W\): ([0-9.]+)", txt)
            sc = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
            rows.append((time.time(), float(pw.group(1)) if pw else None, int(sc.group(1)) if sc else None))
        except Exception as e:  # noqa: BLE001
            rows.append((time.time(), None, str(e)))
        stop.wait(1.0)


def loop(name, fn, flop, seconds, chunk=200):
    fn()
    torch.cuda.synchronize()
    stop, rows = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, rows))
    t0 = time.time()
    th.start()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(chunk):
            fn()
        torch.cuda.synchronize()
        n += chunk
    dt = time.time() - t0
    stop.set()
    th.join()
    inside = [(p, c) for (t, p, c) in rows if t0 + 1.0 <= t <= t0 + dt - 0.5 and p is not None]
    pw = [p for p, _ in inside]
    ck = [c for _, c in inside if isinstance(c, int)]
    print(f"{name:28s} {dt / n * 1e6:8.1f} us/launch {flop / (dt / n) / 1e12:7.1f} TF/s | {len(inside)} samples inside the loop: "
          f"power {min(pw):.0f}-{max(pw):.0f} W (mean {sum(pw) / len(pw):.0f}), sclk {min(ck)}-{max(ck)} MHz" if pw else f"{name}: no samples", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--forward", action="store_true")
    a = ap.parse_args()
    if a.forward:
        from src.models import get_model
        from src.util.synth_weights import fill_state_dict_, make_inputs
        m = get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="entire_target_object", encoder="vitl", pretrained=False).eval()
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        fill_state_dict_(sd, 0)
        m.load_state_dict(sd)
        m = m.cuda()
        x, _, mask, obs = make_inputs(32, 518, 518, 100, device="cuda")

        def fwd():
            with torch.no_grad():
                m(x, guide_rgb=None, guide_mask=mask, observation=obs)
        loop("AmodalDAv2 ViT-L 32 x 518^2", fwd, 32 * 1389.65e9, a.seconds, chunk=10)
        return
    op = H.operand_dtype()
    dev = "cuda"
    torch.manual_seed(0)
    T = 43840
    for name, M, N, K, kw, mode in [("qkv bias->op", T, 3072, 1024, dict(flags=H.EP_BIAS), "op"),
                                    ("proj ls+res f32", T, 1024, 1024, dict(flags=H.EP_BIAS | H.EP_GAMMA | H.EP_RESIDUAL), "res"),
                                    ("fc1 gelu->op", T, 4096, 1024, dict(flags=H.EP_BIAS | H.EP_GELU), "op"),
                                    ("fc2 ls+res f32", T, 1024, 4096, dict(flags=H.EP_BIAS | H.EP_GAMMA | H.EP_RESIDUAL), "res")]:
        A = torch.randn(M, K, device=dev).to(op)
        W = (torch.randn(N, K, device=dev) * K ** -0.5).to(op)
        args = dict(M=M, N=N, K=K, A=A, lda=K, W=W, bias=torch.randn(N, device=dev), **kw)
        if mode == "op":
            args.update(out_op=torch.empty(M, N, dtype=op, device=dev), ldo_op=N)
        else:
            x = torch.zeros(M, N, device=dev)
            args.update(gamma=torch.rand(N, device=dev) * 1e-3, res=x, ldr=N, out_f32=x, ldo_f32=N)
        loop(name, lambda: H.igemm(**args), 2.0 * M * N * K, a.seconds)
    B, Nt, heads = 32, 1370, 16
    D = heads * 64
    qkv = torch.randn(B * Nt, 3 * D, device=dev)
    qkv[:, :D] *= 0.125 * 1.4426950408889634
    qkv = qkv.to(op)
    out = torch.empty(B * Nt, D, dtype=op, device=dev)
    loop("attention 32x16x1370", lambda: H.attention(qkv, out, B, Nt, heads), 4.0 * B * heads * 64 * Nt * Nt, a.seconds)


if __name__ == "__main__":
    main()
