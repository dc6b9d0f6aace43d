#!/usr/bin/env python
"""Per-workgroup phase timestamps of one igemm launch (debug hook ada_debug_set_timestamps): entry, first slab landed,
main loop done, epilogue done (s_memtime ticks) + XCC / HW id.  Prints phase statistics and the round structure."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hip_ext as H  # noqa: E402

op = H.operand_dtype()
lib = H.load()
lib.ada_debug_set_timestamps.argtypes = [ctypes.c_void_p]
T = 43840
cases = {"fc1_op": (T, 4096, 1024, "op"), "fc1_op_pad": (T, 4096, 1024, "op_pad"), "fc1_pad": (T, 4096, 1024, "gelu_pad"), "qkv_pad": (T, 3072, 1024, "op_pad"), "proj_f32": (T, 1024, 1024, "f32"), "proj_op": (T, 1024, 1024, "op"), "few64": (16384, 256, 1024, "op"), "few64res": (16384, 256, 1024, "f32res"), "few256": (16384, 1024, 1024, "op"), "few256res": (16384, 1024, 1024, "f32res"), "proj": (T, 1024, 1024, "f32res"), "qkv": (T, 3072, 1024, "op"), "fc1": (T, 4096, 1024, "gelu"), "fc2": (T, 1024, 4096, "f32res"), "big": (8192, 8192, 8192, "f32")}
for name in sys.argv[1:] or ["proj"]:
    M, N, K, mode = cases[name]
    A = torch.randn(M, K, device="cuda").to(op)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(op)
    bias = torch.randn(N, device="cuda")
    args = dict(M=M, N=N, K=K, A=A, lda=K, W=W, bias=bias, flags=H.EP_BIAS)
    if mode == "op":
        args.update(out_op=torch.empty(M, N, dtype=op, device="cuda"), ldo_op=N)
    elif mode in ("op_pad", "gelu_pad"):
        args.update(out_op=torch.empty(M, N + 64, dtype=op, device="cuda"), ldo_op=N + 64, flags=H.EP_BIAS | (H.EP_GELU if mode == "gelu_pad" else 0))
    elif mode == "gelu":
        args.update(out_op=torch.empty(M, N, dtype=op, device="cuda"), ldo_op=N, flags=H.EP_BIAS | H.EP_GELU)
    elif mode == "f32res":
        x = torch.randn(M, N, device="cuda")
        args.update(gamma=torch.rand(N, device="cuda"), res=x, ldr=N, out_f32=x, ldo_f32=N, flags=H.EP_BIAS | H.EP_GAMMA | H.EP_RESIDUAL)
    else:
        args.update(out_f32=torch.empty(M, N, device="cuda"), ldo_f32=N)
    nblk = ((M + 255) // 256) * ((N + 255) // 256)
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        H.igemm(**args)
    torch.cuda.synchronize()
    lib.ada_debug_set_timestamps(buf.data_ptr())
    H.igemm(**args)
    torch.cuda.synchronize()
    lib.ada_debug_set_timestamps(None)
    d = buf.cpu().reshape(nblk, 8)
    t0 = int(d[:, 0].min())
    ent, first, loop, end = [(d[:, i] - t0).double() for i in range(4)]
    xcc = (d[:, 4] >> 32) & 0xF
    hw = d[:, 4] & 0xFFFFFFFF
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7  # informational
    # s_memtime counters are per XCD: measure the launch span inside each XCD and average
    key = (d[:, 4] >> 32) * 65536 + (d[:, 4] & 0xFF00)   # (XCC_ID, se/sh/cu of HW_ID): one compute unit
    spans = []
    for kv in key.unique():
        sel = key == kv
        spans.append(float(d[sel, 3].max() - d[sel, 0].min()))
    spans.sort()
    spans = spans[len(spans) // 4: -(len(spans) // 4) or None]   # inter-quartile: counters of different CUs are not comparable
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        H.igemm(**args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    span = sum(spans) / len(spans)
    print(f"== {name}: M={M} N={N} K={K}  blocks={nblk}  span {span:.0f} ticks per CU, {us:.1f} us per launch -> shader clock ~{span / us / 1e3:.2f} GHz")
    print(f"   prologue (entry->first slab): mean {float((first - ent).mean()):8.0f}  max {float((first - ent).max()):8.0f}")
    print(f"   main loop                  : mean {float((loop - first).mean()):8.0f}  min {float((loop - first).min()):8.0f} max {float((loop - first).max()):8.0f}  per k-step {float((loop - first).mean()) / (K // 64 - 0):.0f}")
    print(f"   epilogue                   : mean {float((end - loop).mean()):8.0f}  min {float((end - loop).min()):8.0f} max {float((end - loop).max()):8.0f}")
    print(f"   epilogue issue / store drain: mean {float((d[:, 5] - d[:, 2]).double().mean()):8.0f} / {float((d[:, 3] - d[:, 5]).double().mean()):8.0f}")
    if mode in ("f32res", "f32"):
        print(f"   f32 epilogue phases (overwrite the main-loop wait slots): dump->LDS landed {float(d[:, 6].double().mean()):8.0f}   vmcnt(0) wait {float(d[:, 7].double().mean()):8.0f}")
    print(f"   wave0 waits in main loop   : vmcnt mean {float(d[:, 6].double().mean()):8.0f}  barrier mean {float(d[:, 7].double().mean()):8.0f}  (per k-step {float(d[:, 6].double().mean()) / (K // 64):.0f} / {float(d[:, 7].double().mean()) / (K // 64):.0f})")
    order = torch.argsort(ent)
    # round structure: entry time histogram
    e = ent[order]
    import numpy as np
    hist, edges = np.histogram(e.numpy(), bins=12)
    print("   entry-time histogram:", list(hist), "bin width", int(edges[1] - edges[0]))
    print("   first 6 blocks by entry:", [(int(ent[i]), int(first[i]), int(loop[i]), int(end[i]), int(xcc[i]), int(cu[i])) for i in order[:6]])
    print("   last 3 blocks by end  :", [(int(ent[i]), int(first[i]), int(loop[i]), int(end[i]), int(xcc[i])) for i in torch.argsort(end)[-3:]])
