#!/usr/bin/env python
"""Instruction census of the attention kernel's steady-state loop (one iteration pair = two 64-key tiles per wave), from the device listing
build.py keeps (-save-temps): every instruction class with its count per tile and the issue cost measured by tools/ubench/valu_rates.hip
(profiles/r02_a_valu_rates_ubench.txt, two waves per SIMD), summed against the MFMA pipe time of the tile.
    python tools/attn_isa_table.py [listing.s]"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "amodal-depth-anything_amd", "csrc"))
import isa_guard as G  # noqa: E402

# issue cycles per instruction with two waves per SIMD (profiles/r02_a_valu_rates_ubench.txt); MFMA = pipe cycles of the instruction
COST = {"v_exp_f32": 8.3, "v_cvt_pk_f16_f32": 4.6, "v_cvt_pkrtz_f16_f32": 4.6, "v_max3_f32": 4.6, "v_pk_mul_f32": 4.6, "v_pk_add_f32": 4.6, "v_pk_fma_f32": 4.6,
        "v_pk_mul_f16": 4.6, "v_mfma_f32_32x32x16_f16": 32.0, "v_mfma_f32_16x16x32_f16": 16.0}
DEFAULT_VALU, DEFAULT_LDS, DEFAULT_SALU = 2.6, 4.0, 1.0


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_") or op.startswith("global_"):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "sync"
    return "salu"


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "amodal-depth-anything_amd", "csrc", "build", "f16", "ada_attention-hip-amdgcn-amd-amdhsa-gfx950.s")
    body = [b for n, b in G.kernels(open(path).read()).items() if "attention_kernel_mix" in n][0]
    # the steady-state loop: the backward branch that encloses the most MFMAs
    labels, best = {}, None
    for idx, (no, ins) in enumerate(body):
        m = G.LABEL.match(ins)
        if m:
            labels[m.group(1)] = idx
            continue
        op = ins.split()[0]
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = ins.split()[-1]
            if tgt in labels:
                seg = body[labels[tgt]:idx + 1]
                nm = sum(1 for _, i in seg if i.startswith("v_mfma_f32_32x32"))
                if best is None or nm > best[0]:
                    best = (nm, seg)
    nm, seg = best
    tiles = nm / 16.0
    # rare paths carry ";;ADA_RARE_BEGIN" markers in the source (asm volatile comments; the scheduler moves the path's instructions around them, so
    # they do not bracket anything): a basic block of the listing -- between labels / branches -- that contains a marker is a rare block
    raw = open(path).read().splitlines()
    marks = [no for no, ln in enumerate(raw, 1) if "ADA_RARE_BEGIN" in ln]
    bounds = [no for no, ins in seg if G.LABEL.match(ins) or ins.split()[0].startswith("s_cbranch") or ins.split()[0] == "s_branch"]
    bounds = [seg[0][0] - 1] + bounds + [seg[-1][0] + 1]
    rare_lines = set()
    for lo, hi in zip(bounds, bounds[1:]):
        if any(lo < m < hi for m in marks):
            rare_lines.update(range(lo + 1, hi))
    counts, rare = collections.Counter(), collections.Counter()
    for no, ins in seg:
        if G.LABEL.match(ins):
            continue
        (rare if no in rare_lines else counts)[ins.split()[0].replace("_e32", "").replace("_e64", "")] += 1
    print(f"# instructions inside the marked rare paths (first / last tile, deferred rescale), not counted below: {sum(rare.values())} "
          f"({', '.join(f'{k} {v}' for k, v in rare.most_common(6))} ...)")
    print(f"# {os.path.basename(path)}: steady-state loop = {len(seg)} lines, {nm} v_mfma_f32_32x32x16 = {tiles:g} tiles of 64 keys x 32 queries per wave")
    print(f"{'instruction':34s} {'class':6s} {'per loop':>9s} {'per tile':>9s} {'cycles each':>12s} {'cycles / tile':>14s}")
    tot = collections.Counter()
    for op, c in sorted(counts.items(), key=lambda kv: (-kv[1], kv[0])):
        cl = classify(op)
        each = COST.get(op, {"valu": DEFAULT_VALU, "lds": DEFAULT_LDS, "salu": DEFAULT_SALU, "vmem": 4.0, "sync": 0.0, "mfma": 16.0}[cl])
        tot[cl] += c / tiles * each
        print(f"{op:34s} {cl:6s} {c:9d} {c / tiles:9.1f} {each:12.1f} {c / tiles * each:14.1f}")
    print("# issue cycles per tile and wave by class: " + ", ".join(f"{k} {v:.0f}" for k, v in tot.items()))
    print(f"# VALU + LDS issue per tile: {tot['valu'] + tot['lds']:.0f} cycles against {tot['mfma']:.0f} cycles of matrix-pipe time; two waves per SIMD share the issue port")


if __name__ == "__main__":
    main()
