"""Builds libada_hip.so (gfx950) in-tree with plain hipcc -- no cmake, no torch extension machinery.

    python build.py [--bf16] [--force]

The .so lands next to this file so it travels to the GPU box with the repo snapshot.  A second
library built with -DADA_OPERAND_BF16 (libada_hip_bf16.so) is produced with --bf16; it exists only
to *measure* the bf16-operand variant the north star names against the fp16 default.
"""
import argparse
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["ada_api.hip", "ada_igemm.hip", "ada_attention.hip", "ada_elementwise.hip", "ada_pipeline.hip", "ada_eval.hip", "ada_tail.hip"]
HEADERS = ["ada_common.h", "ada_igemm_pipe4.inc", os.path.join("..", "..", "include", "ada_hip.h")]
ARCH = "gfx950"
# ada_tail.hip: no SLP vectorisation.  Vectorised, the producers' interpolation becomes v_pk_mul_f32 / v_pk_fma_f32 on register pairs
# gathered with v_mov, and with the other wave of the SIMD issuing MFMAs single elements of the interpolated tile came out wrong (lanes
# 48-63, reproducible, gone when either wave sleeps or when the packed ops are off) -- profiles/r03_p_fused_tail.txt.
# ada_igemm.hip: also without SLP.  Packed fp32 VALU ops (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, which is what SLP turns the epilogues'
# adjacent scalar adds / muls into) cost more than the two scalar ops they replace when MFMAs are issuing beside them
# (MI355X_MICROARCH.md): +0.4 % end to end on three interleaved pairs, outputs bit-identical (profiles/r03_w_no_slp_ab.txt).  For the other
# files (attention, LayerNorm, resizes) the vectoriser helps or is neutral -- attention is 1 % slower without it.
PER_FILE_FLAGS = {"ada_tail.hip": ["-fno-slp-vectorize"], "ada_igemm.hip": ["-fno-slp-vectorize"]}
NO_SCRATCH = {"ada_tail.hip"}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _digest(defines):
    h = hashlib.sha256()
    for name in SOURCES + HEADERS + ["build.py"]:   # build.py itself: flags are part of the digest
        with open(os.path.join(HERE, name), "rb") as f:
            h.update(f.read())
    h.update((" ".join(defines) + os.environ.get("ADA_EXTRA_FLAGS", "")).encode())
    return h.hexdigest()


def lib_path(bf16=False, tag=None):
    return os.path.join(HERE, f"libada_hip_{tag}.so" if tag else "libada_hip_bf16.so" if bf16 else "libada_hip.so")


def build(bf16=False, force=False, verbose=True, tag=None, extra_defines=()):
    """tag + extra_defines: experiment builds (libada_hip_<tag>.so with extra -D flags; select with ADA_HIP_LIB)."""
    defines = (["-DADA_OPERAND_BF16"] if bf16 else []) + list(extra_defines)
    out = lib_path(bf16, tag)
    stamp = out + ".stamp"
    digest = _digest(defines)
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return out
    objdir = os.path.join(HERE, "build", tag or ("bf16" if bf16 else "f16"))
    os.makedirs(objdir, exist_ok=True)
    cc = _hipcc()
    common = [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
              # the fused epilogues are fully unrolled over the accumulator registers; without this the
              # unroller gives up and the accumulators spill to scratch (cdna_hip_programming.md rule 20)
              "-mllvm", "-pragma-unroll-threshold=200000"] + defines + os.environ.get("ADA_EXTRA_FLAGS", "").split()   # ADA_EXTRA_FLAGS: A/B builds (with --tag)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = common + PER_FILE_FLAGS.get(src, []) + ["-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        if src in NO_SCRATCH:
            # kernels that keep in-flight loads in registers managed by hand (inline-asm fetch + hand-counted s_waitcnt): a spill or a
            # register copy inserted by the compiler between the fetch and its wait would save / move stale data -- refuse such a build
            res = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True)
            if res.returncode != 0:
                sys.stderr.write(res.stderr)
                raise subprocess.CalledProcessError(res.returncode, cmd)
            usage = [ln for ln in res.stderr.splitlines() if "ScratchSize" in ln or "VGPRs Spill" in ln]
            bad = [ln for ln in usage if not ln.rstrip().endswith(": 0 [-Rpass-analysis=kernel-resource-usage]")]
            if not usage or bad:
                raise RuntimeError(f"{src}: the kernel must not use scratch / spill registers:\n" + "\n".join(bad or ["no resource-usage remarks found"]))
        else:
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    link = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    with open(stamp, "w") as f:
        f.write(digest)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--tag", default=None)
    ap.add_argument("-D", dest="defs", action="append", default=[])
    a = ap.parse_args()
    print(build(bf16=a.bf16, force=a.force, tag=a.tag, extra_defines=["-D" + d for d in a.defs]))
