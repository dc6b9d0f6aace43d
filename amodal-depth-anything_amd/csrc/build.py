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
sys.path.insert(0, HERE)     # isa_guard.py lives next to this file: the package builds without the repository's tools/
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
# (-Wno-inline-asm: the generated loop lists m0 among its clobbers -- it rewrites m0 for its LDS-DMA copies and the backend's merging of
# identical m0 initialisations must see that -- and clang warns about every reserved register in a clobber list.)
PER_FILE_FLAGS = {"ada_tail.hip": ["-fno-slp-vectorize"], "ada_igemm.hip": ["-fno-slp-vectorize", "-Wno-inline-asm"]}
# ada_igemm.hip joined in round 4: a by-reference lambda capture of the k-walk counters put them into scratch behind the loop's "memory"-clobbering
# waits -- 12 bytes reloaded every k-step of every GEMM, silently, for most of a round.  No kernel of these files may use scratch or spill VGPRs.
NO_SCRATCH = {"ada_tail.hip", "ada_igemm.hip"}
# Round 4: the wrong results above were root-caused (profiles/r04_a_tail_inflight_register_root_cause.txt) -- NOT a hardware hazard of packed
# fp32 beside MFMAs (tools/ubench/pk_f32_beside_mfma.hip: 0 mismatches) but the compiler copying registers that the kernel's inline-asm
# fetches were still writing: with SLP on, the allocator parks a source row in other registers with v_mov_b64 placed ABOVE the hand-counted
# s_waitcnt.  The flag only happens to avoid that allocation, so the build now CHECKS the generated ISA (isa_guard.py, next to this file) and fails if
#   * any instruction of an ISA_GUARD["inflight"] file touches a VGPR while a load into it may still be outstanding,
#   * an ISA_GUARD["no_packed_f32"] file contains v_pk_*_f32 (the configuration the kernel was validated in),
#   * in an ISA_GUARD["agpr_after_pipe4"] file anything but v_accvgpr_read touches an AGPR between the end of the generated 4-wave GEMM
#     loop and the end of the kernel (the accumulators live in a[0:255] there and the compiler only knows them as clobbered).
# (ada_attention.hip has the same construction on the LDS counter -- inline-asm ds_read into C++ variables, "+v"-tied lgkmcnt waits -- and is held to the same rule)
ISA_GUARD = {"inflight": {"ada_tail.hip", "ada_attention.hip"}, "no_packed_f32": {"ada_tail.hip"}, "agpr_after_pipe4": {"ada_igemm.hip"}}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _digest(defines):
    h = hashlib.sha256()
    for name in SOURCES + HEADERS + ["build.py", "isa_guard.py"]:   # build.py itself: flags are part of the digest
        with open(os.path.join(HERE, name), "rb") as f:
            h.update(f.read())
    h.update((" ".join(defines) + os.environ.get("ADA_EXTRA_FLAGS", "")).encode())
    return h.hexdigest()


def lib_path(bf16=False, tag=None):
    return os.path.join(HERE, f"libada_hip_{tag}.so" if tag else "libada_hip_bf16.so" if bf16 else "libada_hip.so")


def isa_guard(src, asm_path, kinds):
    """Checks the device assembly the object was assembled from (-save-temps=obj: the very listing, not a re-compile)."""
    import isa_guard as G
    if not os.path.exists(asm_path):
        raise RuntimeError(f"{src}: device assembly {asm_path} not found -- cannot run the ISA guards")
    problems = []
    nk = npipe4 = 0
    for name, body in G.kernels(open(asm_path).read()).items():
        if not name.startswith("_Z"):
            continue
        nk += 1
        if "inflight" in kinds:
            problems += [f"{name} line {no}: `{ins}` touches v{regs} while the load at line {own} may still be writing it"
                         for no, ins, regs, own in G.check_inflight(body)]
        if "no_packed_f32" in kinds:
            problems += [f"{name} line {no}: packed fp32 VALU op `{ins}`" for no, ins in G.check_packed_f32(body)]
        if "agpr_after_pipe4" in kinds:
            found = G.check_agpr_after_loop(body)
            npipe4 += found is not None
            problems += [f"{name} line {no}: `{ins}` touches an AGPR after the generated main loop" for no, ins in (found or [])]
    if not nk or ("agpr_after_pipe4" in kinds and not npipe4):
        raise RuntimeError(f"{src}: no kernel{' with the generated 4-wave loop' if nk else ''} found in {asm_path}")
    if problems:
        raise RuntimeError(f"{src}: ISA guard failed ({len(problems)} finding(s), listing {asm_path}):\n  " + "\n  ".join(problems[:20]))


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
        guarded = [kind for kind, files in ISA_GUARD.items() if src in files]
        cmd = common + PER_FILE_FLAGS.get(src, []) + (["-save-temps=obj"] if guarded else []) + ["-c", os.path.join(HERE, src), "-o", obj]
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
        if guarded:
            isa_guard(src, os.path.join(objdir, src.replace(".hip", f"-hip-amdgcn-amd-amdhsa-{ARCH}.s")), guarded)
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
