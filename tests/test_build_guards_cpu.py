"""CPU: the build-time ISA guards (csrc/isa_guard.py, run by csrc/build.py) catch what they exist for, the shipped build passes them,
and the generated 4-wave GEMM loop checked into csrc/ is what its generator emits."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "amodal-depth-anything_amd", "csrc"))
import isa_guard as G  # noqa: E402

CSRC = os.path.join(ROOT, "amodal-depth-anything_amd", "csrc")

# the shape of the round-3 failure: a row fetched by inline asm is copied ABOVE the hand-counted wait (profiles/r04_a_*)
BAD = """
_Z4badkv:
	global_load_dwordx4 v[2:5], v[6:7], off
	global_load_dwordx4 v[10:13], v[14:15], off
	v_mov_b64_e32 v[96:97], v[4:5]
	s_waitcnt vmcnt(1)
	v_pk_mul_f32 v[20:21], v[96:97], v[96:97]
	s_endpgm
.Lfunc_end0:
"""
GOOD = """
_Z5goodkv:
.LBB0_1:
	global_load_dwordx4 v[2:5], v[6:7], off
	global_load_dwordx4 v[10:13], v[14:15], off
	s_waitcnt vmcnt(1)
	v_mov_b64_e32 v[96:97], v[4:5]
	global_load_dwordx4 v[2:5], v[6:7], off
	v_mul_f32_e32 v20, v96, v97
	s_waitcnt vmcnt(1)
	v_add_f32_e32 v21, v10, v11
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
	s_endpgm
.Lfunc_end1:
"""
# loop-carried: the load issued at the bottom of the loop is read at its top before any wait -- only visible around the back edge
CARRIED = """
_Z7carriedv:
	s_waitcnt vmcnt(0)
.LBB0_1:
	v_add_f32_e32 v20, v2, v3
	s_waitcnt vmcnt(0)
	global_load_dwordx4 v[2:5], v[6:7], off
	s_cbranch_scc1 .LBB0_1
	s_endpgm
.Lfunc_end2:
"""


def test_inflight_guard_flags_a_copy_above_the_wait():
    ks = G.kernels(BAD)
    rep = G.check_inflight(ks["_Z4badkv"])
    assert len(rep) == 1 and "v_mov_b64" in rep[0][1] and rep[0][2] == [4, 5]
    assert len(G.check_packed_f32(ks["_Z4badkv"])) == 1


def test_inflight_guard_passes_counted_waits_and_sees_back_edges():
    assert G.check_inflight(G.kernels(GOOD)["_Z5goodkv"]) == []
    rep = G.check_inflight(G.kernels(CARRIED)["_Z7carriedv"])
    assert len(rep) == 1 and "v_add_f32" in rep[0][1]


def test_agpr_guard():
    body = G.kernels("_Z1kv:\n\tv_mfma_f32_16x16x32_f16 a[0:3], v[0:3], v[4:7], a[0:3]\nLPIPE4_END_7:\n\tv_accvgpr_read_b32 v1, a0\n\tv_accvgpr_write_b32 a3, v9\n.Lfunc_end0:\n")["_Z1kv"]
    bad = G.check_agpr_after_loop(body)
    assert [ins.split()[0] for _, ins in bad] == ["v_accvgpr_write_b32"]


def test_shipped_build_passed_the_guards():
    """build.py keeps the device listings it checked (-save-temps=obj) next to the objects: re-run the guards on them when present."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ada_build", os.path.join(CSRC, "build.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    ran = 0
    for kind, files in B.ISA_GUARD.items():
        for src in files:
            asm = os.path.join(CSRC, "build", "f16", src.replace(".hip", f"-hip-amdgcn-amd-amdhsa-{B.ARCH}.s"))
            if os.path.exists(asm):
                B.isa_guard(src, asm, [kind])
                ran += 1
    assert ran or not os.path.isdir(os.path.join(CSRC, "build", "f16"))


def test_generated_pipe4_loop_is_up_to_date(tmp_path):
    """csrc/ada_igemm_pipe4.inc is generated code: it must be exactly what tools/gen_pipe4_asm.py emits (and list m0 among its clobbers)."""
    inc = os.path.join(CSRC, "ada_igemm_pipe4.inc")
    before = open(inc).read()
    try:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_pipe4_asm.py")], stdout=subprocess.DEVNULL)
        after = open(inc).read()
    finally:
        open(inc, "w").write(before)
    assert after == before
    assert '"m0"' in before.rsplit(":", 1)[1]
