#!/usr/bin/env python
"""Generates amodal-depth-anything_amd/csrc/ada_igemm_pipe4.inc: the hand-scheduled main loop of the 4-wave 256x256x64 tile of ada_igemm
(one wave per SIMD, wave tile 128 x 128, 256 accumulators in AGPRs) as ONE inline-asm statement with fixed registers.

hipcc cannot be made to produce this stream from C++ (round 2: the compiler-scheduled tile was 30-40 % slower; round 3: with inline-asm reads
and sched_barriers it spilled fragments to scratch and wrapped every LDS-DMA copy in a waterfall loop), so the loop is written as assembly
text -- by this script, so that the interleave (which read / copy goes behind which MFMA) is a table, not 600 hand-typed lines.

Register plan (physical; all listed as clobbers, the compiler keeps its own values elsewhere):
    a[0:255]     accumulators: sub-tile (16-row block ab, 16-column block bb) -> a[16 * (4 * (ab >> 1) + (bb >> 1)) + 4 * (2 * (ab & 1) + (bb & 1)) ...]
                 i.e. the acc[i][j][2a+b] order of the C++ epilogue
    v[0:63]      fragment set 0 (k half 0): A blocks 0-7 = v[0:31], B blocks 0-7 = v[32:63]
    v[64:127]    fragment set 1 (k half 1)
    v[128:135]   LDS read addresses: A (stage 0 set 0, stage 0 set 1, stage 1 set 0, stage 1 set 1), B likewise
    s[84:93]     loop scalars
k-tile t (stage S = t & 1), fragment set 0 of k-tile t already in registers:
    phase 1a  MFMAs 0-31 on set 0   ||  the 16 ds_read_b128 of set 1 (stage S), one per MFMA from the start
              s_waitcnt lgkmcnt(0), s_barrier        [every wave is done reading stage S: it is free]
    phase 1b  MFMAs 32-63 on set 0  ||  the first LDS-DMA copies of k-tile t+2 into stage S (one per 6 MFMAs)
              s_waitcnt vmcnt(n issued so far), s_barrier   [k-tile t+1 (copied one iteration ago) has landed for every wave]
    phase 2   64 MFMAs on set 1     ||  the 16 ds_read_b128 of set 0 of k-tile t+1 (stage S^1)  ||  the rest of the copies of k-tile t+2
              s_waitcnt lgkmcnt(0)
(v1 of this schedule -- one barrier per k-tile, copies behind it with vmcnt(0) -- measured 5 % slower than the 8-wave loop on the K = 1024
shapes and 27 % slower at 8192^3: the copies had only ~1.5 phases to land.  Here they have 2.5.)
Copies of k-tiles past the end use a scalar offset beyond the buffer's num_records: the hardware zero-fills without fetching.
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "amodal-depth-anything_amd", "csrc", "ada_igemm_pipe4.inc")

RB = 128            # bytes per LDS row
STAGE = 0x10000     # bytes per stage (A 32 KB + B 32 KB)
PIECE = 0x1000      # bytes per LDS-DMA instruction of a 256-thread workgroup

# where the 16 reads / 16 copies of a phase go: index of the MFMA (0-63) they are issued BEHIND
READ_STRIDE = int(os.environ.get("PIPE4_READ_STRIDE", "1"))
READ_SLOTS = [READ_STRIDE * k for k in range(16)]   # one read behind every READ_STRIDE-th MFMA from the start of a phase
MID = 32                                     # the mid-phase synchronisation point sits in front of MFMA 32 of phase 1
# the 16 copies of a k-tile are spread over the 96 MFMAs between the mid-phase barrier and the end of the k-tile (global MFMA index 32 .. 127,
# phase 2 = 64 ..): one every 6 MFMAs -- an LDS-DMA instruction costs the issuing wave ~60-180 cycles (MI355X_MICROARCH.md), and with one wave per
# SIMD nobody else issues MFMAs meanwhile, so the copies must not bunch up
COPY_START, COPY_STRIDE = int(os.environ.get("PIPE4_COPY_START", "34")), int(os.environ.get("PIPE4_COPY_STRIDE", "6"))
COPY_GLOBAL = [COPY_START + COPY_STRIDE * k for k in range(16)]
assert COPY_GLOBAL[0] >= 33 and COPY_GLOBAL[-1] <= 126 and READ_SLOTS[-1] < 32
COPIES_BEFORE_WAIT = sum(1 for g in COPY_GLOBAL if g < 64)   # issued before the vmcnt wait at the end of phase 1


def areg(ab, bb):
    i, a, j, b = ab >> 1, ab & 1, bb >> 1, bb & 1
    return 16 * (4 * i + j) + 4 * (2 * a + b)


def frag(set_, kind, blk):          # kind 0 = A, 1 = B
    base = 64 * set_ + 32 * kind + 4 * blk
    return f"v[{base}:{base + 3}]"


def read_instr(set_, stage, idx):   # idx 0-7 A blocks, 8-15 B blocks; order: B0-3, A0, B4-7, A1..A7 (what the next phase's first MFMAs need first)
    order = [8, 9, 10, 11, 0, 12, 13, 14, 15, 1, 2, 3, 4, 5, 6, 7]
    r = order[idx]
    kind, blk = (0, r) if r < 8 else (1, r - 8)
    addr = 128 + 4 * kind + 2 * stage + set_
    return f"ds_read_b128 {frag(set_, kind, blk)}, v{addr} offset:{blk * 16 * RB}"


def copy_instr(idx, soff_a, soff_b):   # idx 0-7 A passes, 8-15 B passes
    if idx < 8:
        return f"buffer_load_dwordx4 %[ao{idx}], %[ra], {soff_a} offen lds"
    return f"buffer_load_dwordx4 %[bo{idx - 8}], %[rb], {soff_b} offen lds"


def mfma(set_, n, zero_c=False):    # n-th MFMA of a phase: A block n >> 3, B block n & 7; zero_c: the very first k half starts from C = 0
    ab, bb = n >> 3, n & 7
    c = areg(ab, bb)
    src_c = "0" if zero_c else f"a[{c}:{c + 3}]"
    return f'"v_mfma_f32_16x16x32_" ADA_MFMA_SUFFIX " a[{c}:{c + 3}], {frag(set_, 0, ab)}, {frag(set_, 1, bb)}, {src_c}\\n"'


def q(line):
    return f'"{line}\\n"'


def phase(set_, g0, reads=None, copies=None, mid=None, zero_c=False):
    """64 MFMAs on fragment set `set_` (global MFMA indices g0 .. g0 + 63 of the k-tile); reads: (set, stage) of the 16 fragment reads to
    interleave; copies: (soffA, soffB) or None; mid: instructions placed in front of MFMA number MID."""
    out = []
    ri = 0
    for n in range(64):
        if mid is not None and n == MID:
            out += mid
        out.append(mfma(set_, n, zero_c))
        if copies is not None and (g0 + n) in COPY_GLOBAL:
            # m0 was bumped right after the previous copy (at least one instruction ago: the SALU-writes-M0 hazard needs one wait state)
            out.append(q(copy_instr(COPY_GLOBAL.index(g0 + n), *copies)))
            out.append(q(f"s_add_u32 m0, m0, {PIECE}"))
        if reads is not None and ri < 16 and READ_SLOTS[ri] == n:
            out.append(q(read_instr(reads[0], reads[1], ri)))
            ri += 1
    assert reads is None or ri == 16
    return out


def k_tile(stage, first=False):
    S = stage
    o = []
    o.append(q(f"; ---- k-tile on stage {S}" + (" (k-tile 0: accumulators start from C = 0)" if first else "")))
    # has2 = (t + 2 < nk): scalar offsets of this iteration's copies (tile t+2), out of bounds past the end
    o += [q("s_add_u32 s88, s84, 2"), q("s_cmp_lt_u32 s88, %[nk]"), q("s_cselect_b32 s92, s85, s91"), q("s_cselect_b32 s93, s86, s91")]
    mid = [q("s_waitcnt lgkmcnt(0)"), q("s_barrier"), q("s_mov_b32 m0, %[m0s0]" if S == 0 else "s_mov_b32 m0, s90")]
    o += phase(0, 0, reads=(1, S), copies=("s92", "s93"), mid=mid, zero_c=first)
    o += [q(f"s_waitcnt vmcnt({COPIES_BEFORE_WAIT})"), q("s_barrier")]
    o += phase(1, 64, reads=(0, S ^ 1), copies=("s92", "s93"))
    o.append(q("s_waitcnt lgkmcnt(0)"))
    # offsets of the next k-tile to copy: +128 B along k; a 3x3 conv jumps to the next input row every `period` k-tiles
    o += [q("s_add_u32 s85, s85, 128"), q("s_add_u32 s86, s86, 128"), q("s_add_u32 s87, s87, 1"), q("s_cmp_eq_u32 s87, %[period]"),
          q("s_cselect_b32 s88, %[jump], 0"), q("s_cselect_b32 s87, 0, s87"), q("s_add_u32 s85, s85, s88")]
    return o


def main():
    L = []
    L.append(q("; ==== ada_igemm 4-wave pipelined main loop (generated by tools/gen_pipe4_asm.py) ===="))
    L += [q("s_mov_b32 s91, 0x7fffffff"), q(f"s_add_u32 s90, %[m0s0], {STAGE}")]
    L += [q("v_mov_b32 v128, %[abase]"), q("v_xor_b32 v129, 64, %[abase]"), q(f"v_add_u32 v130, {STAGE}, %[abase]"), q("s_nop 0"),
          q(f"v_add_u32 v131, {STAGE}, v129"),
          q("v_mov_b32 v132, %[bbase]"), q("v_xor_b32 v133, 64, %[bbase]"), q(f"v_add_u32 v134, {STAGE}, %[bbase]"), q("s_nop 0"),
          q(f"v_add_u32 v135, {STAGE}, v133")]
    # prologue: k-tile 0 -> stage 0, k-tile 1 -> stage 1
    L.append(q("s_mov_b32 m0, %[m0s0]"))
    L.append(q("s_nop 0"))
    for idx in range(16):
        L.append(q(copy_instr(idx, "%[sa0]", "0")))
        L.append(q(f"s_add_u32 m0, m0, {PIECE}"))
        L.append(q("s_nop 0"))
    L.append(q("s_mov_b32 m0, s90"))
    L.append(q("s_nop 0"))
    for idx in range(16):
        L.append(q(copy_instr(idx, "%[sa1]", "%[sb1]")))
        L.append(q(f"s_add_u32 m0, m0, {PIECE}"))
        L.append(q("s_nop 0"))
    L += [q("s_waitcnt vmcnt(16)"), q("s_barrier")]
    for idx in range(16):
        L.append(q(read_instr(0, 0, idx)))
    L += [q("s_mov_b32 s84, 0"), q("s_mov_b32 s85, %[sa2]"), q("s_mov_b32 s86, 256"), q("s_mov_b32 s87, %[cnt]"), q("s_waitcnt lgkmcnt(0)")]
    # k-tile 0 is peeled: its first 64 MFMAs take C = 0 (no 256 v_accvgpr_write to zero the accumulators), then it joins the loop
    L += k_tile(0, first=True)
    L += [q("s_add_u32 s84, s84, 1"), q("s_cmp_ge_u32 s84, %[nk]"), q("s_cbranch_scc1 LPIPE4_END_%="), q("s_branch LPIPE4_ODD_%=")]
    L.append(q("LPIPE4_%=:"))
    L += k_tile(0)
    L += [q("s_add_u32 s84, s84, 1"), q("s_cmp_ge_u32 s84, %[nk]"), q("s_cbranch_scc1 LPIPE4_END_%=")]
    L.append(q("LPIPE4_ODD_%=:"))
    L += k_tile(1)
    L += [q("s_add_u32 s84, s84, 1"), q("s_cmp_lt_u32 s84, %[nk]"), q("s_cbranch_scc1 LPIPE4_%=")]
    L.append(q("LPIPE4_END_%=:"))
    # the zero-filling copies past the last k-tile must not land in the epilogue's LDS slabs; MFMA results need wait states before v_accvgpr_read
    L += [q("s_waitcnt vmcnt(0)"), q("s_nop 15"), q("s_nop 15")]

    clob = [f'"v{i}"' for i in range(136)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(84, 94)] + ['"m0"', '"scc"', '"memory"']
    ins = ['[ra] "s"(ra)', '[rb] "s"(rb)'] + [f'[ao{i}] "v"(ao[{i}])' for i in range(8)] + [f'[bo{i}] "v"(bo[{i}])' for i in range(8)] + \
          ['[abase] "v"(abase)', '[bbase] "v"(bbase)', '[m0s0] "s"(m0s0)', '[sa0] "s"(sa0)', '[sa1] "s"(sa1)', '[sb1] "s"(sb1)', '[sa2] "s"(sa2)',
           '[nk] "s"(nk)', '[period] "s"(period)', '[cnt] "s"(cnt)', '[jump] "s"(jump)']
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_pipe4_asm.py -- do not edit; the schedule (READ_SLOTS / COPY_SLOTS) lives in the generator.\n")
        f.write("// Main loop of the 4-wave 256x256x64 tile: see the generator's docstring for the register plan and the phase structure.\n")
        f.write("// ra / rb: buffer resources of the A / W tile; ao / bo: per-lane byte offsets of the 8 + 8 copy passes; abase / bbase: LDS byte address of\n")
        f.write("// this lane's fragment reads (stage 0, k half 0); m0s0: LDS byte address of this wave's copy destination in stage 0; sa0 / sa1 / sb1: scalar byte\n")
        f.write("// offsets of k-tiles 0 / 1 (0x7fffffff = no such tile); sa2: A offset of k-tile 2; period / cnt / jump: the conv row jump (bytes) every\n")
        f.write("// `period` k-tiles, cnt = 2 % period.\n")
        f.write("#ifdef ADA_OPERAND_BF16\n#define ADA_MFMA_SUFFIX \"bf16\"\n#else\n#define ADA_MFMA_SUFFIX \"f16\"\n#endif\n")
        f.write("ADA_DEV void pipe4_main_loop(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, const uint32_t (&ao)[8], const uint32_t (&bo)[8], uint32_t abase,\n")
        f.write("                             uint32_t bbase, uint32_t m0s0, uint32_t sa0, uint32_t sa1, uint32_t sb1, uint32_t sa2, uint32_t nk, uint32_t period,\n")
        f.write("                             uint32_t cnt, uint32_t jump) {\n")
        f.write("    asm volatile(\n")
        for ln in L:
            f.write("        " + ln + "\n")
        f.write("        :\n        : " + ", ".join(ins) + "\n        : " + ", ".join(clob) + ");\n}\n")
        f.write("#undef ADA_MFMA_SUFFIX\n")
    print(OUT, len(L), "asm lines")


if __name__ == "__main__":
    main()
