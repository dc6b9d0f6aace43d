#!/usr/bin/env python
"""Build-time ISA guards for kernels whose correctness depends on things the compiler is never told (csrc/build.py runs them; they
can be run by hand on any gfx9 assembly listing:  python amodal-depth-anything_amd/csrc/isa_guard.py file.s [--kernel substr] [--no-packed-f32]).

1. **In-flight registers.**  ada_tail.hip fetches with inline-asm ``global_load_dwordx4`` into C++ variables and waits with a hand-counted
   ``s_waitcnt vmcnt(N)``.  The compiler does not know that those registers are still being written: to it the value exists as soon as the
   load statement has "executed", so it may copy or read the register *before* the wait (a tied "+v" operand of the wait statement is
   satisfied by copying the input into another register first; a back-edge phi is a copy as well).  That is the root cause of the wrong
   results recorded in profiles/r03_p_fused_tail.txt -- with SLP vectorisation on, the register allocator parks row 0 in other registers
   with eight ``v_mov_b64`` placed just above ``s_waitcnt vmcnt(16)`` (profiles/r04_a_tail_inflight_register_root_cause.txt).
   The guard simulates the vmcnt and lgkmcnt FIFOs over the listing (gfx9: vector loads, stores and LDS-DMA copies retire in issue order on
   vmcnt; LDS reads in issue order on lgkmcnt, which scalar loads share out of order) and reports every instruction that reads or writes a
   VGPR while a load into it may still be outstanding.  ada_attention.hip has the same construction on the LDS side (inline-asm ds_read
   into C++ variables, "+v"-tied s_waitcnt lgkmcnt) and is checked by the same rule.
2. **Packed fp32 VALU ops** (``v_pk_*_f32``) in files that must not contain them (the configuration the tail kernel was validated in).
3. **AGPR contract of the generated 4-wave GEMM loop** (ada_igemm_pipe4.inc): after the asm block the 256 accumulators live in a[0:255]
   and are fetched by later asm statements; the compiler only knows them as clobbered.  Between the end of the loop and the end of the
   kernel nothing but ``v_accvgpr_read`` may touch an AGPR (no MFMA, no ``v_accvgpr_write``, no AGPR spill traffic).
"""
import re
import sys

VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
AREG = re.compile(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b")
LABEL = re.compile(r"^([.\w$]+):")
VMEM = re.compile(r"^(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|flat_load|flat_store|flat_atomic|scratch_load|scratch_store)")


def _regs(text, pat=VREG):
    out = set()
    for m in pat.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernels(asm_text):
    """name -> list of (line number, instruction or label text) for every function of the listing (labels declared ``.type X,@function``,
    or mangled C++ names when the listing carries no .type directives)."""
    funcs = set(re.findall(r"^\s*\.type\s+([.\w$]+),@function", asm_text, flags=re.M))
    out, name, body = {}, None, []
    for no, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.split(";")[0].strip() if not raw.lstrip().startswith(";;#") else ""
        if not line:
            continue
        m = LABEL.match(line)
        if m and (m.group(1) in funcs or (not funcs and m.group(1).startswith("_Z"))):
            if name:
                out[name] = body
            name, body = m.group(1), []
            continue
        if line.startswith(".Lfunc_end"):
            if name:
                out[name] = body
            name, body = None, []
            continue
        if name is not None and (not line.startswith(".") or LABEL.match(line)):
            body.append((no, line))
    if name:
        out[name] = body
    return out


class Counters:
    """Outstanding memory operations of one wave: ``vm`` = vmcnt FIFO, ``lgkm`` = lgkmcnt FIFO; entries are (dst VGPR set, line, kind)."""

    def __init__(self, other=None):
        self.vm = [(set(r), ln, k) for r, ln, k in other.vm] if other else []
        self.lgkm = [(set(r), ln, k) for r, ln, k in other.lgkm] if other else []


def _returns(ins):
    """A global / buffer / flat atomic returns its pre-operation value (and therefore writes its destination VGPRs) when the return bit is set:
    spelled ``sc0`` on gfx940 / gfx950, ``glc`` on older gfx9 listings."""
    toks = ins.replace(",", " ").split()
    return "sc0" in toks or "glc" in toks


def _step(c, no, ins, report):
    """Advances the counters over one instruction; appends (line, instruction, registers, lines of the pending loads) to report."""
    op = ins.split()[0]
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", ins)
        if m:
            del c.vm[:max(0, len(c.vm) - int(m.group(1)))]
        m = re.search(r"lgkmcnt\((\d+)\)", ins)
        if m:
            keep = int(m.group(1))
            # LDS operations return in order; scalar loads (same counter) do not: with one pending only lgkmcnt(0) retires anything for sure
            if keep == 0 or not any(k == "smem" for _, _, k in c.lgkm):
                del c.lgkm[:max(0, len(c.lgkm) - keep)]
        return
    if op.startswith("s_waitcnt_"):   # s_waitcnt_vscnt etc. do not exist on gfx9; be conservative: nothing retires
        return
    is_vmem = VMEM.match(op) is not None
    is_lds = op.startswith("ds_")
    is_smem = op.startswith("s_load") or op.startswith("s_buffer_load")
    dst = set()
    touched = _regs(ins)
    to_lds = "_lds_" in op or " lds" in ins   # LDS-DMA (global_load_lds_*, buffer_load ... lds): counted by vmcnt, no VGPR destination
    if is_vmem and ("load" in op or ("atomic" in op and _returns(ins))) and not to_lds:
        dst = _regs(ins[len(op):].split(",")[0])
    if is_lds and (op.startswith("ds_read") or "_rtn" in op or op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle")):
        dst = _regs(ins[len(op):].split(",")[0])
    pending = set()
    for regs, _, _ in c.vm + c.lgkm:
        pending |= regs
    bad = (touched - dst) & pending if (is_vmem or is_lds) else touched & pending   # a load may re-target a register whose older load is pending (in-order return)
    if bad:
        owners = sorted({ln for regs, ln, _ in c.vm + c.lgkm if regs & bad})
        report.append((no, ins, sorted(bad), owners))
    if is_vmem:
        c.vm.append((dst, no, "vmem"))
    elif is_lds:
        c.lgkm.append((dst, no, "lds"))
    elif is_smem:
        c.lgkm.append((set(), no, "smem"))


def check_inflight(body):
    """Linear pass + one extra trip around every backward branch with the counter state found at the branch.
    Control-flow limits (known false negatives, by construction): a FORWARD branch is treated as fall-through -- the state that reaches its target
    over the taken edge is not simulated -- and a loop is walked once more with the state at its back edge, not to a fixed point.  The guarded
    kernels (fused tail, attention) keep their hand-counted fetches inside straight-line loop bodies, which is the shape this covers; the
    -fno-slp-vectorize flag on ada_tail.hip stays as the second line of defence."""
    report, c = [], Counters()
    labels = {}
    state_at = {}
    for idx, (no, ins) in enumerate(body):
        m = LABEL.match(ins)
        if m:
            labels[m.group(1)] = idx
            continue
        _step(c, no, ins, report)
        op = ins.split()[0]
        if op in ("s_branch",) or op.startswith("s_cbranch"):
            tgt = ins.split()[-1]
            if tgt in labels:   # backward edge
                state_at[(labels[tgt], idx)] = Counters(c)
    for (lo, hi), st in state_at.items():
        c2 = Counters(st)
        for no, ins in body[lo:hi + 1]:
            if LABEL.match(ins):
                continue
            _step(c2, no, ins, report)
    seen, uniq = set(), []
    for r in report:
        if r[0] not in seen:
            seen.add(r[0])
            uniq.append(r)
    return sorted(uniq)


def check_packed_f32(body):
    return [(no, ins) for no, ins in body if re.match(r"v_pk_\w+_f32\b", ins)]


def check_agpr_after_loop(body, end_marker="LPIPE4_END", begin_marker="LPIPE4_BEGIN"):
    """Instructions OUTSIDE the generated loop that touch an AGPR other than by v_accvgpr_read.  The generated asm brackets itself with the labels
    LPIPE4_BEGIN_<n> / LPIPE4_END_<n>; everything outside those brackets is compiler-generated code that knows the AGPRs only as clobbered while the
    epilogue's dumps fetch the accumulators from them -- and in the persistent kernel (a tile loop around loop + epilogue) the code textually BEFORE
    the asm runs after it as well.  Listings without a begin label (older form): everything after the last end label."""
    begins = [idx for idx, (no, ins) in enumerate(body) if begin_marker in ins and LABEL.match(ins)]
    ends = [idx for idx, (no, ins) in enumerate(body) if end_marker in ins and LABEL.match(ins)]
    if not ends:
        return None
    bad = []
    if begins:
        inside = False
        for no, ins in body:
            if LABEL.match(ins):
                if begin_marker in ins:
                    inside = True
                elif end_marker in ins:
                    inside = False
                continue
            if inside or not _regs(ins, AREG):
                continue
            if not ins.startswith("v_accvgpr_read"):
                bad.append((no, ins))
        return bad
    for no, ins in body[ends[-1] + 1:]:
        if LABEL.match(ins) or not _regs(ins, AREG):
            continue
        if not ins.startswith("v_accvgpr_read"):
            bad.append((no, ins))
    return bad


def main(argv):
    path = argv[1]
    sub = argv[argv.index("--kernel") + 1] if "--kernel" in argv else ""
    text = open(path).read()
    rc = 0
    for name, body in kernels(text).items():
        if sub not in name:
            continue
        rep = check_inflight(body)
        print(f"{name}: {len(body)} instructions, {len(rep)} in-flight register violation(s)")
        for no, ins, regs, owners in rep[:40]:
            print(f"  line {no}: {ins}    <- v{regs} still being loaded (load at line {owners})")
        rc |= bool(rep)
        if "--no-packed-f32" in argv:
            pk = check_packed_f32(body)
            print(f"  packed fp32 VALU ops: {len(pk)}")
            rc |= bool(pk)
        ag = check_agpr_after_loop(body)
        if ag is not None:
            print(f"  AGPR accesses after the generated loop other than v_accvgpr_read: {len(ag)}")
            for no, ins in ag[:10]:
                print(f"    line {no}: {ins}")
            rc |= bool(ag)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv))
