#!/usr/bin/env python3
"""
Build-time guard for the hand-scheduled kernels of gf2_slabs.hip (run by `make`, see Makefile: build/slabs_isa.ok).

slab_compact_kernel<T> and slab_gather_fast_kernel<EXTRA, CROSS, SYN> issue their loads through inline assembly and wait with hand-counted
`s_waitcnt vmcnt(N)`.  The compiler neither sees those loads nor knows that their destination registers are still being filled,
so two things must hold in the generated code, and a compiler upgrade or an innocent edit can break either:

  1. Counts.  Vector memory operations complete in order; `vmcnt(N)` is right only if exactly the assumed operations are issued
     between two counted waits.  Checked: in the gather kernel's step loop every counted wait is followed by exactly
     1 store (2 with SYN: the syndrome piece) + 1 record load + 1 identity load (+ 1 for EXTRA) before the next one; in the compact kernel every sub-pass issues
     exactly T loads between counted waits (plus the tile's 4 record stores once per tile); no scratch (spill) traffic at all.
  2. Registers in flight.  Between a load's issue and the wait that covers it no instruction may read or write its destination
     registers (DESIGN.md section 3 records a compiler-placed v_mov of such a register that produced wrong syndromes).
     Checked by replaying every kernel's instructions in text order against a FIFO of outstanding operations: `vmcnt(N)`
     retires all but the youngest N; any other access to a register of an operation still in the FIFO is an error (see replay).

Usage: check_isa.py <file.s> (the output of hipcc -S --cuda-device-only gf2_slabs.hip), or check_isa.py --elim <gf2_elim.s> for the
panel kernels of gf2_elim.hip (see check_elim).  Exit code 1 on a violation.
"""
import re
import sys

KERNELS = {
    # <EXTRA, CROSS, SYN>: operations per step = stores (the partial weight; SYN: and the syndrome piece) + loads (record, identity
    # words; EXTRA: and the fifth dword)
    "slab_gather_fast_kernelILb0ELb0ELb0E": {"step_vmem": 3, "wait": 3, "stores": 1},
    "slab_gather_fast_kernelILb0ELb1ELb0E": {"step_vmem": 3, "wait": 3, "stores": 1},
    "slab_gather_fast_kernelILb1ELb0ELb0E": {"step_vmem": 4, "wait": 4, "stores": 1},
    "slab_gather_fast_kernelILb1ELb1ELb0E": {"step_vmem": 4, "wait": 4, "stores": 1},
    "slab_gather_fast_kernelILb0ELb0ELb1E": {"step_vmem": 4, "wait": 4, "stores": 2},
    "slab_gather_fast_kernelILb1ELb0ELb1E": {"step_vmem": 5, "wait": 5, "stores": 2},
    "slab_compact_kernelILi4ELb0E": {"rounds": 4},
    "slab_compact_kernelILi4ELb1E": {"rounds": 4},
    "slab_compact_kernelILi5ELb0E": {"rounds": 5},
    "slab_compact_kernelILi6ELb0E": {"rounds": 6},
    "slab_compact_kernelILi8ELb0E": {"rounds": 8},
}
VMEM = re.compile(r"^(global|buffer|scratch|flat)_(load|store|atomic)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernels_in(path):
    """{mangled name: [(line number, instruction text, inside inline asm)]}"""
    found, name, in_asm = {}, None, False
    for no, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            found[name] = []
            continue
        if name is None:
            continue
        text = line.split(";")[0].strip() if "#ASM" not in line else line.strip()
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        if not text or text.startswith("."):
            if text.startswith(".LBB") or re.match(r"^\.LBB\w+:", line):
                found[name].append((no, line.strip().split(":")[0] + ":", False))
            continue
        found[name].append((no, text, in_asm))
        if text.startswith("s_endpgm"):
            name = None
    return found


def step(fifo, inst, errors, kernel):
    """Applies one instruction to the FIFO of outstanding vector memory operations (tuples (destination registers, line))."""
    no, text, _ = inst
    op = text.split()[0]
    operands = text[len(op):]
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", text)
        return fifo[max(0, len(fifo) - int(m.group(1))):] if m else fifo
    touched = regs_of(operands)
    for dst, where in fifo:
        clash = dst & touched
        if clash:
            errors.add("%s line %d: `%s` touches v%s, still being filled by the load of line %d"
                       % (kernel, no, text, sorted(clash), where))
    if VMEM.match(op):
        first = operands.split(",")[0]
        dst = frozenset(regs_of(first)) if "_load" in op or "rtn" in op else frozenset()
        fifo = (fifo + ((dst, no),))[-MAX_FIFO:]
    return fifo


MAX_FIFO = 48


def replay(stream, errors, kernel):
    """Replays the kernel in text order.  The hand-scheduled loops are unrolled (four gather steps, two compact sub-passes) over
    the same rotating registers and laid out in execution order, so the text shows every steady-state overlap of a load in
    flight with the code of the following steps, except the one across the back edge, which repeats the pattern inside.
    if / else diamonds are exclusive paths: the else part starts from the FIFO as it was at the branch."""
    fifo = ()
    at_branch = {}
    prev = ""
    for inst in stream:
        no, text, in_asm = inst
        if text.endswith(":"):
            label = text[:-1]
            if prev.startswith("s_branch") and label in at_branch:
                fifo = at_branch[label]
            prev = text
            continue
        m = re.match(r"s_cbranch\w*\s+(\.LBB\w+)", text)
        if m:
            at_branch.setdefault(m.group(1), fifo)
        fifo = step(fifo, inst, errors, kernel)
        prev = text


def check_counts(name, stream, spec, errors):
    ops = [(no, text, in_asm) for no, text, in_asm in stream if not text.endswith(":")]
    if any(text.startswith("scratch_") for _, text, _ in ops):
        errors.add("%s: scratch (spill) traffic in a hand-scheduled kernel" % name)
    waits = [i for i, (_, text, in_asm) in enumerate(ops) if in_asm and text.startswith("s_waitcnt vmcnt(")]
    if "step_vmem" in spec:
        counted = [i for i in waits if ops[i][1] == "s_waitcnt vmcnt(%d)" % spec["wait"]]
        if len(counted) < 4:
            errors.add("%s: expected the four counted waits of the unrolled step loop, found %d" % (name, len(counted)))
        for a, b in zip(counted, counted[1:]):
            between = [text.split()[0] for _, text, _ in ops[a:b] if VMEM.match(text.split()[0])]
            stores = sum(1 for o in between if "_store" in o)
            loads = sum(1 for o in between if "_load" in o)
            if (stores, loads) != (spec["stores"], spec["step_vmem"] - spec["stores"]):
                errors.add("%s line %d: a step issues %d stores and %d loads, the wait counts assume %d and %d"
                              % (name, ops[a][0], stores, loads, spec["stores"], spec["step_vmem"] - spec["stores"]))
    else:
        rounds = spec["rounds"]
        loads_between = []
        for a, b in zip(waits, waits[1:] + [len(ops)]):
            between = [text.split()[0] for _, text, asm in ops[a:b] if asm and VMEM.match(text.split()[0])]
            loads_between.append((ops[a][0], sum(1 for o in between if "_load" in o and "dwordx2" in o), ops[a][1]))
        inside = [(no, n, w) for no, n, w in loads_between if w in ("s_waitcnt vmcnt(%d)" % rounds, "s_waitcnt vmcnt(%d)" % (rounds + 4))]
        if sum(1 for _, n, _ in inside if n) < 2:
            errors.add("%s: the two unrolled sub-passes with their counted waits were not found" % name)
        for no, n, w in inside:
            if n == 0:                                  # the two alternative waits of a sub-pass (first two / later ones) stand next to each other
                continue
            if n not in (rounds, 2 * rounds):          # one sub-pass refill (uniform path and clamped path are both in the text)
                errors.add("%s line %d: %d row loads follow `%s`, a sub-pass refills exactly %d" % (name, no, n, w, rounds))


def hot_loop_spills(stream):
    """Scratch operations inside the innermost loop (label .. backward branch) that contains a v_writelane; None if there is none."""
    labels, loops = {}, []
    for i, (_, text, _) in enumerate(stream):
        if text.endswith(":"):
            labels[text[:-1]] = i
    for i, (_, text, _) in enumerate(stream):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", text)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    best = None
    for a, b in loops:
        if any("v_writelane" in stream[k][1] for k in range(a, b)) and (best is None or b - a < best[1] - best[0]):
            best = (a, b)
    if best is None:
        return None
    return sum(1 for k in range(best[0], best[1]) if stream[k][1].startswith("scratch_"))


def check_elim(path):
    """gf2_elim.s: window_round writes a lane of two registers through `s_mov_b32 m0, ...; v_writelane_b32 ..., m0` in inline
    assembly with m0 on the clobber list (this clang has no v_writelane builtin and warns that clobbering a reserved register "may
    lead to undefined behaviour").  That is safe as long as the COMPILER never keeps anything in m0 in these kernels: checked here --
    no instruction outside the inline-assembly blocks mentions m0 -- together with: no scratch (spill) traffic in the panel kernels
    (rref_panel_kernel<8>, whose eight rows per lane do not fit 128 registers, is the one known exception and is reported)."""
    errors, notes = set(), []
    found = kernels_in(path)
    seen = 0
    for name, stream in found.items():
        panel = any(k in name for k in ("rref_panel_kernel", "rref_panel_stream_kernel", "norm_panel_kernel", "rref_sweep_panel_kernel", "sweep_stream_panel_kernel"))
        seen += 1 if panel else 0
        outside = [(no, text) for no, text, in_asm in stream if not in_asm and re.search(r"\bm0\b", text)]
        for no, text in outside:
            errors.add("%s line %d: `%s` uses m0 outside the inline assembly that owns it" % (name, no, text))
        spills = sum(1 for _, text, _ in stream if text.startswith("scratch_"))
        if spills and panel and "rref_sweep_panel_kernel" in name and not re.search(r"rref_sweep_panel_kernelILi\dELi8E", name):
            # (four rows per lane: the K column words and coefficients of a lane's rows) spills are accepted outside the pivot loop
            # of window_round -- the innermost loop that holds its v_writelane
            hot = hot_loop_spills(stream)
            if hot is None:
                errors.add("%s: the pivot loop (v_writelane) was not found" % name)
            elif hot > 2:
                errors.add("%s: %d scratch (spill) operations inside the pivot loop" % (name, hot))
            elif hot:
                notes.append("%s: %d scratch operations, %d of them in the pivot loop" % (name, spills, hot))
            else:
                notes.append("%s: %d scratch operations, none in the pivot loop" % (name, spills))
            continue
        if spills and panel:
            if "rref_panel_kernelILi8E" in name or re.search(r"rref_sweep_panel_kernelILi\dELi8E", name):
                notes.append("%s: %d scratch operations (eight rows per lane: known)" % (name, spills))
            else:
                errors.add("%s: %d scratch (spill) operations in a panel kernel" % (name, spills))
    if seen < 6:
        errors.add("only %d panel kernels were found in %s" % (seen, path))
    for e in sorted(errors):
        print("check_isa: " + e)
    for n in notes:
        print("check_isa: note: " + n)
    print("check_isa: %d panel kernels, %d problems" % (seen, len(errors)))
    return 1 if errors else 0


def main(path):
    errors = set()
    found = kernels_in(path)
    seen = 0
    for name, stream in found.items():
        spec = next((v for k, v in KERNELS.items() if k in name), None)
        if spec is None:
            continue
        seen += 1
        check_counts(name, stream, spec, errors)
        replay(stream, errors, name)
    if seen < 3:
        errors.add("only %d of the hand-scheduled kernels were found in %s" % (seen, path))
    for e in sorted(errors):
        print("check_isa: " + e)
    print("check_isa: %d kernels, %d problems" % (seen, len(errors)))
    return 1 if errors else 0


if __name__ == "__main__":
    if sys.argv[1] == "--elim":
        sys.exit(check_elim(sys.argv[2]))
    sys.exit(main(sys.argv[1]))
