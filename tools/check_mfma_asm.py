"""Audit of the hand-counted load streams of gemv_mfma_kernel (cdna_hip_programming.md 5.7 item 4): while an asm load is in flight
(issued by an asm statement, not yet covered by an asm `s_waitcnt vmcnt(N)` - in-order completion: a wait leaves the N youngest in
flight), NO compiler-generated instruction may read or write its destination registers (a copy, spill or reuse of a register whose
load has not landed is silent corruption).  Linear scan of the kernel's code; the stream loops keep the invariant from iteration to
iteration.  Usage: python tools/check_mfma_asm.py <file.s> [kernel-name-regex ...]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pats = sys.argv[2:] or [r"gemv_mfma_kernelILb1ELi16ELb0E", r"gemv_mfma_kernelILb1ELi16ELb1E"]


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


rc = 0
for pat in pats:
    i0 = [i for i, l in enumerate(lines) if re.match(r"_ZN\S*" + pat + r"\S*:", l)][0]
    i1 = next(i for i in range(i0, len(lines)) if "s_endpgm" in lines[i])
    fifo, inasm, bad, nload = [], False, [], 0
    for i in range(i0, i1):
        code = lines[i].split(";")[0]
        if "ASMSTART" in lines[i]:
            inasm = True
        elif "ASMEND" in lines[i]:
            inasm = False
        elif inasm:
            if "global_load_dwordx4" in code:
                fifo.append(regs(re.findall(r"v\[\d+:\d+\]", code)[0]))
                nload += 1
            m = re.search(r"s_waitcnt vmcnt\((\d+)\)", code)
            if m:
                n = int(m.group(1))
                fifo = fifo[len(fifo) - n:] if n else []
        else:
            used = set()
            for tok in re.findall(r"v\[\d+:\d+\]|v\d+", code):
                used |= regs(tok)
            if any(used & f for f in fifo):
                bad.append((i - i0, code.strip()))
            # compiler-issued vector memory operations and waits move the same in-order counter (kernels that mix both kinds)
            if re.search(r"\b(global|buffer|scratch|flat)_(load|store|atomic)", code):
                fifo.append(set())
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", code)
            if m:
                n = int(m.group(1))
                fifo = fifo[len(fifo) - n:] if n else []
    print(f"{pat}: {nload} asm loads; compiler instructions that touch a register while its asm load is in flight: {len(bad)}")
    for b in bad[:20]:
        print("  ", b)
    rc |= 1 if bad else 0
sys.exit(rc)
