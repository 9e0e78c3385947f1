"""Audit of the hand-counted load stream of gemv_mfma_kernel<true, 16> (cdna_hip_programming.md 5.7 item 4): between the first and
the last asm statement of the kernel, NO compiler-generated instruction may read or write a register that an asm load targets
(a copy or a reuse of a register whose load is still in flight is silent corruption).  Usage: python tools/check_mfma_asm.py <file.s>"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
i0 = [i for i, l in enumerate(lines) if re.match(r"_ZN\S*gemv_mfma_kernelILb1ELi16E\S*:", l)][0]
i1 = next(i for i in range(i0, len(lines)) if "s_endpgm" in lines[i])
body = lines[i0:i1]


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


inflight, inasm, bad, nload = set(), False, [], 0
for i, l in enumerate(body):
    code = l.split(";")[0]
    if "ASMSTART" in l:
        inasm = True
    elif "ASMEND" in l:
        inasm = False
    elif inasm:
        if "global_load_dwordx4" in code:                      # an asm load: its destination is in flight until an asm wait consumes it
            inflight |= regs(re.findall(r"v\[\d+:\d+\]", code)[0])
            nload += 1
        elif "v_mfma" in code:                                  # (preceded by its s_waitcnt inside the same statement): srcA has landed
            inflight -= regs(re.findall(r"v\[\d+:\d+\]", code)[1])
        elif "s_waitcnt vmcnt(0)" in code:
            inflight = set()
    else:
        used = set()
        for tok in re.findall(r"v\[\d+:\d+\]|v\d+", code):
            used |= regs(tok)
        if used & inflight:
            bad.append((i, code.strip()))
print(f"{nload} asm loads; compiler instructions that touch a register while its asm load is in flight: {len(bad)}")
for b in bad[:30]:
    print(" ", b)
sys.exit(1 if bad else 0)
