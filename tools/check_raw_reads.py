"""ISA check behind csrc/s2st_asm.h: every hand-issued `ds_read_b64_tr_b16` (inline assembly: the compiler does not know its
destination is pending) must be followed by `s_waitcnt lgkmcnt(0)` before ANY instruction names one of its destination
registers.  usage: python tools/check_raw_reads.py file.s [...]   (tests/test_isa_checks.py compiles the GEMM sources and
runs it: ADVICE r2)"""
import re
import sys


def check(path):
    lines = open(path).read().splitlines()
    name, in_asm, pend, bad, total, msgs = "?", False, {}, 0, 0, []
    for idx, raw in enumerate(lines):
        t = raw.strip()
        m = re.match(r"^(_ZN\S+):", t)
        if m:
            name, pend = m.group(1), {}
            continue
        if "ASMSTART" in t:
            in_asm = True
            continue
        if "ASMEND" in t:
            in_asm = False
            continue
        ins = t.split(";")[0].strip()
        if not ins or ins.endswith(":") or ins.startswith("."):
            continue
        m = re.match(r"ds_read_b64_tr_b16\s+v\[(\d+):(\d+)\]", ins)
        if m and in_asm:
            total += 1
            for r in range(int(m.group(1)), int(m.group(2)) + 1):
                pend[r] = idx
            continue
        if ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins:
            pend.clear()
            continue
        if not pend:
            continue
        regs = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", ins):
            regs.update(range(int(a), int(b) + 1))
        for a in re.findall(r"\bv(\d+)\b", ins):
            regs.add(int(a))
        hit = [r for r in regs if r in pend]
        if hit:
            bad += 1
            msgs.append("%s line %d: %s  <- read at line %d" % (name[-80:], idx, ins[:90], pend[hit[0]]))
            for r in hit:
                pend.pop(r, None)
    return total, bad, msgs


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:] or ["/tmp/gemm_bf16-hip-amdgcn-amd-amdhsa-gfx950.s"]:
        total, bad, msgs = check(p)
        print("\n".join(msgs))
        print(p, ": hand-issued tr reads", total, "; uses of their destinations before an lgkmcnt(0):", bad)
        rc |= 1 if bad else 0
    sys.exit(rc)
