#!/usr/bin/env python3
"""tools/lds_audit.py <file.s> [kernel-name-substring] — audits hand-counted LDS reads in gfx950 assembly.

The overlap kernels (melonix_amd/csrc/stft_overlap.h) issue `ds_read_b64` from inline asm and consume the results behind
counted `s_waitcnt lgkmcnt(n)` of their own.  hipcc neither counts those reads nor knows that a destination register is
not valid until the covering wait: it may read, copy or overwrite it in between (cdna_hip_programming.md, "what hipcc
does not do").  This walks every kernel of the file instruction by instruction, keeps the LGKM queue the hardware keeps
(LDS operations complete in issue order; `lgkmcnt(n)` returns when at most n are outstanding; scalar memory operations
complete out of order, so with one in flight only `lgkmcnt(0)` is a statement about anything) and reports
  * an instruction that reads or writes a VGPR whose ds_read is still outstanding,
  * a scalar memory operation issued or outstanding while counted reads are in flight and a counted (n > 0) wait follows,
  * outstanding reads at a label or branch (the walk is linear),
  * a counted wait with more than 15 LGKM operations outstanding of which some are hand-issued (the counter has four
    bits; a batch of any size behind `lgkmcnt(0)` is fine, and hipcc's own counted waits over its own reads are its business).
Exit code 1 on any finding.  Used by tests/test_abi.py on every STFT / phase-vocoder instantiation.
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def audit(lines, name):
    findings = []
    # outstanding LGKM operations in issue order: ("lds_rd", regs, line) — a ds_read issued from inline asm, the ones hipcc
    # does not track — | ("lds", None, line) — any other LDS operation (hipcc waits for its own) — | ("smem", None, line)
    queue = []
    in_asm = False
    for ln, raw in lines:
        if ";;#ASMSTART" in raw:
            in_asm = True
        elif ";;#ASMEND" in raw:
            in_asm = False
        s = raw.split(";")[0].strip()
        if not s or s.startswith(".") and not s.endswith(":"):
            continue
        if s.endswith(":") or s.startswith("s_cbranch") or s.startswith("s_branch") or s.startswith("s_endpgm") or s.startswith("s_setpc"):
            pend = [q for q in queue if q[0] == "lds_rd"]
            if pend:
                findings.append(f"{name}:{ln}: {len(pend)} counted ds_read(s) outstanding at '{s}' (first issued at line {pend[0][2]})")
            queue = []  # a new block starts from what the compiler itself guarantees
            continue
        op = s.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                if n > 0 and any(q[0] == "smem" for q in queue) and any(q[0] == "lds_rd" for q in queue):
                    findings.append(f"{name}:{ln}: counted wait '{s}' with a scalar memory operation in flight (returns out of order)")
                if n > 0 and len(queue) > 15 and any(q[0] == "lds_rd" for q in queue):
                    findings.append(f"{name}:{ln}: counted wait '{s}' with {len(queue)} LGKM operations outstanding (the counter has four bits)")
                if n == 0:
                    queue = []
                elif len(queue) > n:
                    queue = queue[len(queue) - n:]
            continue
        if op == "s_barrier" or op.startswith("s_nop") or op.startswith("s_sleep"):
            continue
        touched = vregs(s.split(None, 1)[1]) if " " in s else set()
        for kind, regs, at in queue:
            if kind == "lds_rd" and regs & touched:
                findings.append(f"{name}:{ln}: '{s}' touches v{sorted(regs & touched)} whose ds_read (line {at}) is still outstanding")
        if op.startswith("ds_"):
            if op.startswith("ds_read") or op.startswith("ds_load") or "rtn" in op or op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"):
                dst = s.split(None, 1)[1].split(",")[0]
                queue.append(("lds_rd" if in_asm else "lds", vregs(dst), ln))
            else:
                queue.append(("lds", None, ln))
        elif op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store") or op.startswith("s_memtime") or op.startswith("s_memrealtime"):
            queue.append(("smem", None, ln))
    return findings


def kernels(path):
    text = open(path).read().split("\n")
    cur, body = None, []
    for i, l in enumerate(text, 1):
        m = re.match(r"^(_Z\w+):", l)
        if m and cur is None:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            if l.startswith(".Lfunc_end"):
                yield cur, body
                cur = None
            else:
                body.append((i, l))


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    bad = 0
    n = 0
    for name, body in kernels(path):
        if sub not in name:
            continue
        n += 1
        f = audit(body, name[:60])
        reads = sum(1 for _, l in body if re.match(r"\s*ds_read", l))
        waits = sum(1 for _, l in body if re.search(r"lgkmcnt\(([1-9]|1[0-5])\)", l))
        print(f"{name}: {reads} ds_read, {waits} counted waits, {len(f)} finding(s)")
        for x in f[:20]:
            print("  " + x)
        bad += len(f)
    print(f"{n} kernel(s) audited, {bad} finding(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
