#!/usr/bin/env python3
"""tools/isa_mix.py <file.s> <kernel-name-substring> — instruction mix of a kernel's hottest loop.

On gfx950 a wavefront issues about one instruction every 6 cycles whatever its kind (tools/ubench_issue.hip),
so the number of instructions in the frame loop — not their arithmetic content — is what a wave's time is made
of.  Prints the counts per class for the largest backward-branch loop of the kernel (and for the whole kernel).
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op in ("v_sqrt_f32", "v_rsq_f32", "v_rcp_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"):
        return "valu_trans"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "valu_mov"
    if op.startswith("v_cndmask"):
        return "valu_cndmask"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, sub = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    # function bodies: from "<name>:" to ".Lfunc_end"
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m and sub in m.group(1):
            start = i
            name = m.group(1)
            break
    if start is None:
        print("kernel not found")
        return 1
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start + 1:end]
    labels = {}
    insts = []  # (index, op, text)
    for l in body:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.startswith("//"):
            continue
        op = s.split()[0]
        insts.append((op, s))
    # loops = backward branches
    loops = []
    for i, (op, s) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((i - labels[tgt] + 1, labels[tgt], i))
    loops.sort(reverse=True)

    def mix(lo, hi):
        c = collections.Counter(classify(op) for op, _ in insts[lo:hi + 1])
        return c

    print(name)
    tot = mix(0, len(insts) - 1)
    print("whole kernel: %d instructions  %s" % (sum(tot.values()), dict(sorted(tot.items()))))
    for n, lo, hi in loops[:3]:
        c = mix(lo, hi)
        issue = sum(v for k, v in c.items())
        print("loop of %d instructions: %s" % (n, dict(sorted(c.items()))))
        ops = collections.Counter(op for op, _ in insts[lo:hi + 1])
        print("   top ops:", ", ".join("%s x%d" % kv for kv in ops.most_common(14)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
