#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -save-temps .s file (build-time evidence for DESIGN.md's
"instructions issued per MFMA" figures).  usage: isa_loop_stats.py file.s <substring of the mangled kernel name>
Prints, for every backward branch (loop), the instruction counts by class between the label and the branch."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "lds_write"
    if op.startswith("ds_"):
        return "lds_other"
    if op.startswith("buffer_load") or op.startswith("global_load") or op.startswith("scratch_load"):
        return "vmem_load"
    if op.startswith("buffer_store") or op.startswith("global_store") or op.startswith("scratch_store"):
        return "vmem_store"
    if op.startswith("buffer_") or op.startswith("global_"):
        return "vmem_other"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "s_branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l.split(":")[0])
    end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    # find the last s_endpgm of the function: up to .Lfunc_end
    fend = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:fend]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    print(f"{lines[start][:110]}  ({fend - start} lines)")
    for i, l in enumerate(body):
        m = re.match(r"^\s+(s_cbranch\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in labels and labels[m.group(2)] < i:
            lo = labels[m.group(2)]
            c = Counter()
            for k in body[lo:i + 1]:
                t = k.strip()
                if not t or t.startswith(".") or t.startswith(";") or t.endswith(":"):
                    continue
                c[classify(t.split()[0])] += 1
            tot = sum(c.values())
            if c["mfma"] or tot > 40:
                print(f"  loop {m.group(2)} lines {lo}-{i}: total {tot}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
