"""Instruction-class histogram of one kernel of a hipcc -S listing, cut at every s_barrier (diagnostics).
  python tools/isa_phases.py fused.s '<mangled-name-substring>'
Classes: mfma, valu (trans = v_exp/v_rcp/v_sqrt/v_sin/v_cos/v_rsq/v_log, pk = v_pk_*), lds (ds_*), vmem
(global_/buffer_/flat_), salu (s_* except waitcnt/barrier/nop), wait (s_waitcnt), nop."""
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "pk"
    if re.match(r"v_(exp|rcp|sqrt|sin|cos|rsq|log)_", op):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if re.match(r"(global|buffer|flat|scratch)_", op):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return None


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    seg, idx, label = {}, 0, None
    print(f"kernel at line {start + 1}..{end + 1}")
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if t.startswith(".LBB"):
                label = t.split(":")[0]
            continue
        op = t.split()[0]
        c = classify(op)
        if c is None:
            continue
        if c == "barrier":
            print(f"segment {idx:3d} (ends at barrier, last label {label}): " +
                  " ".join(f"{k}={v}" for k, v in sorted(seg.items())))
            seg, idx = {}, idx + 1
            continue
        seg[c] = seg.get(c, 0) + 1
    print(f"tail segment {idx}: " + " ".join(f"{k}={v}" for k, v in sorted(seg.items())))
    m = re.search(r"\.vgpr_count:\s+(\d+)", "\n".join(lines))
    for l in lines[end:end + 80]:
        if "NumVgprs" in l or "NumAgprs" in l or "ScratchSize" in l or "Occupancy" in l or "TotalNumVgprs" in l:
            print(l.strip())


main()
