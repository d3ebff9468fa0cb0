"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (the engine's launches are serial on one
stream): python tools/trace_gaps.py <kernel_trace.csv>.  Prints, per kernel name, calls / mean duration, and the mean
gap between a kernel's end and the next kernel's start, over the last steps of the run."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]   # the second half: timed region and later
dur, gap_after, n = defaultdict(float), defaultdict(float), defaultdict(int)
tot_busy = tot_gap = 0
for a, b in zip(rows[:-1], rows[1:]):
    name = a["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")[-48:]
    d = int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 200000:   # a host-side pause (sync, region boundary): not a launch gap
        continue
    dur[name] += d; gap_after[name] += g; n[name] += 1
    tot_busy += d; tot_gap += g
for k in sorted(dur, key=lambda k: -dur[k]):
    print(f"{k:50s} calls {n[k]:5d}  mean {dur[k] / n[k] / 1e3:8.1f} us   gap after {gap_after[k] / n[k] / 1e3:6.2f} us")
print(f"busy {tot_busy / 1e6:.2f} ms, gaps {tot_gap / 1e6:.2f} ms = {100 * tot_gap / (tot_busy + tot_gap):.1f} % of the stream time")
