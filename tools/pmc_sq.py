"""Per-kernel averages per launch of every counter found in rocprofv3 --pmc output directories.
  python tools/pmc_sq.py <dir> [<dir> ...]  > sq_counters.json
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import sys

res = {}
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                if "at::native" in k or "rocclr" in k or "Cijk" in k:
                    continue
                k = k.split("(")[0][:80]
                c = res.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
                c[0] += float(row["Counter_Value"])
                c[1] += 1
out = {k: {c: v[0] / v[1] for c, v in cs.items()} | {"launches_sampled": max(v[1] for v in cs.values())}
       for k, cs in res.items()}
print(json.dumps(out, indent=1, sort_keys=True))
