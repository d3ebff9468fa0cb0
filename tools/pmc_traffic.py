"""Turn rocprofv3 --pmc passes into profiles/pmc_traffic.json (HBM-side bytes per launch, per engine kernel).

  python tools/pmc_traffic.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Following MI355X_MICROARCH.md (HBM section):
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled; WRITE_SIZE is used as is.
The doubling is calibrated for wide coalesced reads; the pair kernels' row gathers are 4 bytes per lane
(256 contiguous bytes per half wave), so "hbm_bytes_per_launch" is an upper estimate for them.
"""
import csv
import glob
import json
import os
import sys

ENGINE = ["pairs_kernel", "node_fwd_kernel", "edge_fwd", "node_out_kernel", "node_bwd_kernel", "edge_bwd",
          "finish_kernel"]
NAMES = {"edge_fwd": "edge_fwd_kernel", "edge_bwd": "edge_bwd_kernel"}


def collect(d, counter):
    per = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                k = row["Kernel_Name"]
                for e in ENGINE:
                    if e in k:
                        per.setdefault(NAMES.get(e, e), []).append(float(row["Counter_Value"]))
                        break
    return per


def main():
    fdir, wdir, out = sys.argv[1:4]
    fetch, write = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [])))
        w = sum(write.get(k, [0])) / max(1, len(write.get(k, [])))
        res[k] = {"fetch_size_kib_raw": f, "write_size_kib": w, "launches_sampled": len(fetch.get(k, [])),
                  "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
