#!/bin/bash
# Same-box A/B of library variants: tools/ab_libs.sh gpurun_ab/lib_a.so gpurun_ab/lib_b.so ...  (two passes over all of
# them; each run is bench.py --no-cpu-baseline $BENCH_ARGS with the variant copied over the in-tree library; the original comes back)
LIB=flash-molecular-dynamics_amd/flashmd_amd/libflashmd_hip.so
cp $LIB /tmp/lib_orig.so
for pass in 1 2; do
  for v in "$@"; do
    cp $v $LIB
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-w16a16-leg $BENCH_ARGS > /tmp/ab.json 2>/dev/null
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
k = d["kernel_ms_per_launch"]
print(sys.argv[1], round(d["value"]), "fwd", k["edge_fwd_kernel"], "bwd", k["edge_bwd_kernel"], "node fwd/out/bwd", k["node_fwd_kernel"], k["node_out_kernel"], k["node_bwd_kernel"], "pairs/finish", k["pairs_kernel"], k["finish_kernel"])
PY
  done
done
cp /tmp/lib_orig.so $LIB
