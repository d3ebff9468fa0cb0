#!/bin/bash
# Same-box A/B of library variants: tools/ab_libs.sh gpurun_ab/lib_a.so gpurun_ab/lib_b.so ...  (PASSES passes over all of
# them, default 2; each run is bench.py --no-cpu-baseline $BENCH_ARGS with the variant copied over the in-tree library;
# the original comes back)
LIB=flash-molecular-dynamics_amd/flashmd_amd/libflashmd_hip.so
cp $LIB /tmp/lib_orig.so
trap 'cp /tmp/lib_orig.so $LIB' EXIT   # whatever happens in between, the in-tree library comes back
for pass in $(seq 1 ${PASSES:-2}); do
  for v in "$@"; do
    cp $v $LIB
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 $BENCH_ARGS > /tmp/ab.json 2>/dev/null
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
k = d["kernel_ms_per_launch"]
w = d.get("w16a16")
print(sys.argv[1], round(d["value"]), "regions", [round(x, 4) for x in d["timed_regions_ms_per_step"]], "fwd", k["edge_fwd_kernel"],
      "bwd", k["edge_bwd_kernel"], "node fwd/out/bwd", k["node_fwd_kernel"], k["node_out_kernel"], k["node_bwd_kernel"],
      "pairs/finish", k["pairs_kernel"], k["finish_kernel"],
      *(["| w16", round(w["value"]), [round(x, 4) for x in w["timed_regions_ms_per_step"]], "fwd", w["kernel_ms_per_launch"]["edge_fwd_kernel"],
         "bwd", w["kernel_ms_per_launch"]["edge_bwd_kernel"]] if w else []))
PY
  done
done
cp /tmp/lib_orig.so $LIB
