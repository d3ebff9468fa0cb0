#!/bin/bash
# SQ instruction / issue / LDS counters of the engine kernels on the GPU box: two rocprofv3
# --pmc passes (8 SQ slots each) of a short bench run, counters only beside the kernel trace.
#   bash tools/profile_sq.sh r02_a [bench args]
set -e
TAG=${1:-rXX}; shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/sq_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
P1="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"
P2="SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/p1 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-w16a16-leg --no-graph-leg --settle-seconds 0 "$@" > $OUT/p1.out 2> $OUT/p1.err || echo "pass 1 failed"
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/p2 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-w16a16-leg --no-graph-leg --settle-seconds 0 "$@" > $OUT/p2.out 2> $OUT/p2.err || echo "pass 2 failed"
python3 tools/pmc_sq.py $OUT/p1 $OUT/p2 > $OUT/sq_counters.json
rm -rf $OUT/p1/*/*kernel_trace.csv $OUT/p2/*/*kernel_trace.csv
tail -c 3000 $OUT/sq_counters.json
