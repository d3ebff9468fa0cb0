#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the bench command, then three PMC passes
# (FETCH_SIZE, WRITE_SIZE, MFMA busy cycles), each in its own run.
#   bash tools/profile_round.sh r01_c [extra bench.py arguments, e.g. "--precision w16a16"]
set -e
TAG=${1:-rXX}
EXTRA=${2:-}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-w16a16-leg --no-graph-leg --settle-seconds 0 $EXTRA > $OUT/bench_line.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-w16a16-leg --no-graph-leg --settle-seconds 0 $EXTRA > $OUT/pmc_fetch.out 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-w16a16-leg --no-graph-leg --settle-seconds 0 $EXTRA > $OUT/pmc_write.out 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-w16a16-leg --no-graph-leg --settle-seconds 0 $EXTRA > $OUT/pmc_mfma.out 2> $OUT/pmc_mfma.err
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json $OUT/pmc_mfma > /dev/null
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace/*/*kernel_trace.csv $OUT/pmc_fetch/*/*kernel_trace.csv $OUT/pmc_write/*/*kernel_trace.csv $OUT/pmc_mfma/*/*kernel_trace.csv
ls $OUT
