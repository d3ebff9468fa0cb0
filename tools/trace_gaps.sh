#!/bin/bash
# kernel trace of a short bench run + the idle gaps between its kernels: bash tools/trace_gaps.sh <tag> [bench args]
TAG=${1:-tmp}; shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/gaps_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-w16a16-leg --settle-seconds 0 "$@" > $OUT/bench_line.json 2> $OUT/err.txt
find $OUT/t -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
find $OUT/t -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/t
python3 tools/trace_gaps.py $OUT/kernel_trace.csv > $OUT/gaps.txt
rm -f $OUT/kernel_trace.csv
cat $OUT/gaps.txt
