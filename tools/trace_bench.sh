#!/bin/bash
# kernel-trace summary of the bench command (true per-kernel durations): bash tools/trace_bench.sh <tag> [bench args]
set -e
TAG=${1:-tmp}; shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/trace_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-w16a16-leg --no-graph-leg "$@" > $OUT/bench_line.json 2> $OUT/err.txt
find $OUT/t -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/t
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'at::native' in n or 'rocclr' in n: continue
    print(n.replace('(anonymous namespace)::', '')[:64].ljust(64), r['Calls'].rjust(4), round(float(r['AverageNs']) / 1000, 1))
PY
