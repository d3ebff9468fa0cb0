#!/bin/bash
# kernel traces of the eager and the replayed timestep loop, per-kernel durations and launch gaps of each
#   bash tools/graph_vs_eager.sh [f32|w16a16]
PREC=${1:-f32}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for mode in eager graph; do
  OUT=gpurun_out/gve_${PREC}_$mode
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 tools/graph_vs_eager.py --mode $mode --precision $PREC > $OUT/line.txt 2> $OUT/err.txt
  find $OUT/t -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
  rm -rf $OUT/t
  echo "== $mode: $(cat $OUT/line.txt)"
  python3 tools/trace_gaps.py $OUT/kernel_trace.csv
  rm -f $OUT/kernel_trace.csv
done
for mode in eager graph; do   # and outside the profiler
  python3 tools/graph_vs_eager.py --mode $mode --precision $PREC
done
