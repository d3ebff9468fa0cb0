# node-kernel diagnostic variants (gpurun_ab/nd_K.so from tools/diag/node_diag_build.sh: bit 0 = no global stores, bit 1 = row
# loads from 8 cached tiles -- WRONG results, timing only): per-launch times.  The variants are loaded through FMD_DIAG_LIB
# (tools/slot_times.py); the product library is never overwritten.
for v in 0 1 2 3 0; do
  echo "== nd_$v"
  FMD_DIAG_LIB=gpurun_ab/nd_$v.so timeout -k 10 200 python tools/slot_times.py --reps 20 2>&1 | grep -E "node_|embed|sum"
done
