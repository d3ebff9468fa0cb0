# node-kernel diagnostic variants (gpurun_ab/nd_K.so: bit 0 = no global stores, bit 1 = row loads from 8 cached tiles): per-launch times
LIB=flash-molecular-dynamics_amd/flashmd_amd/libflashmd_hip.so
cp $LIB /tmp/lib_keep.so
for v in 0 1 2 3 0; do
  cp gpurun_ab/nd_$v.so $LIB
  echo "== nd_$v"
  timeout -k 10 200 python tools/slot_times.py --reps 20 2>&1 | grep -E "node_|embed|sum"
done
cp /tmp/lib_keep.so $LIB
