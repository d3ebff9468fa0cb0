# phase stamps of the W16A16 pair kernels (diagnostic build copied to gpurun_ab/stamps.so)
export FMD_STAMPS_LIB=gpurun_ab/stamps.so
for a in "--precision w16a16" "--precision w16a16 --fwd" "--precision w16a16 --block 0" "" "--fwd"; do
  echo "== $a"; timeout -k 10 200 python tools/phase_stamps.py $a 2>&1 | grep -v amdgpu.ids
done
