for prec in f32 w16a16; do
  echo "== $prec fwd baseline (stamps build)"
  timeout -k 10 120 python tools/phase_stamps.py --precision $prec --fwd 2>&1 | grep -v amdgpu.ids
  echo "== $prec fwd, producers idle (dbg 2)"
  timeout -k 10 120 python tools/phase_stamps.py --precision $prec --fwd --dbg 2 2>&1 | grep -v amdgpu.ids
  echo "== $prec fwd HOISTED aggregate reads (experiment, wrong sums)"
  FMD_STAMPS_LIB=gpurun_ab/stamps_hoist.so timeout -k 10 120 python tools/phase_stamps.py --precision $prec --fwd 2>&1 | grep -v amdgpu.ids
done
