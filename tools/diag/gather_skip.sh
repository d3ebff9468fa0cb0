set -x
for prec in f32 w16a16; do
for extra in "" "--fwd"; do
for dbg in 0 64 128; do
  if [ "$extra" = "--fwd" ] && [ $dbg = 128 ]; then continue; fi
  echo "== $prec $extra dbg=$dbg"
  timeout -k 10 120 python tools/phase_stamps.py --precision $prec $extra --dbg $dbg 2>&1 | grep -v amdgpu.ids
done; done; done
