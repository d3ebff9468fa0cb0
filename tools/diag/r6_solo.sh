# round 6: solo rates of the two wave roles of the W16A16 forward pair kernel (diagnostic build, timing only)
for dbg in 0 1 8 9 2; do
  echo "== w16a16 fwd dbg=$dbg"
  timeout -k 10 200 python tools/phase_stamps.py --precision w16a16 --fwd --dbg $dbg 2>&1 | grep -v amdgpu.ids
done
