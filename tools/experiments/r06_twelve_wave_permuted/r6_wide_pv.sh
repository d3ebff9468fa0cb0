# round 6: twelve-wave W16A16 forward kernel, PRODUCER diagnostic variants with the consumers idle (dbg bits 0 + 3):
# bits 12..15: no transcendentals / no Gaussians / no layer-0 MFMAs / no stores
for v in 0 16 32 64 128 48 112 240; do
  dbg=$((9 + v * 256))
  echo "== producers alone, variant $v (dbg=$dbg)"
  timeout -k 10 200 python tools/phase_stamps.py --precision w16a16 --fwd --dbg $dbg 2>&1 | grep "waves 8-11"
done
