# round 6: twelve-wave W16A16 forward kernel, diagnostic variants (dbg bits 8..11: no gathers / no aggregate accesses / no
# lane swaps / no list loads; bit 0, 3: consumers skip update / layer 1; bit 1: producers skip staging)
for dbg in 0 9 2 256 512 1024 2048 768 1792 3840; do
  echo "== w16a16 fwd (twelve waves) dbg=$dbg"
  timeout -k 10 200 python tools/phase_stamps.py --precision w16a16 --fwd --dbg $dbg 2>&1 | grep -v amdgpu.ids
done
