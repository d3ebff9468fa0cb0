# round 6: eight-wave W16A16 forward kernel with the producers' d / C values requested four tiles ahead
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q > gpurun_out/r6_ring_tests.txt 2>&1; tail -3 gpurun_out/r6_ring_tests.txt
PASSES=3 bash tools/ab_libs.sh gpurun_ab/r6_nodeswz.so gpurun_ab/r6_ring8.so > gpurun_out/r6_ab_ring.txt 2>&1; cat gpurun_out/r6_ab_ring.txt
for dbg in 0 9 2; do
  echo "== w16a16 fwd (eight waves, rings) dbg=$dbg"
  timeout -k 10 200 python tools/phase_stamps.py --precision w16a16 --fwd --dbg $dbg 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r6_ring_stamps.txt 2>&1; grep -v "^tiles\|^per-work" gpurun_out/r6_ring_stamps.txt
