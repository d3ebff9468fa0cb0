timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "w16a16 or W16A16 or half or gptq or queue or balanced or chained or captured or wide_basis or reproduc" > gpurun_out/r6_wide_tests.txt 2>&1; tail -3 gpurun_out/r6_wide_tests.txt
PASSES=2 bash tools/ab_libs.sh gpurun_ab/r6_nodeswz.so $1 > gpurun_out/r6_ab_wide.txt 2>&1; cat gpurun_out/r6_ab_wide.txt
for dbg in 0 9 2; do
  echo "== w16a16 fwd (sixteen waves) dbg=$dbg"
  timeout -k 10 200 python tools/phase_stamps.py --precision w16a16 --fwd --dbg $dbg 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r6_wide_stamps.txt 2>&1; grep -v "^tiles\|^per-work" gpurun_out/r6_wide_stamps.txt
