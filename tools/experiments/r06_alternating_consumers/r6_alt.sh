# round 6: W16A16 forward kernel with two consumer waves per slice taking tiles in turn (ALT, 12 waves): tests, A/B, stamps
cp flash-molecular-dynamics_amd/flashmd_amd/libflashmd_hip.so /tmp/lib_keep.so
trap 'cp /tmp/lib_keep.so flash-molecular-dynamics_amd/flashmd_amd/libflashmd_hip.so' EXIT
cp gpurun_ab/r6_alt.so flash-molecular-dynamics_amd/flashmd_amd/libflashmd_hip.so
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "w16a16 or W16A16 or half or gptq or queue or balanced or chained or captured or reproduc or groups" > gpurun_out/r6_alt_tests.txt 2>&1; tail -3 gpurun_out/r6_alt_tests.txt
cp /tmp/lib_keep.so flash-molecular-dynamics_amd/flashmd_amd/libflashmd_hip.so
PASSES=3 BENCH_ARGS="--precision w16a16 --no-graph-leg" bash tools/ab_libs.sh gpurun_ab/r6_head.so gpurun_ab/r6_alt.so > gpurun_out/r6_ab_alt.txt 2>&1; cut -c1-220 gpurun_out/r6_ab_alt.txt
for dbg in 0 9 2; do
  echo "== w16a16 fwd ALT dbg=$dbg"
  FMD_STAMPS_LIB=gpurun_ab/r6_alt_stamps.so timeout -k 10 200 python tools/phase_stamps.py --precision w16a16 --fwd --dbg $dbg 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r6_alt_stamps.txt 2>&1; grep -v "^tiles\|^per-work" gpurun_out/r6_alt_stamps.txt
