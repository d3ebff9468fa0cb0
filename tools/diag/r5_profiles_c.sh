# final round-5 profile set (tag r05_c): both precisions, kernel trace + PMC + SQ passes, phase stamps, plain bench lines
export TMPDIR=/tmp
bash tools/profile_round.sh r05_c > gpurun_out/prof_r05_c.log 2>&1; echo "profile_round fp32 rc=$?"
bash tools/profile_round.sh r05_c_w16a16 "--precision w16a16" > gpurun_out/prof_r05_c_w16.log 2>&1; echo "profile_round w16a16 rc=$?"
bash tools/profile_sq.sh r05_c > gpurun_out/sq_r05_c.log 2>&1; echo "profile_sq fp32 rc=$?"
bash tools/profile_sq.sh r05_c_w16a16 --precision w16a16 > gpurun_out/sq_r05_c_w16.log 2>&1; echo "profile_sq w16a16 rc=$?"
export FMD_STAMPS_LIB=gpurun_ab/stamps.so
for args in "" "--fwd" "--precision w16a16" "--precision w16a16 --fwd" "--block 0" "--precision w16a16 --block 0"; do
  echo "== $args"
  timeout -k 10 120 python tools/phase_stamps.py $args 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05_c_phase_stamps.txt 2>&1
unset FMD_STAMPS_LIB
echo stamps done
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_c_bench_driver_args.json 2> gpurun_out/r05_c_bench_driver_args.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --precision w16a16 --no-cpu-baseline > gpurun_out/r05_c_bench_w16a16.json 2> gpurun_out/r05_c_bench_w16a16.err; echo "bench w16 rc=$?"
