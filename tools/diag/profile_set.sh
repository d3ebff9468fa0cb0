# a round's profile set on the GPU box: both precisions, kernel trace + PMC + SQ passes, phase stamps, plain bench lines, GPU tests
#   bash tools/diag/profile_set.sh r06_a      (needs `make -C flash-molecular-dynamics_amd/csrc stamps` for the stamps)
TAG=${1:-rXX}
export TMPDIR=/tmp
bash tools/profile_round.sh ${TAG} > gpurun_out/prof_${TAG}.log 2>&1; echo "profile_round fp32 rc=$?"
bash tools/profile_round.sh ${TAG}_w16a16 "--precision w16a16" > gpurun_out/prof_${TAG}_w16.log 2>&1; echo "profile_round w16a16 rc=$?"
bash tools/profile_sq.sh ${TAG} > gpurun_out/sq_${TAG}.log 2>&1; echo "profile_sq fp32 rc=$?"
bash tools/profile_sq.sh ${TAG}_w16a16 --precision w16a16 > gpurun_out/sq_${TAG}_w16.log 2>&1; echo "profile_sq w16a16 rc=$?"
for args in "" "--fwd" "--precision w16a16" "--precision w16a16 --fwd" "--block 0" "--precision w16a16 --block 0"; do
  echo "== $args"
  timeout -k 10 120 python tools/phase_stamps.py $args 2>&1 | grep -v amdgpu.ids
done > gpurun_out/${TAG}_phase_stamps.txt 2>&1
echo stamps done
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_args.json 2> gpurun_out/${TAG}_bench_driver_args.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --precision w16a16 --no-cpu-baseline > gpurun_out/${TAG}_bench_w16a16.json 2> gpurun_out/${TAG}_bench_w16a16.err; echo "bench w16 rc=$?"
