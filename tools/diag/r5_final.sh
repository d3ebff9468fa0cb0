export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/profile_round.sh r05_b > gpurun_out/prof_r05_b.log 2>&1; echo "profile_round fp32 rc=$?"
bash tools/profile_round.sh r05_b_w16a16 "--precision w16a16" > gpurun_out/prof_r05_b_w16.log 2>&1; echo "profile_round w16a16 rc=$?"
bash tools/profile_sq.sh r05_b > gpurun_out/sq_r05_b.log 2>&1; echo "profile_sq fp32 rc=$?"
bash tools/profile_sq.sh r05_b_w16a16 --precision w16a16 > gpurun_out/sq_r05_b_w16.log 2>&1; echo "profile_sq w16a16 rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_final_bench_driver_args.json 2> /dev/null; echo "bench rc=$?"
python bench.py > gpurun_out/r5_final_bench_default.json 2> /dev/null; echo "bench default rc=$?"
python bench.py --pt --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r5_final_bench_pt_1gpu.json 2> /dev/null; echo "pt rc=$?"
python bench.py --precision w16a16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_final_bench_w16a16.json 2> /dev/null; echo "w16 rc=$?"
FMD_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --pt --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_final_bench_pt_2ranks_rehearsal.json 2> gpurun_out/r5_final_rehearsal.err; echo "rehearsal rc=$?"
timeout -k 10 300 python tools/small_molecules.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_final_small_molecules.txt; echo "small rc=$?"
timeout -k 10 100 python tools/slot_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_final_slot_times.txt
timeout -k 10 100 python tools/slot_times.py --precision w16a16 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5_final_slot_times.txt
