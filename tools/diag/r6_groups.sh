# round 6: molecule groups for batches of small molecules -- tests, then throughput beside the ungrouped engine
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q > gpurun_out/r6_groups_tests.txt 2>&1; tail -4 gpurun_out/r6_groups_tests.txt
timeout -k 10 300 python tools/small_molecules.py > gpurun_out/r6_small_molecules.txt 2>&1; grep -v amdgpu.ids gpurun_out/r6_small_molecules.txt
for a in "--batch 2048 --beads 10" "--batch 4096 --beads 35"; do echo "== $a"; timeout -k 10 150 python tools/slot_times.py $a --reps 10 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r6_small_slots.txt 2>&1; cat gpurun_out/r6_small_slots.txt
