python -m pytest tests/test_model_gpu.py tests/test_reference_simulation.py tests/test_parallel_tempering_reference.py -m gpu -x -q 2>&1 | tail -5
bash tools/ab_libs.sh gpurun_ab/lib_node8.so gpurun_ab/lib_queue.so
timeout -k 10 200 python tools/small_molecules.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 100 python tools/slot_times.py --batch 1024 --beads 35 2>&1 | grep -v amdgpu.ids
