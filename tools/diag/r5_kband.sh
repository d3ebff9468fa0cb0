python -m pytest tests/test_model_gpu.py tests/test_reference_simulation.py tests/test_parallel_tempering_reference.py -m gpu -x -q 2>&1 | tail -12
PASSES=2 bash tools/ab_libs.sh gpurun_ab/lib_fused_bwd.so gpurun_ab/lib_kband.so
timeout -k 10 100 python tools/slot_times.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 100 python tools/slot_times.py --precision w16a16 2>&1 | grep -v amdgpu.ids
