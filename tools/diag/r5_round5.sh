python -m pytest tests -m gpu -x -q 2>&1 | tail -3
PASSES=2 bash tools/ab_libs.sh gpurun_ab/lib_cur.so gpurun_ab/lib_fused_bwd.so gpurun_ab/lib_fused_bwd_finish.so gpurun_ab/lib_fused_all.so
timeout -k 10 100 python tools/slot_times.py 2>&1 | grep -v amdgpu.ids
