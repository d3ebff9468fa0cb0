python -m pytest tests -m gpu -x -q 2>&1 | tail -3
PASSES=3 bash tools/ab_libs.sh gpurun_ab/lib_fused_bwd.so gpurun_ab/lib_fused_bwd_k4.so
