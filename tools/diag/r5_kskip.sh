python -m pytest tests/test_model_gpu.py tests/test_reference_simulation.py -m gpu -x -q 2>&1 | tail -3
PASSES=2 bash tools/ab_libs.sh gpurun_ab/lib_fused_bwd.so gpurun_ab/lib_kskip.so
