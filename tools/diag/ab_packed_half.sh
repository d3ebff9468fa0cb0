# W16A16 pair kernels with the packed pair type (lib_split: fp32 kernels scalar, W16A16 kernels packed; lib_pk: both packed) against lib_base
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_reference_kernel_vectors_gpu.py -m gpu -x -q 2>&1 | tail -4 &&
PASSES=3 bash tools/ab_libs.sh gpurun_ab/lib_base.so gpurun_ab/lib_split.so
