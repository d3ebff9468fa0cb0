# W16A16 consumers: aggregate update on packed pairs (lib_pkupd) against lib_split
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_reference_kernel_vectors_gpu.py -m gpu -x -q 2>&1 | tail -4 &&
BENCH_ARGS="--precision w16a16" PASSES=3 bash tools/ab_libs.sh gpurun_ab/lib_split.so gpurun_ab/lib_pkupd.so
