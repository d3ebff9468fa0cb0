# generic: GPU tests of the engine, then a same-box A/B of gpurun_ab/lib_cur.so against every other library under gpurun_ab/
timeout -k 10 1000 python -m pytest tests/test_model_gpu.py tests/test_reference_kernel_vectors_gpu.py -m gpu -x -q 2>&1 | tail -3 &&
PASSES=${PASSES:-3} bash tools/ab_libs.sh gpurun_ab/lib_cur.so $(ls gpurun_ab/*.so | grep -v lib_cur.so)
