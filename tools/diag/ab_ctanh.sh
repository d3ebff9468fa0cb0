# C tanh(z) as one fma on the reciprocal (lib_ctanh) against tanh then a multiplication (lib_base): tests, then same-box A/B
timeout -k 10 700 python -m pytest tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -5 &&
PASSES=3 bash tools/ab_libs.sh gpurun_ab/lib_base.so gpurun_ab/lib_ctanh.so
