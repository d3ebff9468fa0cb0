python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "one_and_four or queue or captured" 2>&1 | tail -4
