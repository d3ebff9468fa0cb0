python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 10 300 python tools/small_molecules.py 2>&1 | grep -v amdgpu.ids
PASSES=2 bash tools/ab_libs.sh gpurun_ab/lib_cur.so gpurun_ab/lib_node_default_sched.so gpurun_ab/lib_node_ilp.so
