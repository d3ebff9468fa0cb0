LIB=flash-molecular-dynamics_amd/flashmd_amd/libflashmd_hip.so
cp $LIB /tmp/lib_keep.so
cp gpurun_ab/lib_node8.so $LIB
timeout -k 10 500 python -m pytest tests/test_model_gpu.py tests/test_reference_simulation.py -m gpu -x -q 2>&1 | tail -15
cp /tmp/lib_keep.so $LIB
bash tools/ab_libs.sh gpurun_ab/lib_base.so gpurun_ab/lib_node8.so
timeout -k 10 200 python tools/small_molecules.py 2>&1 | grep -v amdgpu.ids
