python -m pytest tests -m gpu -x -q 2>&1 | tail -3
PASSES=2 bash tools/ab_libs.sh gpurun_ab/lib_cur.so gpurun_ab/lib_nodeout_res.so
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_args.json 2> /dev/null; echo "bench rc=$?"
python bench.py --pt --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r5_bench_pt_1gpu.json 2> /dev/null; echo "pt rc=$?"
python bench.py --precision w16a16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_bench_w16a16.json 2> /dev/null; echo "w16 rc=$?"
