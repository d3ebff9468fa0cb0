timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gpu_tests_a.txt 2>&1; tail -3 gpurun_out/r6_gpu_tests_a.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench_a.json 2> gpurun_out/r6_bench_a.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --pt --no-cpu-baseline > gpurun_out/r6_bench_pt_1gpu.json 2> gpurun_out/r6_bench_pt.err; echo "pt rc=$?"
FMD_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --pt --steps 20 --warmup 5 > gpurun_out/r6_bench_pt_2ranks_rehearsal.json 2> gpurun_out/r6_bench_pt2.err; echo "pt2 rc=$?"
FMD_BENCH_REHEARSE=1 timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r6_bench_2ranks_rehearsal.json 2> gpurun_out/r6_bench_2.err; echo "2 ranks rc=$?"
bash tools/graph_vs_eager.sh f32 > gpurun_out/r6_graph_vs_eager_f32.txt 2>&1; tail -4 gpurun_out/r6_graph_vs_eager_f32.txt
bash tools/graph_vs_eager.sh w16a16 > gpurun_out/r6_graph_vs_eager_w16a16.txt 2>&1; tail -4 gpurun_out/r6_graph_vs_eager_w16a16.txt
