python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 4000 --reset-every 40 --no-cpu-baseline --no-w16a16-leg --no-graph-leg > gpurun_out/r5_bench_sustained_4000.json 2>/dev/null; echo "sustained rc=$?"
python bench.py --precision w16a16 --steps 2000 --no-cpu-baseline --no-graph-leg > gpurun_out/r5_bench_w16a16_2000.json 2>/dev/null; echo "w16 2000 rc=$?"
