for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_clk_bench.json 2> gpurun_out/r5_clk_bench.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r5_clk_bench.json'))
print(round(d['value']), d['ms_per_step'], d['device_clock']['avg_shader_mhz_over_timed_regions'], d['smi'], 'w16', round(d['w16a16']['value']), d['w16a16']['avg_shader_mhz_over_timed_regions'], d['w16a16']['smi'])"
done
python bench.py --precision w16a16 --steps 200 --no-cpu-baseline --no-graph-leg > gpurun_out/r5_clk_w16.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r5_clk_w16.json'))
print('w16 200 steps', round(d['value']), d['ms_per_step'], d['device_clock']['avg_shader_mhz_over_timed_regions'], d['smi'])"
