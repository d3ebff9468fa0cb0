python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_end_bench.json 2>/dev/null; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r5_end_bench.json'))
print(round(d['value']), d['ms_per_step'], d['timed_regions_ms_per_step'], d['parity']['forces_rel_worst_molecule'], 'w16', round(d['w16a16']['value']), d['w16a16']['parity']['forces_rel'], d['smi'], d['w16a16']['smi'])"
