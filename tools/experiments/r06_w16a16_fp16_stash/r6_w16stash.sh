# W16A16 fp16 filter stash (FMD_FLAG_W16A16_FILTER_STASH through FMD_W16_STASH=1): parity of the stash form, then a same-box A/B
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/r6_w16stash_parity.txt 2>&1
import os, sys, torch
sys.path[:0] = [".", "flash-molecular-dynamics_amd"]
import flashmd_amd.schnet as S
from flashmd_amd import engine, synthetic
torch.manual_seed(0)
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64], num_interactions=3).to("cuda").eval()
for B, n in ((128, 269), (8, 269), (5, 400), (700, 35)):
    pos, types, ptr, _ = synthetic.batch(B, n, 5.0)
    pos, types, ptr = pos.cuda(), types.cuda(), ptr.cuda()
    os.environ.pop("FMD_W16_STASH", None)
    a = engine.FusedSchNetEngine(model, precision="w16a16")
    ref = engine.FusedSchNetEngine(model, precision="f32")
    os.environ["FMD_W16_STASH"] = "1"
    b = engine.FusedSchNetEngine(model, precision="w16a16")
    ea, fa = a.energy_forces(pos, types, ptr, n)
    eb, fb = b.energy_forces(pos, types, ptr, n)
    eb2, fb2 = b.energy_forces(pos, types, ptr, n)
    er, fr = ref.energy_forces(pos, types, ptr, n)
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    print(f"B={B} n={n}: stash vs plain {rel(fb, fa):.3e}  plain vs f32 {rel(fa, fr):.3e}  stash vs f32 {rel(fb, fr):.3e}  energy equal {bool(torch.equal(ea, eb))}  repeat bit-identical {bool(torch.equal(fb, fb2))}  ws {a._ws.numel()} -> {b._ws.numel()}")
PY
cat gpurun_out/r6_w16stash_parity.txt
FMD_W16_STASH=1 timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "w16 and not no_filter_stash" > gpurun_out/r6_w16stash_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6_w16stash_tests.txt
for pass in 1 2 3; do
  for v in 0 1; do
    FMD_W16_STASH=$v timeout -k 10 300 python bench.py --precision w16a16 --no-cpu-baseline --steps 20 --warmup 5 > /tmp/ab.json 2>/dev/null
    python - $v <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
k = d["kernel_ms_per_launch"]
print("stash", sys.argv[1], round(d["value"]), [round(x, 4) for x in d["timed_regions_ms_per_step"]], "fwd", k["edge_fwd_kernel"], "bwd", k["edge_bwd_kernel"])
PY
  done
done 2>&1 | tee gpurun_out/r6_w16stash_ab.txt
