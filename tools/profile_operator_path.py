"""Operator path (the drop-in Python operators of SURVEY 8b, [E,F] tensors in HBM) on the bench batch: one
energy+force evaluation per step, engine gate off.  Run under rocprofv3 --kernel-trace --stats to get the
per-kernel times that DESIGN.md section 4 compares with each kernel's algorithmic HBM bytes."""
import sys, time, torch
sys.path[:0] = ['.', 'flash-molecular-dynamics_amd']
import flashmd_amd.schnet as S
from flashmd_amd import synthetic
dev = 'cuda:0'
torch.manual_seed(0)
S.USE_FUSED_ENGINE = False
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64]).to(dev)
go = S.GradientsOut(model).eval()
pos, types, ptr, masses = synthetic.batch(128, 269, 5.0)
pos, types, ptr = pos.to(dev), types.to(dev), ptr.to(dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for i in range(steps + 2):
    if i == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    d = go(S.Batch(pos.clone(), types, ptr, max_beads_per_mol=269))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"operator path: {1e3*dt:.2f} ms per force evaluation, {128/dt:.0f} force-eval*mol/s; edges {d.neighbor_list_edges if hasattr(d,'neighbor_list_edges') else 'n/a'}")
