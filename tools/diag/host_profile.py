import cProfile, pstats, io, os, sys, time
import torch
sys.path[:0] = [".", "flash-molecular-dynamics_amd"]
from flashmd_amd import engine, simulate, synthetic
import flashmd_amd.schnet as S
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64], num_interactions=3).to(dev).eval()
B, n = 2, 20
pos, types, ptr, masses = synthetic.batch(B, n, 5.0)
pos, types, ptr, masses = pos.to(dev), types.to(dev), ptr.to(dev), masses.to(dev)
beta = torch.full((B,), 1.67, device=dev)
for prec in ("f32",):
    eng = engine.FusedSchNetEngine(model, precision=prec)
    go = S.GradientsOut(model).eval()
    model._engine = eng
    batch_obj = S.Batch(pos, types, ptr, max_beads_per_mol=n)
    def via_go(x):
        batch_obj.pos, batch_obj.out = x, {}
        d = go(batch_obj)
        return d.out["SchNet"]["energy"], d.out["SchNet"]["forces"]
    def direct(x):
        return eng.energy_forces(x, types, ptr, n)
    for name, fn in (("direct", direct), ("GradientsOut", via_go)):
        sim = simulate.LangevinSimulation(fn, pos, masses, beta, n)
        for _ in range(300): sim.timestep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000): sim.timestep()
        torch.cuda.synchronize()
        print(f"{name}: {1e3*(time.perf_counter()-t0)/2000:.4f} ms per timestep (tiny batch: host bound)")
        pr = cProfile.Profile(); pr.enable()
        for _ in range(2000): sim.timestep()
        pr.disable(); torch.cuda.synchronize()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
