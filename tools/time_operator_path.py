"""Diagnostics: throughput of the operator path (the drop-in Python operators, [E,F] tensors in HBM) on the bench
batch, for comparison with the fused engine; also what a molecule above the engine's size limit costs."""
import sys, time, torch
sys.path[:0] = ['.', 'flash-molecular-dynamics_amd']
import flashmd_amd.schnet as S
from flashmd_amd import synthetic
dev = 'cuda:0'
torch.manual_seed(0)
for beads, n_mol in ((269, 128), (300, 115)):
    model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64]).to(dev)
    go = S.GradientsOut(model).eval()
    pos, types, ptr, masses = synthetic.batch(n_mol, beads, 5.0)
    pos, types, ptr = pos.to(dev), types.to(dev), ptr.to(dev)
    for engine_on in (False, True):
        S.USE_FUSED_ENGINE = engine_on
        def step():
            d = go(S.Batch(pos.clone(), types, ptr, max_beads_per_mol=beads))
            return d.out["SchNet"]["forces"]
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{beads} beads x {n_mol}: engine gate {'on ' if engine_on else 'off'} {1e3*dt:8.2f} ms/eval  {n_mol/dt:9.0f} force-eval*mol/s")
