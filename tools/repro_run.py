"""Two identical Langevin runs of the bench batch, compared bit for bit after every block of steps, plus sanity
of the trajectory (finite, bounded kinetic temperature).  A race in any of the engine's kernels (two-workgroup
finish / pairs, slab sums) would show up as a difference between the runs.

  python tools/repro_run.py [steps] [precision]
"""
import sys
import torch

sys.path[:0] = [".", "flash-molecular-dynamics_amd"]
import flashmd_amd.schnet as S
from flashmd_amd import engine, simulate, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
precision = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64]).to(dev)
B, A = 128, 269
pos, types, ptr, masses = synthetic.batch(B, A, 5.0)
pos, types, ptr, masses = pos.to(dev), types.to(dev), ptr.to(dev), masses.to(dev)
beta = torch.full((B,), 1.67, device=dev)


def run():
    eng = engine.FusedSchNetEngine(model, precision=precision)
    sim = simulate.LangevinSimulation(lambda x: eng.energy_forces(x, types, ptr, A), pos, masses, beta, A,
                                      dt=0.004, friction=1.0, seed=7)
    snaps = []
    for t in range(steps):
        sim.timestep()
        if (t + 1) % 100 == 0:
            snaps.append((sim.x.clone(), sim.v.clone(), sim.energy.clone()))
    assert int(eng.status[0]) == 0
    return snaps


a, b = run(), run()
for k, ((xa, va, ea), (xb, vb, eb)) in enumerate(zip(a, b)):
    same = torch.equal(xa, xb) and torch.equal(va, vb) and torch.equal(ea, eb)
    ke = 0.5 * (masses[:, None] * va * va).sum() / (1.5 * B * A) * 1.67  # kT units: ~1 at equilibrium
    print(f"step {100 * (k + 1):5d}: identical {same}  finite {bool(torch.isfinite(xa).all())}  "
          f"<KE>/(3/2 kT) {float(ke):.3f}  mean energy {float(ea.mean()):.4f}")
    assert same and torch.isfinite(xa).all()
print("OK")
