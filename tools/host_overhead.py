"""Is the timestep loop host-bound?  For the direct engine call and for the reference-shaped entry
GradientsOut(model)(Batch): host time to ENQUEUE a step (no sync inside the loop) against the device time of the
same steps.  python tools/host_overhead.py [--steps 300]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
from flashmd_amd import engine, simulate, synthetic
import flashmd_amd.schnet as S

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--beads", type=int, default=269)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--precision", default="f32")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64],
                         num_interactions=3).to(dev).eval()
eng = engine.FusedSchNetEngine(model, precision=args.precision)
pos, types, ptr, masses = synthetic.batch(args.batch, args.beads, 5.0)
pos, types, ptr, masses = pos.to(dev), types.to(dev), ptr.to(dev), masses.to(dev)
beta = torch.full((args.batch,), 1.67, device=dev)


def direct(x):
    return eng.energy_forces(x, types, ptr, args.beads)


go = S.GradientsOut(model).eval()
model._engine = eng
batch_obj = S.Batch(pos, types, ptr, max_beads_per_mol=args.beads)


def via_go(x):
    batch_obj.pos, batch_obj.out = x, {}
    d = go(batch_obj)
    return d.out["SchNet"]["energy"], d.out["SchNet"]["forces"]


def measure(name, fn, per_step):
    sim = simulate.LangevinSimulation(fn, pos, masses, beta, args.beads)
    x0, v0 = sim.x.clone(), sim.v.clone()
    for _ in range(200):
        per_step(sim)
    sim.x.copy_(x0); sim.v.copy_(v0)
    sim.energy, sim.f = fn(sim.x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        per_step(sim)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{name:34s} enqueue {1e3 * t_enq / args.steps:7.3f} ms/step   complete {1e3 * t_all / args.steps:7.3f} ms/step")


def measure_graph(name, fn, k):
    sim = simulate.LangevinSimulation(fn, pos, masses, beta, args.beads).capture(steps_per_graph=k)
    for _ in range(200 // k):
        sim.timestep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps // k):
        sim.timestep()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    n = args.steps // k * k
    print(f"{name:34s} enqueue {1e3 * t_enq / n:7.3f} ms/step   complete {1e3 * t_all / n:7.3f} ms/step")


measure("force only, direct engine", direct, lambda s: s.force_fn(s.x))
measure("force only, GradientsOut", via_go, lambda s: s.force_fn(s.x))
measure("BAOAB step, direct engine", direct, lambda s: s.timestep())
measure("BAOAB step, GradientsOut", via_go, lambda s: s.timestep())
# the host alone: same python, launches replaced by nothing is not possible; instead the cost of the pieces
t0 = time.perf_counter()
for _ in range(2000):
    torch.cuda.current_stream().cuda_stream
print(f"current_stream().cuda_stream        {1e6 * (time.perf_counter() - t0) / 2000:7.2f} us")
t0 = time.perf_counter()
for _ in range(2000):
    torch.empty(args.batch * args.beads, 3, dtype=torch.float32, device=dev)
print(f"torch.empty                         {1e6 * (time.perf_counter() - t0) / 2000:7.2f} us")
measure_graph("timestep, hipGraph (1 step/graph)", via_go, 1)
measure_graph("timestep, hipGraph (4 steps/graph)", via_go, 4)
