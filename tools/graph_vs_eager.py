"""Why the fp32 hipGraph replay is not faster than the eager loop: the same 300 timesteps eager or
replayed, to be run under `rocprofv3 --kernel-trace` once per mode; tools/trace_gaps.py on the two traces then gives the
per-kernel durations and the idle gaps between launches of each mode.
  python tools/graph_vs_eager.py --mode eager|graph [--precision f32|w16a16] [--steps 300]
Prints the loop's ms per step (barrier + synchronize brackets) for the mode."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
from flashmd_amd import engine, simulate, synthetic
import flashmd_amd.schnet as S

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="eager", choices=["eager", "graph"])
ap.add_argument("--precision", default="f32")
ap.add_argument("--steps", type=int, default=300)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64],
                         num_interactions=3).to(dev).eval()
eng = engine.FusedSchNetEngine(model, precision=args.precision)
pos, types, ptr, masses = synthetic.batch(128, 269, 5.0)
pos, types, ptr, masses = pos.to(dev), types.to(dev), ptr.to(dev), masses.to(dev)
beta = torch.full((128,), 1.67, device=dev)
sim = simulate.LangevinSimulation(lambda x: eng.energy_forces(x, types, ptr, 269), pos, masses, beta, 269, dt=0.004,
                                  friction=1.0)
for _ in range(20):
    sim.timestep()
if args.mode == "graph":
    sim.capture()
for _ in range(100):   # clocks settle, same loop
    sim.timestep()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    sim.timestep()
torch.cuda.synchronize()
print(f"{args.mode} {args.precision}: {1e3 * (time.perf_counter() - t0) / args.steps:.4f} ms per step over {args.steps} steps")
