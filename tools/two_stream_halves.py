"""Does running the batch as TWO independent half-batches on two HIP streams fill the launch tails (and let one half's
HBM-bound node kernels run beside the other half's pair kernels)?  Diagnostic: one engine object per half, each on its
own torch stream, against the whole batch on one stream.
  python tools/two_stream_halves.py [--precision f32|w16a16] [--n-split S]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
from flashmd_amd import engine, synthetic
import flashmd_amd.schnet as S

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--beads", type=int, default=269)
ap.add_argument("--precision", default="f32")
ap.add_argument("--n-split", type=int, default=0, help="workgroups per molecule of the HALF-batch engines (0: the library's choice)")
ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--parts", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64],
                         num_interactions=3).to(dev).eval()
pos, types, ptr, _ = synthetic.batch(args.batch, args.beads, 5.0)
pos, types, ptr = pos.to(dev), types.to(dev), ptr.to(dev)
P = args.parts
bounds = [args.batch * k // P for k in range(P + 1)]
halves = []
for k in range(P):
    lo, hi = int(ptr[bounds[k]]), int(ptr[bounds[k + 1]])
    halves.append((pos[lo:hi].contiguous(), types[lo:hi].contiguous(), (ptr[bounds[k]: bounds[k + 1] + 1] - lo).contiguous()))
whole = engine.FusedSchNetEngine(model, precision=args.precision)
engs = [engine.FusedSchNetEngine(model, precision=args.precision, n_split=args.n_split) for _ in range(P)]
streams = [torch.cuda.Stream() for _ in range(P)]


def run_whole(n):
    for _ in range(n):
        whole.energy_forces(pos, types, ptr, args.beads)


def run_halves(n):
    for _ in range(n):
        for k in range(P):
            with torch.cuda.stream(streams[k]):
                engs[k].energy_forces(*halves[k], args.beads)


def timed(fn, n):
    fn(100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


e, f = whole.energy_forces(pos, types, ptr, args.beads)
fs = []
for k in range(P):
    with torch.cuda.stream(streams[k]):
        fs.append(engs[k].energy_forces(*halves[k], args.beads)[1])
torch.cuda.synchronize()
fh = torch.cat(fs)
print(f"halves vs whole: forces {float((fh - f).norm() / f.norm()):.2e}")
for rnd in range(2):
    tw = timed(run_whole, args.reps)
    th = timed(run_halves, args.reps)
    print(f"{args.precision} n_split={args.n_split}: whole batch {tw:.4f} ms per evaluation | {P} parts on {P} streams {th:.4f} ms  ({100 * (tw / th - 1):+.1f} %)")
