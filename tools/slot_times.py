"""Per-launch durations of one force evaluation, in launch order (HIP events through the timed C-ABI entry), mean over
--reps evaluations.  python tools/slot_times.py [--batch B] [--n-split S] [--precision f32|w16a16]"""
import argparse, ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
from bench import HipEvents
from flashmd_amd import _ffi
if os.environ.get("FMD_DIAG_LIB"):   # a diagnostic build of the library (tools/diag/*.sh): this tool only, never the package
    _ffi.LIB_PATH = os.path.abspath(os.environ["FMD_DIAG_LIB"])
from flashmd_amd import engine, synthetic
import flashmd_amd.schnet as S

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--beads", type=int, default=269)
ap.add_argument("--n-split", type=int, default=0)
ap.add_argument("--precision", default="f32")
ap.add_argument("--num-rbf", type=int, default=50)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--replicas", action="store_true", help="every molecule a copy of the first one (no size spread)")
ap.add_argument("--no-stash", action="store_true", help="FMD_FLAG_NO_FILTER_STASH: the backward kernels recompute filter layer 1")
args = ap.parse_args()
dev = torch.device("cuda:0")
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), args.num_rbf), S.CosineCutoff(0.0, 5.0), [128, 64],
                         num_interactions=3).to(dev).eval()
eng = engine.FusedSchNetEngine(model, n_split=args.n_split, precision=args.precision, filter_stash=not args.no_stash)
names = eng.kernel_names()
pos, types, ptr, _ = synthetic.batch(args.batch, args.beads, 5.0)
if args.replicas:
    pos = pos[: args.beads].repeat(args.batch, 1)
pos, types, ptr = pos.to(dev), types.to(dev), ptr.to(dev)
nk = len(names)
evs = HipEvents(2 * nk * args.reps)
for _ in range(100):   # (past the clock ramp: tools/clock_timeline.py)
    eng.energy_forces(pos, types, ptr, args.beads)
for i in range(args.reps):
    ev = (ctypes.c_void_p * (2 * nk)).from_buffer(evs.ev, i * 2 * nk * ctypes.sizeof(ctypes.c_void_p))
    eng.energy_forces(pos, types, ptr, args.beads, events=ev)
torch.cuda.synchronize()
tot = 0.0
for k, name in enumerate(names):
    ms = sum(evs.elapsed_ms(i * 2 * nk + 2 * k, i * 2 * nk + 2 * k + 1) for i in range(args.reps)) / args.reps
    tot += ms
    print(f"{k:2d} {name:18s} {1e3 * ms:8.1f} us")
print(f"   sum                {1e3 * tot:8.1f} us   pairs {int(eng.status[2])}")
