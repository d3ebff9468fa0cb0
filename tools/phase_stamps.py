"""Per-phase cycle totals of a pair kernel's tile loop, from the diagnostic build (`make -C csrc stamps`):
  python tools/phase_stamps.py [--fwd] [--block L] [--batch B] [--precision f32|w16a16]
Prints, for consumer waves (0-3) and producer waves (4-7), the mean cycles per tile spent in each marked phase
(see the FMD_MARK calls in csrc/fmd_engine_edge_{fwd,bwd}.hip).  Diagnostics only."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
import torch  # noqa: E402
from flashmd_amd import _ffi  # noqa: E402

_ffi.LIB_PATH = os.environ.get("FMD_STAMPS_LIB") or os.path.join(ROOT, "flash-molecular-dynamics_amd", "csrc", "build_stamps",
                                                                  "libflashmd_hip_stamps.so")
import flashmd_amd.schnet as S  # noqa: E402
from flashmd_amd import engine, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fwd", action="store_true")
ap.add_argument("--block", type=int, default=1)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--beads", type=int, default=269)
ap.add_argument("--precision", default="f32")
ap.add_argument("--dbg", type=int, default=0, help="bit 0: consumers skip their A-B work, bit 1: producers skip staging (timing only, wrong results)")
args = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64],
                         num_interactions=3).to(dev).eval()
eng = engine.FusedSchNetEngine(model, precision=args.precision)
pos, types, ptr, _ = synthetic.batch(args.batch, args.beads, 5.0)
pos, types, ptr = pos.to(dev), types.to(dev), ptr.to(dev)
n_wg = 2 * args.batch if args.batch < 256 else args.batch
NST = 12
buf = torch.zeros(n_wg * 8 * NST * 2, dtype=torch.int64, device=dev)
lib = _ffi.lib()
lib.fmd_debug_set_stamps.restype = ctypes.c_int
for _ in range(3):
    eng.energy_forces(pos, types, ptr, args.beads)
lib.fmd_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()), args.block, 1 if args.fwd else 0, args.dbg)
eng.energy_forces(pos, types, ptr, args.beads)
torch.cuda.synchronize()
lib.fmd_debug_set_stamps(None, -1, 0, 0)
st = eng.status.tolist()
tiles_per_wg = (st[2] / args.batch * 1.008 / 32) / (n_wg / args.batch)   # pairs + ~0.8 % pad entries (tools/pad_stats.py)
t = buf.view(n_wg, 16, NST).double().cpu()   # 16 wave slots per workgroup (the forward kernel runs 12 waves)
groups = [("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8)), ("waves 8-11", slice(8, 12)), ("waves 12-15", slice(12, 16))]
for name, sl in groups:
    if t[:, sl].sum() == 0:
        continue
    m = t[:, sl].mean(dim=(0, 1)) / tiles_per_wg
    print(name, "cycles per tile:", " ".join(f"[{k}]{v:.0f}" for k, v in enumerate(m.tolist())), f" sum {m.sum():.0f}")
print(f"tiles per workgroup ~{tiles_per_wg:.1f}")
tot = t[:, 0].sum(dim=1)   # wave 0 of every workgroup: cycles from the first barrier to the last tile
print(f"per-workgroup cycles: mean {tot.mean():.0f}  min {tot.min():.0f}  max {tot.max():.0f}  (max / mean {tot.max() / tot.mean():.3f})")
