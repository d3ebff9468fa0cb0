"""Energies and forces of fixed batches through the library named by FMD_DIAG_LIB (default: the in-tree one), saved to
a file: bitwise comparison of two builds.  python tools/dump_forces.py out.pt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
from flashmd_amd import _ffi
if os.environ.get("FMD_DIAG_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["FMD_DIAG_LIB"])
from flashmd_amd import engine, synthetic
import flashmd_amd.schnet as S

torch.manual_seed(0)
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64], num_interactions=3).cuda().eval()
out = {}
for prec in ("f32", "w16a16"):
    for B, n in ((128, 269), (5, 400), (700, 35), (2048, 10), (3, 37)):
        pos, types, ptr, _ = synthetic.batch(B, n, 5.0)
        eng = engine.FusedSchNetEngine(model, precision=prec)
        e, f = eng.energy_forces(pos.cuda(), types.cuda(), ptr.cuda(), n)
        out[f"{prec}_{B}x{n}"] = (e.cpu(), f.cpu())
torch.save(out, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    for k, (e, f) in out.items():
        print(k, "energies equal", bool(torch.equal(e, ref[k][0])), "forces equal", bool(torch.equal(f, ref[k][1])),
              "max |df|", float((f - ref[k][1]).abs().max()))
