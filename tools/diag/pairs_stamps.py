"""Phase times of pairs_kernel from the stamped throw-away build (tools/diag/pairs_stamps_patch.py -> gpurun_ab/lib_pairs_stamps.so)."""
import os, sys, ctypes
import torch
sys.path[:0] = [".", "flash-molecular-dynamics_amd"]
from flashmd_amd import _ffi
_ffi.LIB_PATH = os.path.abspath("gpurun_ab/lib_pairs_stamps.so")
from flashmd_amd import engine, synthetic
import flashmd_amd.schnet as S
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64], num_interactions=3).cuda().eval()
pos, types, ptr, _ = synthetic.batch(128, 269, 5.0)
pos, types, ptr = pos.cuda(), types.cuda(), ptr.cuda()
eng = engine.FusedSchNetEngine(model)
for _ in range(50):
    eng.energy_forces(pos, types, ptr, 269)
torch.cuda.synchronize()
lib = _ffi.lib()
buf = (ctypes.c_longlong * (512 * 16))()
lib.fmd_debug_pairs_stamps.restype = ctypes.c_int
assert lib.fmd_debug_pairs_stamps(buf) == 0
t = torch.tensor(list(buf), dtype=torch.float64).view(512, 16)[:256, :9]
d = (t[:, 1:] - t[:, :-1]) * 0.01   # 100 MHz -> us
names = ["load pos/zero", "masks", "orientation", "conflict flags", "offsets", "split rows", "fill", "dist/cutoff"]
for k, nme in enumerate(names):
    print(f"{nme:16s} mean {d[:, k].mean():6.2f} us  max {d[:, k].max():6.2f}")
print("first stamp spread (launch ramp) us:", float((t[:, 0].max() - t[:, 0].min()) * 0.01), " total span:", float((t[:, 8].max() - t[:, 0].min()) * 0.01))
