"""Two-product variants of the split-operand contractions: force error against the
reference's fp64 vectors and engine time per force evaluation, with one of the small products of a filter-layer
contraction left out.  Diagnostic build only (make -C csrc stamps); the product library has no such switch.
  python tools/two_product.py
L0: hidden = W0 . gaussians (K = 50), also Wc in the backward kernel;  L1: filter = hidden . W1 (K = 128)."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
import torch  # noqa: E402
from flashmd_amd import _ffi  # noqa: E402

_ffi.LIB_PATH = os.path.join(ROOT, "flash-molecular-dynamics_amd", "csrc", "build_stamps", "libflashmd_hip_stamps.so")
import flashmd_amd.schnet as S  # noqa: E402
from flashmd_amd import engine, synthetic  # noqa: E402
from oracle import schnet_oracle as so  # noqa: E402

dev = "cuda:0"
p = so.params_from_npz(os.path.join(ROOT, "tests", "golden", "weights_L3_R50.npz"))
z = np.load(os.path.join(ROOT, "tests", "golden", "case_1enh_2x269.npz"))
model = S.model_from_params(p).to(dev).eval()
eng = engine.FusedSchNetEngine(model)
pos, types, ptr = (torch.from_numpy(z[k]).to(dev) for k in ("pos", "types", "ptr"))
bp, bt, bq, _ = synthetic.batch(128, 269, 5.0)
bp, bt, bq = bp.to(dev), bt.to(dev), bq.to(dev)
lib = _ffi.lib()
lib.fmd_debug_set_stamps.restype = ctypes.c_int
f_ref = z["forces_ref64"]
VARIANTS = [("three products (product path)", 0, 0),
            ("L0 without weight-lo x gaussian-hi", 2, 0), ("L0 without weight-hi x gaussian-lo", 1, 0),
            ("L1 without hidden-lo x weight-hi", 0, 2), ("L1 without hidden-hi x weight-lo", 0, 1),
            ("L0 and L1 without the activations' lo words", 1, 2), ("L0 and L1 without the weights' lo words", 2, 1)]
for name, d0, d1 in VARIANTS:
    lib.fmd_debug_set_stamps(None, -2, 0, (d0 << 2) | (d1 << 4))
    e, f = eng.energy_forces(pos, types, ptr, 269)
    err = float(np.linalg.norm(f.double().cpu().numpy() - f_ref) / np.linalg.norm(f_ref))
    for _ in range(3):
        eng.energy_forces(bp, bt, bq, 269)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.energy_forces(bp, bt, bq, 269)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{name:52s} forces vs reference fp64 {err:.2e}   force evaluation of 128 x 269: {ms:.3f} ms")
lib.fmd_debug_set_stamps(None, -1, 0, 0)
