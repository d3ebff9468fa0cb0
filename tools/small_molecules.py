"""Throughput of the fused engine on batches of SMALL molecules (many molecules per CU), per evaluation: eager calls
(the host needs 0.9-1.0 ms of Python / ctypes per 15-launch evaluation, which bounds the small batches) and the same
evaluation replayed from a hipGraph (device time): python tools/small_molecules.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
from flashmd_amd import engine, synthetic
import flashmd_amd.schnet as S
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64],
                         num_interactions=3).to(dev).eval()
eng = engine.FusedSchNetEngine(model)
for n_mol, beads in ((128, 269), (1024, 35), (4096, 35), (2048, 10), (512, 76), (256, 135)):
    pos, types, ptr, _ = synthetic.batch(n_mol, beads, 5.0)
    pos, types, ptr = pos.to(dev), types.to(dev), ptr.to(dev)
    for _ in range(3):
        eng.energy_forces(pos, types, ptr, beads)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.energy_forces(pos, types, ptr, beads)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    pairs = eng.check_status()[2]
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        e_g, f_g = eng.energy_forces(pos, types, ptr, beads)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    dg = (time.perf_counter() - t0) / 20
    eng.check_status()
    print(f"{n_mol:5d} x {beads:3d} beads: {1e3 * dt:7.3f} ms per evaluation, {n_mol / dt:9.0f} evaluation*mol/s, {pairs} pairs, "
          f"{pairs / dt / 1e9:6.2f} G pairs/s eager | hipGraph replay {1e3 * dg:7.3f} ms, {pairs / dg / 1e9:6.2f} G pairs/s")
