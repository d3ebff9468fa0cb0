"""Diagnostics: step time of the bench loop with and without the per-kernel event pairs."""
import sys, time, ctypes, torch
sys.path[:0] = ['.', 'flash-molecular-dynamics_amd']
import flashmd_amd.schnet as S
from flashmd_amd import engine, simulate, synthetic
import bench
dev = 'cuda:0'
torch.manual_seed(0)
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64]).to(dev).eval()
eng = engine.FusedSchNetEngine(model)
pos, types, ptr, masses = synthetic.batch(128, 269, 5.0)
pos, types, ptr, masses = pos.to(dev), types.to(dev), ptr.to(dev), masses.to(dev)
beta = torch.full((128,), 1.67, device=dev)
nk = len(eng.kernel_names())
steps = 40
ev = bench.HipEvents(2 * nk * steps)
state = {'i': -1, 'use': False}
def ff(x):
    i = state['i']
    if state['use'] and 0 <= i < steps:
        e = (ctypes.c_void_p * (2 * nk)).from_buffer(ev.ev, i * 2 * nk * ctypes.sizeof(ctypes.c_void_p))
        return eng.energy_forces(x, types, ptr, 269, events=e)
    return eng.energy_forces(x, types, ptr, 269)
sim = simulate.LangevinSimulation(ff, pos, masses, beta, 269)
for _ in range(10): sim.timestep()
for use in (False, True, False, True):
    state['use'] = use
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        state['i'] = i; sim.timestep()
    torch.cuda.synchronize()
    print('events' if use else 'plain ', 1e3 * (time.perf_counter() - t0) / steps, 'ms/step')
