"""Diagnostics: per-phase cycle shares of the backward split pair kernel (middle block), from the
stamped build.  Not part of any timed run."""
import sys, os, ctypes, torch
sys.path[:0]=['.', 'flash-molecular-dynamics_amd']
import flashmd_amd.schnet as S
from flashmd_amd import engine, synthetic, _ffi
dev='cuda:0'
torch.manual_seed(0)
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64]).to(dev)
eng = engine.FusedSchNetEngine(model)
pos, types, ptr, masses = synthetic.batch(128, 269)
pos, types, ptr = pos.to(dev), types.to(dev), ptr.to(dev)
for _ in range(3): eng.energy_forces(pos, types, ptr, 269)
buf = torch.zeros(256*4*16, dtype=torch.int64, device=dev)
_ffi.lib().fmd_debug_set_stamp_buffer_bwd(ctypes.c_void_p(buf.data_ptr()))
eng.energy_forces(pos, types, ptr, 269)
torch.cuda.synchronize()
_ffi.lib().fmd_debug_set_stamp_buffer_bwd(None)
b = buf.view(256,4,16).double().cpu()
names=['C: L1+gW math+gx1 RMW','C: wait B','C: gW write','C: wait C','C: fill+fetch','C: wait D','C: wait A','-',
       'P: compute','P: wait B','P: wait C','P: u+reduce','P: wait D','P: put+rotate','P: wait A','-']
tc = b[...,:8].sum(-1).mean().item()/130
print('tile cycles (consumer view)', tc)
for k,nm in enumerate(names):
    if nm != '-': print(f'{nm:26s} {b[...,k].mean().item()/130:8.0f}')
