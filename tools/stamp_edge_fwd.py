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
buf = torch.zeros(256*4*8, dtype=torch.int64, device=dev)
_ffi.lib().fmd_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
eng.energy_forces(pos, types, ptr, 269)
torch.cuda.synchronize()
_ffi.lib().fmd_debug_set_stamp_buffer(None)
b = buf.view(256,4,8).double().cpu()
names=['C: L1','C: wait B','C: message+fill+fetch','C: wait A','P: compute','P: wait B','P: put','P: wait A'] if not os.environ.get('FMD_FWD_4WAVE') else ['tanh+write','barrierA','L1','message','barrierB','gather-issue','L0','-']
tot = b.sum(-1).mean()
print('mean cycles per wave', tot.item(), 'tiles ~130 -> per tile', tot.item()/130)
for k,nm in enumerate(names): print(f'{nm:10s} {b[...,k].mean().item()/130:10.0f} cycles/tile   {100*b[...,k].mean().item()/tot.item():5.1f}%   (wave0 {b[:,0,k].mean().item()/130:.0f}, wave3 {b[:,3,k].mean().item()/130:.0f})')
