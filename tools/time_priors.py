"""Diagnostics: time of the prior kernels on a 1ENH-sized batch (128 chains x 269 beads), per kind."""
import sys, torch
sys.path[:0] = ['.', 'flash-molecular-dynamics_amd']
from flashmd_amd import prior, synthetic
dev = 'cuda:0'
n_mol, A, T = 128, 269, 5
pos, types, ptr = synthetic.chain_batch(2, A, T)
pos = torch.cat([pos[:A] + 0.01 * m for m in range(n_mol)]); types = types[:A].repeat(n_mol)
ptr = torch.arange(0, (n_mol + 1) * A, A)
topo = synthetic.chain_topology(n_mol, A, repulsion_stride=1)  # every non-bonded pair: 35k per molecule
g = torch.Generator().manual_seed(3)
tables = {"bonds": {"k": 2 + torch.rand(T, T, generator=g), "x_0": 3.6 + 0.4 * torch.rand(T, T, generator=g)},
          "angles": {"k": 1 + torch.rand(T, T, T, generator=g), "x_0": torch.rand(T, T, T, generator=g) - 0.5},
          "repulsion": {"sigma": 3 + torch.rand(T, T, generator=g)},
          "dihedrals": {"v_0": torch.rand(T, T, T, T, generator=g), "k1s": torch.rand(6, T, T, T, T, generator=g) - 0.5,
                        "k2s": torch.rand(6, T, T, T, T, generator=g) - 0.5}}
P, Q = pos.to(dev), ptr.to(dev)
for k in ("bonds", "angles", "dihedrals", "repulsion"):
    sp = prior.StaticPriors.from_tables({k: (tables[k], topo[k])}, types, ptr).to(dev)
    for _ in range(3): sp.energy_forces(P, Q)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e = torch.zeros(n_mol, device=dev); f = torch.zeros_like(P)
    e0.record()
    for _ in range(20): sp.add_energy_forces(P, Q, e, f)
    e1.record(); torch.cuda.synchronize()
    m, prm, bp, ent, pair = sp.terms[k]
    pl = pair is not None and ent.numel() > 16 * P.shape[0]
    nb = ent.numel() * ((4 + prm.shape[0] * 4) if pl else (4 + m.shape[0] * 8 + prm.shape[0] * 4))
    ms = e0.elapsed_time(e1) / 20
    print(f"{k:10s} terms {m.shape[1]:8d}  {ms*1e3:8.1f} us  entry bytes {nb/1e6:7.2f} MB -> {nb/ms/1e6:7.1f} GB/s")
