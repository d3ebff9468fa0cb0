"""Synthetic 1ENH-shaped batches for benchmarks: per molecule, ``n_beads``
points uniform in a ball of radius 1.44*rc*(n/269)^(1/3) with no two points closer than 0.15*rc
(about 55 neighbours per bead at n = 269, rc = 5, the density of the reference's 1ENH batch),
one bead-type sequence shared by every molecule, constant masses."""
from __future__ import annotations

import torch


def molecule(n_beads: int, cutoff: float, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    rs = 1.44 * cutoff * (n_beads / 269.0) ** (1.0 / 3.0)
    dmin2 = (0.15 * cutoff) ** 2
    pts = torch.empty(0, 3)
    while pts.shape[0] < n_beads:
        cand = (torch.rand(4 * n_beads, 3, generator=g) * 2 - 1) * rs
        cand = cand[(cand * cand).sum(1) <= rs * rs]
        for c in cand:  # sequential acceptance keeps the minimum-distance rule exact
            if pts.shape[0] == n_beads:
                break
            if pts.shape[0] == 0 or float(((pts - c) ** 2).sum(1).min()) >= dmin2:
                pts = torch.cat([pts, c[None]])
    return pts.float()


def batch(n_mol: int, n_beads: int = 269, cutoff: float = 5.0, first_seed: int = 1000):
    """(pos[N,3] f32, atom_types[N] i64, ptr[B+1] i64, masses[N] f32)"""
    pos = torch.cat([molecule(n_beads, cutoff, first_seed + m) for m in range(n_mol)])
    g = torch.Generator().manual_seed(7)
    types = torch.randint(1, 25, (n_beads,), generator=g).repeat(n_mol)
    ptr = torch.arange(n_mol + 1, dtype=torch.int64) * n_beads
    masses = torch.full((n_mol * n_beads,), 12.0)
    return pos, types, ptr, masses


def chain_batch(n_mol: int, n_beads: int, n_types: int = 5, first_seed: int = 2000, bond: float = 3.8):
    """Self-avoiding-ish random-walk chains (bond length ``bond``) for the classical prior terms: positions
    [N,3], types [N] (one sequence shared by all molecules), ptr [n_mol+1]."""
    g = torch.Generator().manual_seed(11)
    types = torch.randint(0, n_types, (n_beads,), generator=g)
    pos = []
    for m in range(n_mol):
        gm = torch.Generator().manual_seed(first_seed + m)
        x = torch.zeros(n_beads, 3)
        d = torch.nn.functional.normalize(torch.randn(3, generator=gm), dim=0)
        for i in range(1, n_beads):
            while True:  # turn by a bounded random angle: bond angles stay away from 0 and pi
                nd = torch.nn.functional.normalize(d + 0.9 * torch.randn(3, generator=gm), dim=0)
                if 0.1 < float(nd @ d) < 0.95:
                    break
            d = nd
            x[i] = x[i - 1] + bond * d
        pos.append(x)
    ptr = torch.arange(0, (n_mol + 1) * n_beads, n_beads, dtype=torch.int64)
    return torch.cat(pos), types.repeat(n_mol), ptr


def chain_topology(n_mol: int, n_beads: int, repulsion_min_sep: int = 4, repulsion_stride: int = 3):
    """index mappings [order, M] (global bead indices, molecule-major) of a linear chain: bonds, angles,
    dihedrals, and a thinned set of non-bonded pairs for the repulsion."""
    out = {}
    i = torch.arange(n_beads)
    base = {"bonds": torch.stack([i[:-1], i[1:]]), "angles": torch.stack([i[:-2], i[1:-1], i[2:]]),
            "dihedrals": torch.stack([i[:-3], i[1:-2], i[2:-1], i[3:]])}
    a, b = torch.triu_indices(n_beads, n_beads, repulsion_min_sep)
    keep = (a + b) % repulsion_stride == 0
    base["repulsion"] = torch.stack([a[keep], b[keep]])
    for k, mp in base.items():
        out[k] = torch.cat([mp + m * n_beads for m in range(n_mol)], dim=1).contiguous()
    return out
