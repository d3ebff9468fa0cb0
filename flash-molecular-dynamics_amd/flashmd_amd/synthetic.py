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
