"""Integrator halves, replica sharding and parallel-tempering exchange around the force path.

Mirrors the pieces of the reference's simulation layer that touch the hot path:
  * Langevin BAOAB (reference simulation/langevin.py:102-177): two fused elementwise HIP kernels
    around the force call instead of ~10 torch ops;
  * throughput = second-half steps * n_sims / second-half wall time, device-synchronised at the
    halfway step and at the end (reference simulation/base.py:564-571,638-642,762-768);
  * replica sharding: molecules are independent graphs, so rank r of W simply owns molecules
    [r*B/W, (r+1)*B/W) -- no collective in the force path;
  * parallel tempering (reference simulation/parallel_tempering.py:349-481): one beta per rank,
    even/odd neighbour exchange over point-to-point RCCL send/recv (the only communication).
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _ffi
from ._ffi import call, ptr, stream


def langevin_pre(x, v, f, masses, beta_node, noise, dt: float, friction: float):
    """In place B, A, O, A (reference simulation/langevin.py:136-157); noise ~ N(0,1) [N,3]."""
    _ffi.require_gpu(x, v, f, masses, beta_node, noise)
    for t in (x, v, f, masses, beta_node, noise):
        assert t.is_contiguous() and t.dtype == torch.float32
    vscale = math.exp(-friction * dt)  # reference langevin.py:76-77
    noisescale = math.sqrt(1.0 - vscale * vscale)
    call("fmd_langevin_pre", ptr(x), ptr(v), ptr(f), ptr(masses), ptr(beta_node), ptr(noise),
         x.shape[0], float(dt), float(vscale), float(noisescale), stream())


def langevin_post(v, f, masses, dt: float):
    """In place second B half-step (reference simulation/langevin.py:169)."""
    _ffi.require_gpu(v, f, masses)
    call("fmd_langevin_post", ptr(v), ptr(f), ptr(masses), v.shape[0], float(dt), stream())


def pt_swap_local(x, v, mol_a, mol_b, accept, beta_mol, n_atoms: int):
    """In place swap of accepted replica pairs living in one batch (reference
    simulation/parallel_tempering.py:415-481): x[a]<->x[b], v[a]=v[b]*sqrt(beta_a/beta_b),
    v[b]=v[a]*sqrt(beta_b/beta_a)."""
    _ffi.require_gpu(x, v, mol_a, mol_b, accept, beta_mol)
    assert accept.dtype == torch.uint8 and mol_a.dtype == torch.int64 and mol_b.dtype == torch.int64
    call("fmd_pt_swap_local", ptr(x), ptr(v), ptr(mol_a), ptr(mol_b), ptr(accept), ptr(beta_mol),
         mol_a.shape[0], int(n_atoms), stream())


def shard_range(n_mol: int, rank: int, world: int):
    """Contiguous block of molecules owned by ``rank`` (replicas never interact: no collective)."""
    per, extra = divmod(n_mol, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


class LangevinSimulation:
    """Minimal BAOAB loop over one rank's batch (reference simulation/langevin.py:21-312 and the
    loop of simulation/base.py:501-649), forces from ``force_fn(pos) -> (energy[B], forces[N,3])``."""

    def __init__(self, force_fn, pos, masses, beta_mol, n_atoms: int, dt: float = 0.004,
                 friction: float = 1.0, seed: int = 103838):
        self.force_fn, self.dt, self.friction, self.n_atoms = force_fn, dt, friction, n_atoms
        self.x = pos.detach().clone().contiguous()
        self.masses = masses.contiguous()
        self.beta_mol = beta_mol.contiguous()
        self.beta_node = beta_mol.repeat_interleave(n_atoms).contiguous()
        self.rng = torch.Generator(device=pos.device).manual_seed(seed)
        # Maxwell-Boltzmann start (reference langevin.py:80-99)
        scale = torch.sqrt(1.0 / (self.beta_node * self.masses))[:, None]
        self.v = (torch.randn(self.x.shape, generator=self.rng, device=pos.device) * scale).contiguous()
        self.noise = torch.empty_like(self.x)
        self.energy, self.f = self.force_fn(self.x)
        self.n_sims = beta_mol.numel()

    def timestep(self):
        self.noise.normal_(generator=self.rng)  # reference langevin.py:146
        langevin_pre(self.x, self.v, self.f, self.masses, self.beta_node, self.noise, self.dt,
                     self.friction)
        self.energy, self.f = self.force_fn(self.x)
        langevin_post(self.v, self.f, self.masses, self.dt)

    def simulate(self, n_timesteps: int, subroutine=None, interval: int = 0):
        """Runs the loop and returns timestep*mol/s measured over the second half
        (reference simulation/base.py:564-571,638-642,762-768)."""
        import time

        half = n_timesteps // 2
        t0 = None
        for t in range(n_timesteps):
            if t == half:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            self.timestep()
            if subroutine is not None and interval and (t + 1) % interval == 0:
                subroutine(self)
        torch.cuda.synchronize()
        return (n_timesteps - half) * self.n_sims / (time.perf_counter() - t0)
