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


def verlet_pre(x, v, f, masses, dt: float):
    """in place: velocity-Verlet half kick + drift (reference simulation/velocity_verlet.py:38-41)"""
    _ffi.require_gpu(x, v, f, masses)
    call("fmd_verlet_pre", ptr(x), ptr(v), ptr(f), ptr(masses), x.shape[0], float(dt), stream())


def overdamped_step(x, f, beta_node, noise, dt: float, friction: float):
    """in place: x <- x + f D dt + sqrt(2 D dt) noise, D = 1 / (beta friction) (reference simulation/langevin.py:370-372,
    417-421)"""
    _ffi.require_gpu(x, f, beta_node, noise)
    for t in (x, f, beta_node, noise):
        assert t.is_contiguous() and t.dtype == torch.float32
    call("fmd_overdamped_step", ptr(x), ptr(f), ptr(beta_node), ptr(noise), x.shape[0], float(dt), float(friction),
         stream())


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
    loop of simulation/base.py:501-649), forces from ``force_fn(pos) -> (energy[B], forces[N,3])``.
    ``use_hip`` (default: the MLCG_USE_* gates) selects the fused HIP halves; with the gates off
    (--disable_optim) the same update runs as plain torch ops, which is what runs on the CPU."""

    def __init__(self, force_fn, pos, masses, beta_mol, n_atoms: int, dt: float = 0.004,
                 friction: float = 1.0, seed: int = 103838, use_hip: Optional[bool] = None):
        from . import schnet

        self.use_hip = schnet.any_hip_gate_on() if use_hip is None else use_hip
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
        self.t_init = 0           # chunk index a resumed run continues from (load_checkpoint); 0 = a fresh run
        self._graph = None        # hipGraph of the timestep (capture()), replayed by timestep()
        self._graph_steps = 1

    def draw_noise(self):
        """fills self.noise with N(0,1) (reference langevin.py:146: ``_noise_buffer.normal_(generator=self.rng)``)"""
        self.noise.normal_(generator=self.rng)

    def capture(self, steps_per_graph: int = 1, warmup: int = 2):
        """Record ``steps_per_graph`` timesteps -- noise draw, first BAOAB half, every launch of the force providers,
        second half -- as ONE hipGraph; ``timestep()`` then replays it (and advances ``steps_per_graph`` steps per
        call).  This build's answer to the reference's ``torch.compile`` wrapper around the model
        (simulation/base.py:357-368): shapes, workspace and weights are static, so the host's per-step work (0.8-1.0 ms
        of Python and ctypes for ~19 launches) shrinks to one graph launch.  Results are those of the un-captured loop
        to the last bit (same kernels, same order, same Philox offsets: the generator is registered with the graph).
        The force providers must already have seen this batch layout (no allocation, status read or other
        synchronisation may happen while a stream is capturing): ``warmup`` un-captured steps run first, on a copy of the
        state that is put back.  Device status words are NOT read during replays -- call ``check_status()`` at the
        run's natural synchronisation points, as ``simulate()`` does; the sticky status word keeps any failure."""
        if not (self.use_hip and self.x.is_cuda):
            raise RuntimeError("capture() needs the HIP path on a GPU")
        assert steps_per_graph >= 1
        self._graph = None
        state = (self.x.clone(), self.v.clone(), self.f.clone(), self.energy.clone(), self.rng.get_state())
        for _ in range(max(1, warmup)):
            self.timestep()
        self.check_status()                     # (also resets the providers' "evaluations since the last read" counters)
        self.x.copy_(state[0]); self.v.copy_(state[1])
        # forces and energies live in fixed buffers from here on: a replay reads what the previous replay wrote
        self.f, self.energy = state[2], state[3]
        self.rng.set_state(state[4])
        g = torch.cuda.CUDAGraph()
        g.register_generator_state(self.rng)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(steps_per_graph):
                self._captured_step()
        self._graph, self._graph_steps = g, steps_per_graph
        return self

    def _captured_step(self):
        """one timestep as it is recorded into the graph: forces and energies go to their fixed buffers"""
        self.noise.normal_(generator=self.rng)
        langevin_pre(self.x, self.v, self.f, self.masses, self.beta_node, self.noise, self.dt, self.friction)
        e, f = self.force_fn(self.x)
        self.energy.copy_(e)
        self.f.copy_(f)
        langevin_post(self.v, self.f, self.masses, self.dt)

    def release_graph(self):
        """back to the un-captured loop (e.g. before the batch layout or the force providers change)"""
        self._graph = None
        self._graph_steps = 1

    def timestep(self):
        if self._graph is not None:   # ``_graph_steps`` timesteps per call
            self._graph.replay()
            return
        self.draw_noise()
        if self.use_hip:
            langevin_pre(self.x, self.v, self.f, self.masses, self.beta_node, self.noise, self.dt,
                         self.friction)
            self.energy, self.f = self.force_fn(self.x)
            langevin_post(self.v, self.f, self.masses, self.dt)
            return
        # PyTorch path (reference simulation/langevin.py:136-170)
        dt, m = self.dt, self.masses[:, None]
        vscale = math.exp(-self.friction * dt)
        noisescale = math.sqrt(1.0 - vscale * vscale)
        v = self.v + 0.5 * dt * self.f / m
        x = self.x + v * dt * 0.5
        v = v * vscale + noisescale * (torch.sqrt(1.0 / self.beta_node / self.masses)[:, None] * self.noise)
        self.x = x + v * dt * 0.5
        self.energy, self.f = self.force_fn(self.x)
        self.v = v + 0.5 * dt * self.f / m

    def simulate(self, n_timesteps: int, subroutine=None, interval: int = 0, writer=None):
        """Runs the loop and returns timestep*mol/s measured over the second half
        (reference simulation/base.py:564-571,638-642,762-768).  ``writer``: a TrajectoryWriter that
        receives the state every ``writer.save_interval`` steps (reference :620-636)."""
        import time

        half = n_timesteps // 2
        t0 = None
        sync = torch.cuda.synchronize if self.x.is_cuda else (lambda: None)
        # the reference writes {filename}_checkpoint_init.pt only when a run starts at t_init == 0
        # (simulation/base.py:538): a run resumed through load_checkpoint() must not overwrite it with mid-run state
        if writer is not None and self.t_init == 0 and writer.index == 0 and writer.count == 0:
            writer.write_initial_checkpoint(self)
        assert self._graph is None or self._graph_steps == 1, "simulate() counts single timesteps: capture(1)"
        for t in range(n_timesteps):
            if t == half:
                sync()
                t0 = time.perf_counter()
            self.timestep()
            if writer is not None and (t + 1) % writer.save_interval == 0:
                writer.save(self, t)
            if subroutine is not None and interval and (t + 1) % interval == 0:
                subroutine(self)
        sync()
        rate = (n_timesteps - half) * self.n_sims / (time.perf_counter() - t0)
        if writer is not None:
            writer.close()
        self.check_status()
        return rate

    def check_status(self):
        """Read the device status words of the force providers (a sync; called where one happens anyway: end of a
        run, chunk export, exchange bookkeeping).  ``force_fn`` may expose ``check_status()`` itself (SumForces does)
        or be a bound method of an object that does (FusedSchNetEngine.energy_forces)."""
        fn = getattr(self.force_fn, "check_status", None) or getattr(getattr(self.force_fn, "__self__", None),
                                                                     "check_status", None)
        if fn is not None:
            fn()

    def kinetic_energy(self):
        """per molecule, 0.5 sum m v^2 (reference simulation/langevin.py:266-271)"""
        ke = 0.5 * self.masses * (self.v * self.v).sum(-1)
        return ke.view(self.n_sims, self.n_atoms).sum(-1)

    def load_checkpoint(self, path: str):
        """Resume positions and velocities from a ``{filename}_checkpoint_XXXX.pt`` written by TrajectoryWriter
        (or by the reference: same keys, simulation/base.py:1357-1418); forces are re-evaluated.  Returns the
        chunk index the run continues from."""
        ck = torch.load(path, weights_only=True)
        self.x = ck["pos"].to(self.x.device, self.x.dtype).contiguous()
        self.v = ck["velocities"].to(self.v.device, self.v.dtype).contiguous()
        if self._graph is not None:   # the captured graph works on the old buffers
            self.release_graph()
        self.energy, self.f = self.force_fn(self.x)
        self.t_init = int(ck["current_timestep"])
        return self.t_init


class TrajectoryWriter:
    """Row N3: the reference's on-disk trajectory layout (simulation/base.py:1294-1453, langevin.py:236-291):
    every ``save_interval`` steps a frame is kept, every ``export_interval`` steps a chunk is written as

      {filename}_coords_XXXX.npy          [n_sims, frames, n_atoms, 3]
      {filename}_forces_XXXX.npy          [n_sims, frames, n_atoms, 3]   (save_forces)
      {filename}_potential_XXXX.npy       [n_sims, frames]               (save_energies)
      {filename}_kineticenergy_XXXX.npy   [n_sims, frames]               (save_energies)
      {filename}_checkpoint_XXXX.pt       {"pos", "velocities", "current_timestep", "export_interval",
                                           "save_interval", "log_interval"}   (create_checkpoints)

    Frames are staged in device memory ([frames, n_sims, n_atoms, 3] per quantity) and leave the GPU once per
    chunk: one device-to-host copy on a side stream into pinned memory, written by a worker thread, so the
    timestep loop never waits for the disk.  The divergence check of the reference (position spread > 1000x the
    initial one, or NaN) is evaluated on the device per saved frame and read at export time."""

    def __init__(self, filename: str, sim, save_interval: int, export_interval: int, save_forces: bool = False,
                 save_energies: bool = False, create_checkpoints: bool = False, log_interval: int = 0, exchange=None,
                 write_acceptance: bool = True):
        """``exchange``: a Local- / DistributedReplicaExchange whose acceptance matrix is exported with every chunk as
        ``{filename}_acceptance_XXXX.npy`` (float32 [n_betas, n_betas], accepted above the diagonal, rejected below) and
        reset, as the reference's ``save_exchanges`` does (simulation/parallel_tempering.py:504-517).  With one beta per
        rank the matrix is summed over the ranks at export (a collective: every rank's writer must export at the same
        step); ``write_acceptance=False`` on all ranks but one keeps a single file."""
        assert export_interval % save_interval == 0 and export_interval >= save_interval
        self.filename, self.save_interval, self.export_interval = filename, save_interval, export_interval
        self.save_forces, self.save_energies, self.create_checkpoints = save_forces, save_energies, create_checkpoints
        self.log_interval = log_interval
        self.exchange, self.write_acceptance = exchange, write_acceptance
        self.frames = export_interval // save_interval
        dev, n_sims, n_atoms = sim.x.device, sim.n_sims, sim.n_atoms
        self.coords = torch.zeros(self.frames, n_sims, n_atoms, 3, device=dev)
        self.forces = torch.zeros_like(self.coords) if save_forces else None
        self.potential = torch.zeros(self.frames, n_sims, device=dev) if save_energies else None
        self.kinetic = torch.zeros(self.frames, n_sims, device=dev) if save_energies else None
        self.initial_spread = sim.x.view(n_sims, -1).std(dim=1).max()
        self.diverged = torch.zeros((), dtype=torch.bool, device=dev)
        self.index = int(getattr(sim, "t_init", 0))   # chunk counter (the XXXX of the file names); a resumed run goes on
                                                       # where its checkpoint stopped (reference simulation/base.py:1100-1102)
        self.count = 0            # frames staged in the current chunk
        self.files = []
        self._stream = torch.cuda.Stream(device=dev) if sim.x.is_cuda else None
        self._worker = None

    def save(self, sim, t: int):
        k = self.count
        x = sim.x.view(sim.n_sims, sim.n_atoms, 3)
        spread = x.reshape(sim.n_sims, -1).std(dim=1)
        self.diverged |= (spread.max() > 1e3 * self.initial_spread) | torch.isnan(spread).any()
        self.coords[k] = x
        if self.save_forces:
            self.forces[k] = sim.f.view(sim.n_sims, sim.n_atoms, 3)
        if self.save_energies:
            self.potential[k] = sim.energy
            self.kinetic[k] = sim.kinetic_energy()
        self.count += 1
        if self.count == self.frames:
            self._export(sim, t)

    def _export(self, sim, t: int):
        import threading

        import numpy as np

        self._join()
        sim.check_status()
        key = f"{self.index:04d}"
        out = {"coords": self.coords}
        if self.save_forces:
            out["forces"] = self.forces
        if self.save_energies:
            out["potential"], out["kineticenergy"] = self.potential, self.kinetic
        ck = None
        if self.create_checkpoints:
            ck = {"pos": sim.x, "velocities": sim.v}
        host, event = {}, None
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                for name, tns in out.items():
                    host[name] = torch.empty(tns.transpose(0, 1).shape, dtype=tns.dtype, pin_memory=True)
                    host[name].copy_(tns.transpose(0, 1), non_blocking=True)  # [n_sims, frames, ...]
                if ck is not None:
                    ck = {n_: v_.to("cpu", non_blocking=True) for n_, v_ in ck.items()}
                flag = self.diverged.to("cpu", non_blocking=True)
                event = torch.cuda.Event()
                event.record(self._stream)
            # the staging buffers are rewritten by the next chunk: that must wait for the copies
            torch.cuda.current_stream().wait_stream(self._stream)
        else:
            host = {name: tns.transpose(0, 1).contiguous().clone() for name, tns in out.items()}
            ck = None if ck is None else {n_: v_.clone() for n_, v_ in ck.items()}
            flag = self.diverged.clone()
        index = self.index
        acceptance = None
        if self.exchange is not None:   # exported BEFORE this step's exchange attempt, then reset (reference order:
            m = self.exchange.acceptance_matrix   # simulation/base.py:596-615)
            m = m() if callable(m) else m
            acceptance = m.detach().to(torch.float32).cpu().numpy().copy()
            self.exchange.reset_acceptance()

        def work():
            if event is not None:
                event.synchronize()
            if bool(flag):
                self._error = RuntimeError(f"Simulation of trajectory blew up before #timestep={t}")
                return
            for name, arr in host.items():
                path = f"{self.filename}_{name}_{key}.npy"
                np.save(path, arr.numpy())
                self.files.append(path)
            if acceptance is not None and self.write_acceptance:
                # (the reference names this file with the chunk counter AFTER write() has advanced it: the matrix of
                # chunk 0000 is {filename}_acceptance_0001.npy -- simulation/base.py:1453 runs before save_exchanges)
                path = f"{self.filename}_acceptance_{index + 1:04d}.npy"
                np.save(path, acceptance)
                self.files.append(path)
            if ck is not None:
                ck.update(current_timestep=index + 1, export_interval=self.export_interval,
                          save_interval=self.save_interval, log_interval=self.log_interval)
                path = f"{self.filename}_checkpoint_{key}.pt"
                torch.save(ck, path)
                self.files.append(path)

        self._error = None
        self._worker = threading.Thread(target=work)
        self._worker.start()
        self.index += 1
        self.count = 0

    def write_initial_checkpoint(self, sim):
        """``{filename}_checkpoint_init.pt``: the state before the first step of a fresh run (reference
        simulation/base.py:545-559), same keys as the per-chunk checkpoints, ``current_timestep`` 0."""
        if not self.create_checkpoints:
            return
        path = f"{self.filename}_checkpoint_init.pt"
        torch.save({"pos": sim.x.detach().cpu().clone(), "velocities": sim.v.detach().cpu().clone(),
                    "current_timestep": 0, "export_interval": self.export_interval,
                    "save_interval": self.save_interval, "log_interval": self.log_interval}, path)
        self.files.append(path)

    def _join(self):
        if self._worker is not None:
            self._worker.join()
            self._worker = None
            if self._error is not None:
                raise self._error

    def close(self):
        """Wait for the pending chunk (frames of an incomplete chunk are dropped, as in the reference)."""
        self._join()


def force_fn_from_state_dict(sd, atom_types, mol_ptr, topology=None, device="cuda", max_beads_per_mol: int = 0,
                             cutoff: Optional[float] = None, prior_kinds=None, **engine_kw):
    """Row N4: the force provider of a reference-format model file.  ``sd``: the ``state_dict`` of the reference's
    ``SumOut{SchNet: GradientsOut(StandardSchNet), <prior name>: GradientsOut(<prior>) ...}`` container
    (models/gradients.py:19-152; a dict, an ``.npz``, or a weights-only ``torch.load`` file -- never the pickled
    module) or of a bare / ``GradientsOut``-wrapped SchNet.  Returns ``force_fn(pos) -> (energy[B], forces[N,3])`` =
    fused SchNet engine + condensed priors (``prior.sum_out``), with ``check_status``; ``force_fn.engine`` and
    ``force_fn.priors`` expose the parts.  ``topology``: prior name -> index mapping [order, M] of the collated batch
    (needed when the file holds priors: a state_dict does not carry the topology); ``atom_types`` [N], ``mol_ptr`` [B+1]
    (CPU or device).  Hyper-parameters the weights do not encode: see ``schnet.model_from_state_dict``.
    ``prior_kinds``: prior name -> ``"raw_angles"`` / ``"shifted_impropers"`` for the two classes whose buffers a
    state_dict cannot tell from ``HarmonicAngles`` / ``HarmonicImpropers`` (``prior.prior_tables_from_state_dict``)."""
    from . import engine as _engine
    from . import prior as _prior
    from .schnet import _state_dict_from_file

    if isinstance(sd, str):
        sd = _state_dict_from_file(sd)
    eng = _engine.FusedSchNetEngine.from_state_dict(sd, device=device, cutoff=cutoff, **engine_kw)
    types_d, ptr_d = atom_types.to(device), mol_ptr.to(device)
    mb = int(max_beads_per_mol) or int((mol_ptr[1:] - mol_ptr[:-1]).max())

    def schnet_fn(pos):
        return eng.energy_forces(pos, types_d, ptr_d, mb)

    schnet_fn.check_status = eng.check_status
    parts = [schnet_fn]
    priors = None
    if _prior.prior_tables_from_state_dict(sd, prior_kinds):
        priors = _prior.StaticPriors.from_state_dict(sd, topology or {}, atom_types.cpu(), mol_ptr.cpu(),
                                                     kinds=prior_kinds).to(device)
        parts.append((priors, ptr_d))
    fn = _prior.sum_out(parts)
    fn.engine, fn.priors = eng, priors
    return fn


class NVESimulation(LangevinSimulation):
    """Microcanonical velocity-Verlet loop (reference simulation/velocity_verlet.py:12-62): same state,
    start-up (Maxwell-Boltzmann velocities) and throughput loop as the Langevin class."""

    def _captured_step(self):
        verlet_pre(self.x, self.v, self.f, self.masses, self.dt)
        e, f = self.force_fn(self.x)
        self.energy.copy_(e)
        self.f.copy_(f)
        langevin_post(self.v, self.f, self.masses, self.dt)

    def timestep(self):
        if self._graph is not None:
            self._graph.replay()
            return
        if self.use_hip:
            verlet_pre(self.x, self.v, self.f, self.masses, self.dt)
            self.energy, self.f = self.force_fn(self.x)
            langevin_post(self.v, self.f, self.masses, self.dt)
            return
        dt, m = self.dt, self.masses[:, None]
        v = self.v + 0.5 * dt * self.f / m
        self.x = self.x + dt * v
        self.energy, self.f = self.force_fn(self.x)
        self.v = v + 0.5 * dt * self.f / m


class OverdampedSimulation(LangevinSimulation):
    """Overdamped Langevin (Brownian) dynamics (reference simulation/langevin.py:315-434): no masses, no velocities;
    x <- x + F D dt + sqrt(2 D dt) noise with D = 1 / (beta friction), then the force call.  Same loop, noise draw,
    throughput definition, trajectory writer and hipGraph capture as the Langevin class (``v`` stays zero: the reference
    saves no velocities here either, and the kinetic energy it reports is 0)."""

    def __init__(self, force_fn, pos, beta_mol, n_atoms: int, dt: float = 0.004, friction: float = 1.0,
                 seed: int = 103838, use_hip: Optional[bool] = None):
        assert friction > 0   # reference langevin.py:342
        super().__init__(force_fn, pos, torch.ones(pos.shape[0], device=pos.device), beta_mol, n_atoms, dt=dt,
                         friction=friction, seed=seed, use_hip=use_hip)
        self.v.zero_()

    def _captured_step(self):
        self.noise.normal_(generator=self.rng)
        overdamped_step(self.x, self.f, self.beta_node, self.noise, self.dt, self.friction)
        e, f = self.force_fn(self.x)
        self.energy.copy_(e)
        self.f.copy_(f)

    def timestep(self):
        if self._graph is not None:
            self._graph.replay()
            return
        self.draw_noise()
        if self.use_hip:
            overdamped_step(self.x, self.f, self.beta_node, self.noise, self.dt, self.friction)
        else:  # PyTorch path (reference langevin.py:370-372,417-421)
            dtau = (1.0 / self.beta_node[:, None] / self.friction) * self.dt
            self.x = self.x + self.f * dtau + torch.sqrt(2 * dtau) * self.noise
        self.energy, self.f = self.force_fn(self.x)


def pt_pairs(n_betas: int, even: bool):
    """Neighbouring temperature pairs proposed in one exchange attempt: (0,1),(2,3).. on even
    attempts, (1,2),(3,4).. on odd ones; with two temperatures there is no odd pair and the reference
    proposes (0,1) every time (reference simulation/parallel_tempering.py:256-284,349-366)."""
    odd = [(a, a + 1) for a in range(1, n_betas - 1, 2)]
    evn = [(a, a + 1) for a in range(0, n_betas - 1, 2)]
    return evn if (even or not odd) else odd


def pt_count(matrix: torch.Tensor, a: int, b: int, approved: int, n_indep: int):
    """The reference's symmetric acceptance matrix (parallel_tempering.py:99-104,407-412): entry [a][b] (a < b)
    accumulates the accepted exchanges of the temperature pair, entry [b][a] the rejected ones."""
    matrix[a, b] += approved
    matrix[b, a] += n_indep - approved


def pt_probability(u_a, u_b, beta_a, beta_b):
    """exp((U_a - U_b) (beta_a - beta_b)), every operand in the energies' dtype as the reference forms it from its
    fp32 ``self.beta`` tensor (parallel_tempering.py:385-393); ``beta_*``: floats or tensors."""
    ba = torch.as_tensor(beta_a, dtype=u_a.dtype, device=u_a.device)
    bb = torch.as_tensor(beta_b, dtype=u_a.dtype, device=u_a.device)
    return torch.exp((u_a - u_b) * (ba - bb))


def pt_accept(u_a, u_b, beta_a, beta_b, rand_u):
    """Metropolis criterion of reference parallel_tempering.py:393-394: accept where rand < p."""
    return rand_u < pt_probability(u_a, u_b, beta_a, beta_b)


def pt_vscale(beta_new: float, beta_old: float, dtype=torch.float32):
    """sqrt(beta_new_slot / beta_of_the_slot_the_velocities_come_from) formed in fp32 like the reference's
    ``torch.sqrt(betas_a / betas_b)`` (parallel_tempering.py:467-470)"""
    return torch.sqrt(torch.tensor(beta_new, dtype=dtype) / torch.tensor(beta_old, dtype=dtype))


FORCE_MODES = ("reference", "moved")


def _check_force_mode(forces: str) -> str:
    if forces not in FORCE_MODES:
        raise ValueError(f"forces must be one of {FORCE_MODES}, got {forces!r}")
    return forces


class _UniformSource:
    """Where an exchange attempt's uniform numbers come from.  The reference draws ``torch.rand(n)`` -- n = all pairs of
    the attempt, pair-major -- from the GLOBAL CPU generator (parallel_tempering.py:394): ``seed=None`` does exactly
    that; an integer seed uses a private CPU generator; ``rand_source(attempt, n) -> Tensor[n]`` injects the numbers
    (the parity tests replay the reference's own draws through it)."""

    def __init__(self, seed, rand_source):
        self.rand_source = rand_source
        self.gen = None if (seed is None or rand_source is not None) else torch.Generator().manual_seed(seed)

    def draw(self, attempt: int, n: int):
        if self.rand_source is not None:
            u = torch.as_tensor(self.rand_source(attempt, n), dtype=torch.float32).cpu()
            assert u.shape == (n,), (u.shape, n)
            return u
        return torch.rand(n) if self.gen is None else torch.rand(n, generator=self.gen)


class DistributedReplicaExchange:
    """Parallel tempering with ONE temperature per rank (BASELINE config 5): rank r simulates
    ``n_indep`` replicas at betas[r]; every ``interval`` steps neighbouring ranks attempt to swap
    replica k of one with replica k of the other (reference simulation/parallel_tempering.py:368-481,
    where all temperatures live in one batch: rank r = the reference's slot block r of its beta-major layout).

    The force path never communicates.  An exchange attempt is point-to-point between the two
    partner ranks only (torch.distributed send/recv = RCCL over the direct xGMI link on GPUs, gloo
    on CPU).  The Metropolis decision has ONE source, the lower rank of the pair: the upper rank sends
    its energies (n_indep floats), the lower rank draws the uniform numbers, decides, and sends the accept mask back
    (n_indep bytes) -- two ranks can never disagree on a borderline ``rand < p`` and then wait for a transfer the other
    never posts.  Uniform numbers: a CPU generator keyed by (seed, attempt, lower rank), or -- ``rand_source(attempt, n)``
    -- the attempt's whole pair-major vector as the reference draws it (parallel_tempering.py:394), of which the lower
    rank of the j-th proposed pair takes entries [j n_indep, (j+1) n_indep).

    ``forces`` selects what happens to forces and energies of an accepted pair:
      "reference" (default)  as parallel_tempering.py:415-481: positions and rescaled velocities cross, ``sim.f`` and
                  ``sim.energy`` stay where they are -- the next timestep's first half-kick uses the forces of the
                  configuration that LEFT the slot (simulation/base.py:612-615 -> langevin.py:137), exactly the
                  reference's trajectory (tests/test_parallel_tempering_reference.py replays its PTSimulation run);
      "moved"     forces and energies travel with the configuration they were computed for (F(x) depends on the
                  replica's own coordinates only, one model for all temperatures), so the half-kick after an exchange
                  uses the forces AT the new positions, still without a force evaluation; 0.4 MB more per way.
    Payload (fixed size, <= 0.83 / 1.24 MB at 128 x 269 beads) crosses the link once each way, only when at least one
    pair was accepted, and is applied where the mask says so."""

    def __init__(self, betas, n_indep: int, n_atoms: int, seed: int = 0, group=None, forces: str = "reference",
                 rand_source=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        assert len(betas) == self.world, "one beta per rank"
        self.betas = [float(b) for b in betas]
        self.n_indep, self.n_atoms, self.seed = n_indep, n_atoms, seed
        self.forces = _check_force_mode(forces)
        self.rand_source = rand_source
        self.attempt = 0
        # this rank's share of the acceptance matrix: the lower rank of a pair counts for the pair
        self.local_matrix = torch.zeros(self.world, self.world, dtype=torch.int64)
        self.exchange_seconds = 0.0   # wall time spent inside exchange attempts that had a partner
        self.exchanges = 0
        self.bytes_sent = 0           # payload bytes this rank put on the link (energies / mask / x+v[+f])
        self.payload_transfers = 0    # attempts in which positions+velocities crossed
        self.last_accept = None       # accept mask of this rank's pair in the last attempt (None: sat out)
        # gloo moves host memory: device tensors are staged through the CPU (tests, rehearsals); RCCL takes them as they are
        self._stage = dist.get_backend(group) != "nccl"

    def partner(self, even: bool) -> Optional[int]:
        for a, b in pt_pairs(self.world, even):
            if self.rank == a:
                return b
            if self.rank == b:
                return a
        return None

    def _send(self, t, peer):
        self.bytes_sent += t.numel() * t.element_size()
        self.dist.send(t.cpu() if self._stage else t, peer, group=self.group)

    def _recv(self, like, peer):
        buf = torch.empty(like.shape, dtype=like.dtype, device="cpu" if self._stage else like.device)
        self.dist.recv(buf, peer, group=self.group)
        return buf.to(like.device)

    def _swap(self, t, peer):
        """send ``t`` to the partner and receive the partner's tensor of the same shape (one round, both ways)"""
        dist = self.dist
        out = t.cpu() if self._stage else t
        inc = torch.empty_like(out)
        self.bytes_sent += t.numel() * t.element_size()
        ops = [dist.P2POp(dist.isend, out, peer, self.group), dist.P2POp(dist.irecv, inc, peer, self.group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return inc.to(t.device)

    def _uniforms(self, attempt: int, even: bool, lo: int):
        if self.rand_source is None:
            g = torch.Generator().manual_seed((self.seed * 1000003 + attempt) * 4099 + lo)
            return torch.rand(self.n_indep, generator=g)
        pairs = pt_pairs(self.world, even)
        j = [a for a, _ in pairs].index(lo)
        u = torch.as_tensor(self.rand_source(attempt, len(pairs) * self.n_indep), dtype=torch.float32).cpu()
        return u[j * self.n_indep:(j + 1) * self.n_indep]

    def __call__(self, sim: "LangevinSimulation"):
        import time

        even = self.attempt % 2 == 0
        peer = self.partner(even)
        attempt = self.attempt
        self.attempt += 1
        self.last_accept = None
        if peer is None:      # an end rank sits this phase out (odd phase: ranks 0 and W-1 for even W)
            return
        t0 = time.perf_counter()
        lo, hi = min(self.rank, peer), max(self.rank, peer)
        u_mine = sim.energy.detach().contiguous()
        if self.rank == hi:
            # 1. my energies to the deciding rank, 2. its decision back
            self._send(u_mine, lo)
            acc = self._recv(torch.empty(self.n_indep, dtype=torch.uint8, device=u_mine.device), lo).bool()
        else:
            u_peer = self._recv(u_mine, hi)
            rand_u = self._uniforms(attempt, even, lo).to(u_mine.device)
            acc = pt_accept(u_mine, u_peer, self.betas[lo], self.betas[hi], rand_u)
            self._send(acc.to(torch.uint8), hi)
        self.last_accept = acc
        n_acc = int(acc.sum())   # one host sync per attempt; the same mask on both partners by construction
        if self.rank == lo:
            pt_count(self.local_matrix, lo, hi, n_acc, self.n_indep)
        if n_acc:
            # 3. positions and velocities (and, "moved", forces and energies) of the whole batch cross the link once,
            #    each way
            moved = self.forces == "moved"
            parts = [sim.x, sim.v, sim.f] if moved else [sim.x, sim.v]
            incoming = self._swap(torch.stack(parts).contiguous(), peer)
            self.payload_transfers += 1
            # 4. apply where accepted; velocities rescaled as reference parallel_tempering.py:467-477
            #    (slot a receives v_b * sqrt(beta_a / beta_b), slot b receives v_a * sqrt(beta_b / beta_a))
            mask = acc.repeat_interleave(self.n_atoms)[:, None]
            scale = pt_vscale(self.betas[self.rank], self.betas[peer], sim.v.dtype).to(sim.v.device)
            # (in place: a captured hipGraph keeps reading and writing these buffers)
            sim.x.copy_(torch.where(mask, incoming[0], sim.x))
            sim.v.copy_(torch.where(mask, incoming[1] * scale, sim.v))
            if moved:
                u_peer_all = self._swap(u_mine, peer)
                sim.f.copy_(torch.where(mask, incoming[2], sim.f))
                sim.energy.copy_(torch.where(acc, u_peer_all, sim.energy))
        self.exchange_seconds += time.perf_counter() - t0
        self.exchanges += 1

    def acceptance_matrix(self):
        """The reference's [n_betas, n_betas] matrix (accepted above the diagonal, rejected below) summed over
        ranks; a collective, for export time only (reference parallel_tempering.py:504-517 writes it to
        ``{filename}_acceptance_XXXX.npy`` and resets it)."""
        dev = "cuda" if self.dist.get_backend(self.group) == "nccl" else "cpu"
        m = self.local_matrix.to(dev).clone()   # (a copy: all_reduce works in place, this rank's counts must survive it)
        self.dist.all_reduce(m, group=self.group)
        return m.cpu()

    def reset_acceptance(self):
        self.local_matrix.zero_()

    def acceptance(self):
        """(accepted[k], attempted[k]) of the pair (k, k+1), every rank."""
        m = self.acceptance_matrix()
        acc = torch.tensor([int(m[k, k + 1]) for k in range(self.world - 1)] + [0])
        rej = torch.tensor([int(m[k + 1, k]) for k in range(self.world - 1)] + [0])
        return acc, acc + rej


class LocalReplicaExchange:
    """All temperatures in one batch on one GPU (the reference's layout: beta-major blocks of
    ``n_indep`` replicas, parallel_tempering.py:204-220); accepted pairs are swapped in place by the
    HIP kernel behind fmd_pt_swap_local (with every MLCG_USE_* gate off -- ``sim.use_hip`` False -- by the same torch
    indexing the reference uses, which is what runs on the CPU).

    ``seed=None`` draws the uniform numbers from the global CPU generator as the reference does
    (parallel_tempering.py:394), an integer from a private one, ``rand_source(attempt, n)`` injects them.
    ``forces``: "reference" (default) leaves ``sim.f`` / ``sim.energy`` in place as parallel_tempering.py:415-481 does;
    "moved" swaps them with the configurations (see DistributedReplicaExchange).  ``last`` keeps the last attempt's
    proposal and decision (pairs, probabilities, uniforms, mask) for bookkeeping and tests."""

    def __init__(self, betas, n_indep: int, n_atoms: int, seed: Optional[int] = 0, forces: str = "reference",
                 rand_source=None):
        self.betas, self.n_indep, self.n_atoms = [float(b) for b in betas], n_indep, n_atoms
        self.forces = _check_force_mode(forces)
        self.uniforms = _UniformSource(seed, rand_source)
        self.attempt = 0
        self.acceptance_matrix = torch.zeros(len(self.betas), len(self.betas), dtype=torch.int64)
        self.last = None
        self._scratch = None

    def reset_acceptance(self):
        """what the reference does after writing ``{filename}_acceptance_XXXX.npy`` (parallel_tempering.py:514-517)"""
        self.acceptance_matrix.zero_()

    def __call__(self, sim: "LangevinSimulation"):
        even = self.attempt % 2 == 0
        attempt = self.attempt
        self.attempt += 1
        pairs = pt_pairs(len(self.betas), even)
        if not pairs:
            return
        dev = sim.x.device
        k = torch.arange(self.n_indep)
        mol_a = torch.cat([a * self.n_indep + k for a, _ in pairs]).to(dev)
        mol_b = torch.cat([b * self.n_indep + k for _, b in pairs]).to(dev)
        beta = sim.beta_mol
        p = pt_probability(sim.energy[mol_a], sim.energy[mol_b], beta[mol_a], beta[mol_b])
        u = self.uniforms.draw(attempt, p.numel())
        acc = (u.to(dev) < p).to(torch.uint8)
        self.last = {"pair_a": mol_a, "pair_b": mol_b, "p": p, "uniform": u, "accept": acc}
        per_pair = acc.view(len(pairs), self.n_indep).sum(1).tolist()   # one host sync per attempt
        for (a, b), n_acc in zip(pairs, per_pair):
            pt_count(self.acceptance_matrix, a, b, int(n_acc), self.n_indep)
        if not sum(per_pair):
            return
        moved = self.forces == "moved"
        if not getattr(sim, "use_hip", True):     # gates off: plain torch, parallel_tempering.py:449-480
            m = acc.bool()
            rows_a = (mol_a[m, None] * self.n_atoms + torch.arange(self.n_atoms, device=dev)).reshape(-1)
            rows_b = (mol_b[m, None] * self.n_atoms + torch.arange(self.n_atoms, device=dev)).reshape(-1)
            ba = beta[mol_a[m]].repeat_interleave(self.n_atoms)[:, None]
            bb = beta[mol_b[m]].repeat_interleave(self.n_atoms)[:, None]
            x, v = sim.x.clone(), sim.v.clone()
            x[rows_a], x[rows_b] = sim.x[rows_b], sim.x[rows_a]
            v[rows_a] = sim.v[rows_b] * torch.sqrt(ba / bb)
            v[rows_b] = sim.v[rows_a] * torch.sqrt(bb / ba)
            sim.x, sim.v = x, v
            if moved:
                f = sim.f.clone()
                f[rows_a], f[rows_b] = sim.f[rows_b], sim.f[rows_a]
                sim.f = f
        else:
            pt_swap_local(sim.x, sim.v, mol_a, mol_b, acc, beta, self.n_atoms)
            if moved:
                # the same in-batch swap on the force rows -- its velocity argument is a scratch array
                if self._scratch is None or self._scratch.shape != sim.f.shape:
                    self._scratch = torch.zeros_like(sim.f)
                pt_swap_local(sim.f, self._scratch, mol_a, mol_b, acc, beta, self.n_atoms)
        if moved:
            ea, eb, m = sim.energy[mol_a], sim.energy[mol_b], acc.bool()
            sim.energy[mol_a] = torch.where(m, eb, ea)
            sim.energy[mol_b] = torch.where(m, ea, eb)
