"""Row (e2) pinned to the reference's OWN ``PTSimulation`` (simulation/parallel_tempering.py:25-536).

tests/golden/pt_run.npz was written in the build container by oracle/make_golden_pt.py: the reference's PTSimulation
over its SumOut{SchNet, bonds, angles, repulsion, dihedrals} container, 4 betas x 2 replicas x 41 beads, 36 timesteps,
an exchange attempt every 3 steps (six of each phase; 31 of 36 proposed pairs accepted, 5 rejected) -- per step the noise
it drew, the forces it handed to the step, the state it reached; per attempt the proposed pairs, probabilities, the
uniform numbers it drew from the global CPU generator, the decisions, positions / velocities before and after, and the
acceptance matrix.

Replayed here through ``LocalReplicaExchange`` (all temperatures in one batch, the reference's layout) and
``DistributedReplicaExchange`` (one temperature per rank over gloo) with the recorded uniforms injected: proposed pairs,
decisions, swapped positions, sqrt(beta_a / beta_b)-scaled velocities and the matrix must match; ``forces="reference"``
(the reference hands the PRE-exchange forces to the next step, simulation/base.py:612-615 -> langevin.py:137) must
reproduce the next steps' states, ``forces="moved"`` must not.  ``-m gpu``: the HIP swap / integrator kernels on the same
replay, and the whole run LIVE on the fused engine + condensed priors built from the fixture's state_dict.
"""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def run(golden_dir):
    return np.load(os.path.join(golden_dir, "pt_run.npz"))


def _replay_sim(run, device, use_hip, sl=None, betas=None):
    """LangevinSimulation fed with the reference's start state, noise and per-step forces / potentials (optionally the
    slot block ``sl`` = slice of molecules, one rank's share); the integrator + exchange arithmetic is under test."""
    from flashmd_amd import simulate

    A = int(run["n_atoms"])
    n_sims = run["potential0"].shape[0]
    sl = sl or slice(0, n_sims)
    rows = slice(sl.start * A, sl.stop * A)
    calls = {"n": 0}

    def force_fn(x):
        k = calls["n"]
        calls["n"] += 1
        u = run["potential0"] if k == 0 else run["potential"][k - 1]
        f = run["forces_in"][0] if k == 0 else run["forces"][k - 1]   # (forces0: a separate evaluation, 1e-7 away)
        return (torch.from_numpy(np.ascontiguousarray(u[sl])).to(device),
                torch.from_numpy(np.ascontiguousarray(f[rows])).to(device))

    class Replay(simulate.LangevinSimulation):
        step = 0

        def draw_noise(self):
            self.noise.copy_(torch.from_numpy(np.ascontiguousarray(run["noise"][self.step][rows])))
            self.step += 1

    beta_mol = torch.from_numpy(run["betas"]).repeat_interleave(int(run["n_indep"]))[sl] if betas is None else betas
    sim = Replay(force_fn, torch.from_numpy(run["pos0"][rows]).to(device), torch.from_numpy(run["masses"][rows]).to(device),
                 beta_mol.to(device), A, dt=float(run["dt"]), friction=float(run["friction"]), use_hip=use_hip)
    sim.v = torch.from_numpy(np.ascontiguousarray(run["vel0"][rows])).to(device).contiguous()
    return sim


def _uniforms_of(run):
    def rand_source(attempt, n):
        u = run["att_uniform"][attempt]
        assert (u[:n] <= 1.0).all() and (u[n:] == 2.0).all()   # the reference drew exactly n numbers
        return torch.from_numpy(u[:n].copy())
    return rand_source


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def _replay_local(run, device, use_hip, forces, tol):
    """Returns the worst relative deviation of positions / velocities from the reference's states over all steps and
    attempts; decisions, pairs and matrices are asserted exactly."""
    from flashmd_amd import simulate

    n_indep, A, interval = int(run["n_indep"]), int(run["n_atoms"]), int(run["exchange_interval"])
    sim = _replay_sim(run, device, use_hip)
    ex = simulate.LocalReplicaExchange(run["betas"].tolist(), n_indep, A, seed=None, forces=forces,
                                       rand_source=_uniforms_of(run))
    n_steps, worst, k, exported = run["pos"].shape[0], 0.0, 0, None
    for t in range(n_steps):
        f_in = sim.f.clone()
        sim.timestep()
        if forces == "reference":
            # what the half-kick used = what the reference handed to its step (stale after an accepted exchange)
            assert np.array_equal(f_in.cpu().numpy(), run["forces_in"][t])
        worst = max(worst, _rel(sim.x.cpu().numpy(), run["pos"][t]), _rel(sim.v.cpu().numpy(), run["vel"][t]))
        if forces == "reference":
            assert worst < tol, (t, worst)
        if t + 1 == n_steps:
            # the reference exports (and resets) its matrix at the end of the chunk, BEFORE that step's attempt
            # (simulation/base.py:596-615, parallel_tempering.py:504-517)
            exported = ex.acceptance_matrix.clone()
            ex.reset_acceptance()
        if (t + 1) % interval == 0:
            assert int(run["att_after_step"][k]) == t + 1
            ex(sim)
            n = int((run["att_pair_a"][k] >= 0).sum())
            assert ex.last["pair_a"].cpu().tolist() == run["att_pair_a"][k][:n].tolist()
            assert ex.last["pair_b"].cpu().tolist() == run["att_pair_b"][k][:n].tolist()
            if forces == "reference":
                assert np.abs(ex.last["p"].cpu().numpy() - run["att_p_pair"][k][:n]).max() < 1e-5
                assert ex.last["accept"].cpu().bool().tolist() == run["att_approved"][k][:n].tolist()
                # decisions are not borderline in this fixture: fp32 noise on the energies cannot flip one
                assert np.abs(run["att_uniform"][k][:n] - run["att_p_pair"][k][:n]).min() > 1e-3
                assert np.array_equal(ex.acceptance_matrix.numpy(), run["att_matrix"][k].astype(np.int64))
                worst = max(worst, _rel(sim.x.cpu().numpy(), run["att_x_after"][k]),
                            _rel(sim.v.cpu().numpy(), run["att_v_after"][k]))
                assert worst < tol, ("attempt", k, worst)
            k += 1
    assert k == run["att_approved"].shape[0] == 12
    return worst, ex, exported


def test_local_exchange_torch_path_reproduces_the_reference_pt_run(run):
    """Host logic (gates off, CPU): 36 BAOAB steps and 12 exchange attempts of the reference's PTSimulation, state by
    state, to fp32 rounding; matrix per attempt and the file the reference exported."""
    worst, ex, exported = _replay_local(run, "cpu", False, "reference", 2e-6)
    print(f"PT replay (torch path): worst relative deviation {worst:.2e}")
    assert np.array_equal(exported.numpy(), run["acceptance_file"].astype(np.int64))     # the file the reference wrote
    assert int(exported.sum()) == 34 and int(ex.acceptance_matrix.sum()) == 2             # 11 attempts, then the last one
    assert int(torch.triu(exported + ex.acceptance_matrix, 1).sum()) == int(run["att_approved"].sum())


def test_acceptance_file_matches_the_reference_export(run, tmp_path):
    """The reference's PTSimulation writes ``{filename}_acceptance_XXXX.npy`` at every export and resets the matrix
    (parallel_tempering.py:504-517), BEFORE that step's exchange attempt (simulation/base.py:596-615).  The same run
    driven through LangevinSimulation.simulate() with a TrajectoryWriter that knows the exchange: same file name, dtype,
    shape and counts as the file the reference wrote."""
    from flashmd_amd import simulate

    n_indep, A, interval = int(run["n_indep"]), int(run["n_atoms"]), int(run["exchange_interval"])
    sim = _replay_sim(run, "cpu", False)
    ex = simulate.LocalReplicaExchange(run["betas"].tolist(), n_indep, A, seed=None, rand_source=_uniforms_of(run))
    n_steps = run["pos"].shape[0]
    wr = simulate.TrajectoryWriter(str(tmp_path / "pt"), sim, save_interval=1, export_interval=n_steps, exchange=ex)
    sim.simulate(n_steps, subroutine=ex, interval=interval, writer=wr)
    name = str(run["acceptance_file_name"])
    assert [os.path.basename(f) for f in wr.files if "_acceptance_" in os.path.basename(f)] == [name]   # (pt_acceptance_0001.npy)
    mine = np.load(str(tmp_path / name))
    assert mine.dtype == run["acceptance_file"].dtype and mine.shape == run["acceptance_file"].shape
    assert np.array_equal(mine, run["acceptance_file"])
    # what was counted after the export (the last attempt) is what the reference's matrix held at the end
    assert np.array_equal(ex.acceptance_matrix.numpy(), run["att_matrix"][-1].astype(np.int64))


def test_moved_forces_mode_is_not_the_reference_trajectory(run):
    """The switch matters: with forces travelling with the configurations the step after the first accepted exchange
    already differs from the reference's state (by the half-kick of the force difference), while every proposal and the
    swap itself are the same."""
    worst, _, _ = _replay_local(run, "cpu", False, "moved", None)
    assert worst > 1e-4


def test_exchange_rejects_an_unknown_force_mode():
    from flashmd_amd import simulate

    with pytest.raises(ValueError, match="forces"):
        simulate.LocalReplicaExchange([1.0, 0.5], 1, 3, forces="fresh")


_WORKER = r"""
import os, sys, numpy as np, torch
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "flash-molecular-dynamics_amd"), os.path.join(%(root)r, "tests")]
import torch.distributed as dist
from flashmd_amd import simulate
import test_parallel_tempering_reference as T

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
run = np.load(os.path.join(%(root)r, "tests", "golden", "pt_run.npz"))
n_indep, A, interval = int(run["n_indep"]), int(run["n_atoms"]), int(run["exchange_interval"])
betas = run["betas"].tolist()
assert world == len(betas)
sl = slice(rank * n_indep, (rank + 1) * n_indep)          # rank r = the reference's slot block r (beta-major layout)
rows = slice(sl.start * A, sl.stop * A)
sim = T._replay_sim(run, "cpu", False, sl=sl)
ex = simulate.DistributedReplicaExchange(betas, n_indep, A, forces="reference", rand_source=T._uniforms_of(run))
worst, k, ok = 0.0, 0, True
for t in range(run["pos"].shape[0]):
    sim.timestep()
    worst = max(worst, T._rel(sim.x.numpy(), run["pos"][t][rows]), T._rel(sim.v.numpy(), run["vel"][t][rows]))
    if t + 1 == run["pos"].shape[0]:
        exported = ex.acceptance_matrix()      # export: a collective, then reset -- as the reference's save_exchanges
        ok = ok and np.array_equal(exported.numpy(), run["acceptance_file"].astype(np.int64))
        ok = ok and np.array_equal(ex.acceptance_matrix().numpy(), exported.numpy())   # reading it twice changes nothing
        ex.reset_acceptance()
    if (t + 1) %% interval == 0:
        ex(sim)
        # this rank's pair in the reference's pair-major vectors
        pa, pb, ap = run["att_pair_a"][k], run["att_pair_b"][k], run["att_approved"][k]
        mine = [i for i in range(len(pa)) if pa[i] >= 0 and (pa[i] // n_indep == rank or pb[i] // n_indep == rank)]
        if mine:
            ok = ok and ex.last_accept is not None and ex.last_accept.tolist() == ap[mine].tolist()
        else:
            ok = ok and ex.last_accept is None
        worst = max(worst, T._rel(sim.x.numpy(), run["att_x_after"][k][rows]), T._rel(sim.v.numpy(), run["att_v_after"][k][rows]))
        k += 1
m = ex.acceptance_matrix()
ok = ok and np.array_equal(m.numpy(), run["att_matrix"][-1].astype(np.int64))
sys.stdout.write(f"PTREF {rank} {int(ok)} {worst:.3e}\n")
sys.stdout.flush()
dist.barrier()
dist.destroy_process_group()
"""


def test_distributed_exchange_over_gloo_reproduces_the_reference_pt_run():
    """One temperature per rank (BASELINE configs[4]'s layout), four ranks over gloo, the same replay: every rank's slot
    block follows the reference's states through all 12 attempts (both phases, idle end ranks in the odd one), decisions
    equal the reference's, the summed acceptance matrix equals the reference's."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as fh:
        fh.write(_WORKER % {"root": ROOT})
        path = fh.name
    env = dict(os.environ, MLCG_USE_TRITON_MESSAGE_PASSING="0", MLCG_USE_FUSED_RBF="0", MLCG_USE_FUSED_TANH_LINEAR="0",
               MLCG_USE_CSR="0", MLCG_USE_SRC_CSR_GRAD_X="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), path]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    finally:
        os.unlink(path)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    import re

    recs = re.findall(r"PTREF (\d+) (\d+) (\S+)", out.stdout)
    assert sorted(int(r[0]) for r in recs) == [0, 1, 2, 3], out.stdout[-800:]
    for rank, ok, worst in recs:
        assert int(ok) == 1 and float(worst) < 2e-6, (rank, ok, worst)


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hip_swap_and_integrator_kernels_reproduce_the_reference_pt_run(run):
    """fmd_pt_swap_local + fmd_langevin_pre/_post on the same replay: the reference's states to fp32 rounding."""
    worst, _, _ = _replay_local(run, "cuda:0", True, "reference", 4e-6)
    print(f"PT replay (HIP kernels): worst relative deviation {worst:.2e}")


@pytest.mark.gpu
def test_live_pt_run_on_the_engine_follows_the_reference(run):
    """The whole reference run LIVE: fused engine + condensed priors built from the fixture's state_dict supply forces and
    energies, the reference's noise and uniform numbers are injected.  Every decision of the 12 attempts equals the
    reference's (its probabilities come from the engine's own energies), the acceptance matrix is the reference's, and
    the trajectory stays within fp32 error growth of the reference's over all 36 steps."""
    from flashmd_amd import simulate

    dev = "cuda:0"
    n_indep, A, interval = int(run["n_indep"]), int(run["n_atoms"]), int(run["exchange_interval"])
    n_sims = run["potential0"].shape[0]
    sd = {k[3:]: torch.from_numpy(run[k]) for k in run.files if k.startswith("sd:")}
    topo1 = {k[len("mapping_"):]: torch.from_numpy(run[k]) for k in run.files if k.startswith("mapping_")}
    types = torch.from_numpy(run["types"])[:A].repeat(n_sims)
    ptr = torch.arange(n_sims + 1) * A
    fn = simulate.force_fn_from_state_dict(sd, types, ptr, topology=topo1, device=dev)

    class Live(simulate.LangevinSimulation):
        step = 0

        def draw_noise(self):
            self.noise.copy_(torch.from_numpy(run["noise"][self.step]))
            self.step += 1

    beta_mol = torch.from_numpy(run["betas"]).repeat_interleave(n_indep).to(dev)
    sim = Live(fn, torch.from_numpy(run["pos0"]).to(dev), torch.from_numpy(run["masses"]).to(dev), beta_mol, A,
               dt=float(run["dt"]), friction=float(run["friction"]))
    sim.v = torch.from_numpy(run["vel0"]).to(dev).contiguous()
    assert _rel(sim.f.cpu().numpy(), run["forces0"]) < 1e-4 and _rel(sim.energy.cpu().numpy(), run["potential0"]) < 1e-4
    ex = simulate.LocalReplicaExchange(run["betas"].tolist(), n_indep, A, seed=None, forces="reference",
                                       rand_source=_uniforms_of(run))
    worst_x = worst_f = 0.0
    k = 0
    for t in range(run["pos"].shape[0]):
        sim.timestep()
        worst_x = max(worst_x, _rel(sim.x.cpu().numpy(), run["pos"][t]))
        worst_f = max(worst_f, np.linalg.norm(sim.f.cpu().numpy() - run["forces"][t]) / np.linalg.norm(run["forces"][t]))
        if t + 1 == run["pos"].shape[0]:
            assert np.array_equal(ex.acceptance_matrix.numpy(), run["acceptance_file"].astype(np.int64))
            ex.reset_acceptance()
        if (t + 1) % interval == 0:
            ex(sim)
            n = int((run["att_pair_a"][k] >= 0).sum())
            assert ex.last["accept"].cpu().bool().tolist() == run["att_approved"][k][:n].tolist(), k
            assert np.abs(ex.last["p"].cpu().numpy() - run["att_p_pair"][k][:n]).max() < 2e-3
            assert _rel(sim.x.cpu().numpy(), run["att_x_after"][k]) < 1e-4
            k += 1
    fn.check_status()
    print(f"live PT run: positions {worst_x:.2e}, forces {worst_f:.2e} from the reference over 36 steps")
    assert np.array_equal(ex.acceptance_matrix.numpy(), run["att_matrix"][-1].astype(np.int64))
    assert worst_x < 1e-4 and worst_f < 1e-3
