"""Rows N1 / N3 / N4 pinned to the reference's OWN simulation layer.

The fixtures tests/golden/langevin_run.npz and sumout_state_dict.npz were written in the build container by
oracle/make_golden_simulation.py from the reference's ``LangevinSimulation`` (simulation/langevin.py:102-177 BAOAB step,
:234-291 and simulation/base.py:1294-1453 save / write) driving its ``SumOut{SchNet, bonds, angles, repulsion,
dihedrals}`` container (models/gradients.py:19-152): the noise it drew, the states it reached, the files it wrote, the
container's ``state_dict()`` and outputs.  CPU tests run the host logic (PyTorch path, loaders); ``-m gpu`` tests run the
HIP integrator kernels, the fused engine and the condensed priors against the same vectors.
"""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def run(golden_dir):
    return np.load(os.path.join(golden_dir, "langevin_run.npz"))


@pytest.fixture(scope="module")
def sumout(golden_dir):
    z = np.load(os.path.join(golden_dir, "sumout_state_dict.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd:")}
    topo = {k[len("mapping_"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mapping_")}
    return z, sd, topo


N_SIMS, N_ATOMS = 3, 41


def _replay(run, device, use_hip):
    """Our LangevinSimulation fed with the reference's start state, the reference's noise and the reference's forces
    (the frames it saved): what remains under test is the integrator arithmetic."""
    from flashmd_amd import simulate

    frames_f = np.concatenate([run["file:run_forces_0000.npy"], run["file:run_forces_0001.npy"]], axis=1)  # [sims, 4, atoms, 3]
    frames_u = np.concatenate([run["file:run_potential_0000.npy"], run["file:run_potential_0001.npy"]], axis=1)
    calls = {"n": 0}

    def force_fn(x):
        k = calls["n"]
        calls["n"] += 1
        if k == 0:
            return torch.zeros(N_SIMS, device=device), torch.from_numpy(run["forces0"]).to(device)
        f = torch.from_numpy(np.ascontiguousarray(frames_f[:, k - 1])).reshape(-1, 3).to(device)
        return torch.from_numpy(np.ascontiguousarray(frames_u[:, k - 1])).to(device), f.contiguous()

    class Replay(simulate.LangevinSimulation):
        step = 0

        def draw_noise(self):
            self.noise.copy_(torch.from_numpy(run["noise"][self.step]))
            self.step += 1

    sim = Replay(force_fn, torch.from_numpy(run["pos0"]).to(device), torch.from_numpy(run["masses"]).to(device),
                 torch.from_numpy(run["betas"]).to(device), N_ATOMS, dt=float(run["dt"]), friction=float(run["friction"]),
                 use_hip=use_hip)
    sim.v = torch.from_numpy(run["vel0"]).to(device).contiguous()   # the reference's own Maxwell-Boltzmann draw
    return sim


def _check_steps(run, sim, tol):
    for t in range(4):
        sim.timestep()
        ex = np.abs(sim.x.cpu().numpy() - run["pos"][t]).max()
        ev = np.abs(sim.v.cpu().numpy() - run["vel"][t]).max()
        assert ex < tol * np.abs(run["pos"][t]).max() and ev < tol * np.abs(run["vel"][t]).max(), (t, ex, ev)


def test_baoab_torch_path_reproduces_the_reference_run(run):
    """Row N1, host logic: the PyTorch branch of LangevinSimulation.timestep follows the reference's own trajectory
    (same noise, same forces) to fp32 rounding."""
    _check_steps(run, _replay(run, "cpu", use_hip=False), 1e-6)


@pytest.mark.gpu
def test_baoab_hip_kernels_reproduce_the_reference_run(run):
    """Row N1 on the device: fmd_langevin_pre / _post against the states the reference's LangevinSimulation reached."""
    _check_steps(run, _replay(run, "cuda:0", use_hip=True), 2e-6)


def test_trajectory_files_match_the_reference_run(run, tmp_path):
    """Row N3: driven through the same four steps (save_interval 1, export_interval 2, forces + energies + checkpoints),
    TrajectoryWriter writes the files the reference wrote: same names, shapes, dtypes, contents (coordinates, forces and
    potential to fp32 rounding of the replayed integrator; kinetic energy from the velocities), and checkpoints with
    the reference's keys and values -- including the one of the start state."""
    from flashmd_amd import simulate

    sim = _replay(run, "cpu", use_hip=False)
    base = str(tmp_path / "run")
    wr = simulate.TrajectoryWriter(base, sim, save_interval=1, export_interval=2, save_forces=True, save_energies=True,
                                   create_checkpoints=True, log_interval=None)
    sim.simulate(4, writer=wr)
    want = sorted(str(n) for n in run["file_names"])
    got = sorted(os.path.basename(f) for f in wr.files)
    assert got == want
    for name in want:
        if name.endswith(".npy"):
            ref, mine = run["file:" + name], np.load(str(tmp_path / name))
            assert mine.shape == ref.shape and mine.dtype == ref.dtype, name
            assert np.abs(mine - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), name
        else:
            ck = torch.load(str(tmp_path / name), weights_only=True)
            keys = {k.split(":")[2] for k in run.files if k.startswith(f"ckpt:{name}:")}
            assert set(ck) == keys, name
            for k in keys:
                ref = run[f"ckpt:{name}:{k}"]
                if k in ("pos", "velocities"):
                    assert tuple(ck[k].shape) == ref.shape and ck[k].dtype == torch.float32
                    assert np.abs(ck[k].numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (name, k)
                else:   # (a None log_interval is stored as -1 in the fixture)
                    assert (-1 if ck[k] is None else int(ck[k])) == int(ref), (name, k)


def test_sumout_state_dict_is_understood(sumout):
    """Row N4, host logic: the state_dict of the reference's SumOut container -- SchNet found under its prefix and
    rebuilt strictly; every prior found under its name, kind told by its buffers, tables equal to the reference's
    buffers; the condensed priors' own torch path gives the reference's prior energies in fp64."""
    import flashmd_amd.schnet as S
    from flashmd_amd import prior

    z, sd, topo = sumout
    assert S.find_schnet_prefix(sd) == "models.SchNet.model."
    model = S.model_from_state_dict(sd)
    assert model.rbf_layer.num_rbf == 20 and len(model.interaction_blocks) == 2 and model.rbf_layer.cutoff.cutoff_upper == 5.0
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd["models.SchNet.model." + k].to(v.dtype)), k
    found = prior.prior_tables_from_state_dict(sd)
    assert {n: k for n, (k, _) in found.items()} == {n: n for n in ("bonds", "angles", "repulsion", "dihedrals")}
    types, ptr = torch.from_numpy(z["types"]), torch.from_numpy(z["ptr"])
    sp = prior.StaticPriors.from_state_dict(sd, topo, types, ptr)
    e = sp.energy_torch(torch.from_numpy(z["pos"]).double(), ptr)
    ref = sum(z[f"energy_{n}_ref64"] for n in ("bonds", "angles", "repulsion", "dihedrals"))
    assert np.abs(e.numpy() - ref).max() < 1e-9 * np.abs(ref).max()
    # a trained basis does not pass for the untrained one; an explicit cutoff loads it
    bad = dict(sd)
    bad["models.SchNet.model.rbf_layer.offset"] = sd["models.SchNet.model.rbf_layer.offset"] * 1.01
    with pytest.raises(ValueError, match="cutoff"):
        S.model_from_state_dict(bad)
    assert S.model_from_state_dict(bad, cutoff=5.0).rbf_layer.cutoff.cutoff_upper == 5.0
    with pytest.raises(KeyError):
        prior.StaticPriors.from_state_dict(sd, {k: v for k, v in topo.items() if k != "angles"}, types, ptr)


@pytest.mark.gpu
def test_force_provider_from_sumout_state_dict_matches_the_reference_container(sumout):
    """Row N4 end to end: fused engine + condensed priors built from the reference container's state_dict reproduce
    that container's own energies and forces (fp64 vectors), in total and part by part."""
    from flashmd_amd import simulate

    z, sd, topo = sumout
    dev = "cuda:0"
    types, ptr = torch.from_numpy(z["types"]), torch.from_numpy(z["ptr"])
    fn = simulate.force_fn_from_state_dict(sd, types, ptr, topology=topo, device=dev)
    pos = torch.from_numpy(z["pos"]).to(dev)
    e, f = fn(pos)
    fn.check_status()
    re = np.abs(e.double().cpu().numpy() - z["energy_ref64"]).max() / np.abs(z["energy_ref64"]).max()
    rf = np.linalg.norm(f.double().cpu().numpy() - z["forces_ref64"]) / np.linalg.norm(z["forces_ref64"])
    e_s, f_s = fn.engine.energy_forces(pos, types.to(dev), ptr.to(dev), N_ATOMS)
    rs = np.linalg.norm(f_s.double().cpu().numpy() - z["forces_SchNet_ref64"]) / np.linalg.norm(z["forces_SchNet_ref64"])
    e_p, f_p = fn.priors.energy_forces(pos, ptr.to(dev))
    ref_p = sum(z[f"forces_{n}_ref64"] for n in ("bonds", "angles", "repulsion", "dihedrals"))
    rp = np.linalg.norm(f_p.double().cpu().numpy() - ref_p) / np.linalg.norm(ref_p)
    print(f"SumOut state_dict: total energy {re:.2e} forces {rf:.2e}; SchNet forces {rs:.2e}; prior forces {rp:.2e}")
    assert re < 1e-4 and rf < 1e-4 and rs < 1e-4 and rp < 1e-4
    assert rf < 2e-5 and rs < 2e-5 and rp < 2e-5   # where fp32 actually lands


@pytest.mark.gpu
def test_captured_timestep_with_the_full_force_provider(sumout):
    """The hipGraph of a timestep over the WHOLE force provider of row N4 (fused engine + the four condensed prior
    kernels, built from the reference container's state_dict): ten replayed steps equal ten un-captured ones bit for bit,
    and the provider's status words are still readable afterwards."""
    from flashmd_amd import simulate

    z, sd, topo = sumout
    dev = "cuda:0"
    types, ptr = torch.from_numpy(z["types"]), torch.from_numpy(z["ptr"])
    pos = torch.from_numpy(z["pos"]).to(dev)
    masses = torch.full((pos.shape[0],), 12.0, device=dev)
    beta = torch.full((N_SIMS,), 1.67, device=dev)

    def make():
        fn = simulate.force_fn_from_state_dict(sd, types, ptr, topology=topo, device=dev)
        return simulate.LangevinSimulation(fn, pos, masses, beta, N_ATOMS, dt=0.002, friction=1.0, seed=5)

    eager, graph = make(), make()
    graph.capture()
    for _ in range(10):
        eager.timestep()
        graph.timestep()
    torch.cuda.synchronize()
    assert torch.isfinite(eager.x).all()
    assert torch.equal(eager.x, graph.x) and torch.equal(eager.v, graph.v)
    assert torch.equal(eager.f, graph.f) and torch.equal(eager.energy, graph.energy)
    graph.check_status()


# ---------------------------------------------------------------------------------------------- overdamped (round 5)
@pytest.fixture(scope="module")
def od_run(golden_dir):
    return np.load(os.path.join(golden_dir, "overdamped_run.npz"))


def _replay_overdamped(od_run, device, use_hip, tol):
    """OverdampedSimulation fed with the reference's start positions, noise and per-step forces: the reference's own
    OverdampedSimulation (simulation/langevin.py:315-434, run by oracle/make_golden_overdamped.py) state by state."""
    from flashmd_amd import simulate

    calls = {"n": 0}

    def force_fn(x):
        k = calls["n"]
        calls["n"] += 1
        f = od_run["forces_in"][0] if k == 0 else od_run["forces"][k - 1]
        u = od_run["potential"][max(k - 1, 0)]
        return torch.from_numpy(u.copy()).to(device), torch.from_numpy(f.copy()).to(device)

    class Replay(simulate.OverdampedSimulation):
        step = 0

        def draw_noise(self):
            self.noise.copy_(torch.from_numpy(od_run["noise"][self.step]))
            self.step += 1

    sim = Replay(force_fn, torch.from_numpy(od_run["pos0"]).to(device), torch.from_numpy(od_run["betas"]).to(device),
                 int(od_run["n_atoms"]), dt=float(od_run["dt"]), friction=float(od_run["friction"]), use_hip=use_hip)
    for t in range(od_run["pos"].shape[0]):
        sim.timestep()
        err = np.abs(sim.x.cpu().numpy() - od_run["pos"][t]).max() / np.abs(od_run["pos"][t]).max()
        assert err < tol, (t, err)
    assert float(sim.v.abs().max()) == 0.0


def test_overdamped_torch_path_reproduces_the_reference_run(od_run):
    _replay_overdamped(od_run, "cpu", False, 1e-6)


@pytest.mark.gpu
def test_overdamped_hip_kernel_reproduces_the_reference_run(od_run):
    """fmd_overdamped_step against the positions the reference's OverdampedSimulation reached."""
    _replay_overdamped(od_run, "cuda:0", True, 2e-6)
