"""Row N2 (classical prior terms).  CPU: the oracle restatement against the reference's own outputs
(tests/golden/case_priors.npz, written by oracle/make_golden_priors.py from the reference's prior classes).
GPU: the HIP path (fmd_prior_energy_forces) against the oracle and the same vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import prior_oracle as po

KINDS = ("bonds", "angles", "repulsion", "dihedrals")
MORE = ("impropers", "quartic_angles", "restricted_angles")   # case_priors_more.npz (make_golden_priors.py more)
UNEXPORTED = ("raw_angles", "shifted_impropers")              # case_priors_unexported.npz (make_golden_priors.py unexported)


def _load(golden_dir, name="case_priors.npz", kinds=KINDS):
    z = np.load(os.path.join(golden_dir, name))
    pos, types, ptr = (torch.from_numpy(z[k]) for k in ("pos", "types", "ptr"))
    tables = {k: {n[len(k) + 7:]: torch.from_numpy(z[n]) for n in z.files if n.startswith(k + "_table_")} for k in kinds}
    maps = {k: torch.from_numpy(z[k + "_mapping"]) for k in kinds}
    return z, pos, types, ptr, tables, maps


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_prior_oracle_matches_reference_vectors(golden_dir):
    z, pos, types, ptr, tables, maps = _load(golden_dir)
    n_mol = ptr.numel() - 1
    mol_of_bead = torch.repeat_interleave(torch.arange(n_mol), ptr[1:] - ptr[:-1])
    terms = []
    for k in KINDS:
        prm = po.lookup(k, tables[k], types, maps[k])
        e, f = po.prior_energy_forces(pos.double(), [(k, maps[k], prm)], mol_of_bead, n_mol)
        assert _rel(e, z[k + "_energy_ref64"]) < 1e-12 and _rel(f, z[k + "_forces_ref64"]) < 1e-12, k
        terms.append((k, maps[k], prm))
    e, f = po.prior_energy_forces(pos.double(), terms, mol_of_bead, n_mol)
    assert _rel(e, z["total_energy_ref64"]) < 1e-12 and _rel(f, z["total_forces_ref64"]) < 1e-12


def test_more_prior_kinds_oracle_and_host_mirror_match_reference_vectors(golden_dir):
    """HarmonicImpropers, QuarticAngles, RestrictedQuartic: oracle and the host mirror's torch path against the
    reference classes' fp64 outputs; a state_dict with these buffers is recognised by kind."""
    from flashmd_amd import prior

    z, pos, types, ptr, tables, maps = _load(golden_dir, "case_priors_more.npz", MORE)
    n_mol = ptr.numel() - 1
    mol_of_bead = torch.repeat_interleave(torch.arange(n_mol), ptr[1:] - ptr[:-1])
    terms = []
    for k in MORE:
        prm = po.lookup(k, tables[k], types, maps[k])
        e, f = po.prior_energy_forces(pos.double(), [(k, maps[k], prm)], mol_of_bead, n_mol)
        assert _rel(e, z[k + "_energy_ref64"]) < 1e-12 and _rel(f, z[k + "_forces_ref64"]) < 1e-11, k
        terms.append((k, maps[k], prm))
    e, f = po.prior_energy_forces(pos.double(), terms, mol_of_bead, n_mol)
    assert _rel(e, z["total_energy_ref64"]) < 1e-12 and _rel(f, z["total_forces_ref64"]) < 1e-11
    sp = prior.StaticPriors.from_tables({k: (tables[k], maps[k]) for k in MORE}, types, ptr)
    x = pos.double().requires_grad_(True)
    e = sp.energy_torch(x, ptr)
    (g,) = torch.autograd.grad(e.sum(), x)
    assert _rel(e.detach(), z["total_energy_ref64"]) < 1e-6 and _rel(-g, z["total_forces_ref64"]) < 1e-6
    sd = {f"models.{k}.model.{n}": t for k in MORE for n, t in tables[k].items()}
    found = prior.prior_tables_from_state_dict(sd)
    assert {k: v[0] for k, v in found.items()} == {k: k for k in MORE}
    sp2 = prior.StaticPriors.from_state_dict(sd, maps, types, ptr)
    assert _rel(sp2.energy_torch(pos.double(), ptr), z["total_energy_ref64"]) < 1e-6
    # two polynomial models of different degree condense into one list (zero coefficients up to the longer one)
    short = {"v_0": tables["quartic_angles"]["v_0"], "ks": tables["quartic_angles"]["ks"][:2]}
    ma = maps["quartic_angles"]
    hf = ma.shape[1] // 2
    both = prior.StaticPriors.from_tables({"quartic_angles": [(tables["quartic_angles"], ma[:, :hf]), (short, ma[:, hf:])]},
                                          types, ptr)
    parts = [prior.StaticPriors.from_tables({"quartic_angles": (tables["quartic_angles"], ma[:, :hf])}, types, ptr),
             prior.StaticPriors.from_tables({"quartic_angles": (short, ma[:, hf:])}, types, ptr)]
    assert _rel(both.energy_torch(pos.double(), ptr), sum(q.energy_torch(pos.double(), ptr) for q in parts)) < 1e-12


def test_unexported_prior_classes_oracle_and_host_mirror(golden_dir):
    """HarmonicAnglesRaw and ShiftedPeriodicHarmonicImpropers (in prior/harmonic.py, not exported by the package): oracle
    and the host mirror's torch path against the reference's fp64 outputs.  Their buffers look like those of the angle /
    improper classes, so a state_dict cannot tell them apart: they are reached through from_tables only."""
    from flashmd_amd import prior

    z, pos, types, ptr, tables, maps = _load(golden_dir, "case_priors_unexported.npz", UNEXPORTED)
    n_mol = ptr.numel() - 1
    mol_of_bead = torch.repeat_interleave(torch.arange(n_mol), ptr[1:] - ptr[:-1])
    for k in UNEXPORTED:
        prm = po.lookup(k, tables[k], types, maps[k])
        e, f = po.prior_energy_forces(pos.double(), [(k, maps[k], prm)], mol_of_bead, n_mol)
        assert _rel(e, z[k + "_energy_ref64"]) < 1e-12 and _rel(f, z[k + "_forces_ref64"]) < 1e-11, k
    sp = prior.StaticPriors.from_tables({k: (tables[k], maps[k]) for k in UNEXPORTED}, types, ptr)
    x = pos.double().requires_grad_(True)
    e = sp.energy_torch(x, ptr)
    (g,) = torch.autograd.grad(e.sum(), x)
    assert _rel(e.detach(), z["total_energy_ref64"]) < 1e-6 and _rel(-g, z["total_forces_ref64"]) < 1e-6
    # ... and through the state_dict loader with the caller saying which class the ambiguous buffers belong to
    sd = {f"models.{k}.model.{n}": t for k in UNEXPORTED for n, t in tables[k].items()}
    kinds = {"raw_angles": "raw_angles", "shifted_impropers": "shifted_impropers"}
    found = prior.prior_tables_from_state_dict(sd, kinds)
    assert {k: v[0] for k, v in found.items()} == kinds
    sp2 = prior.StaticPriors.from_state_dict(sd, maps, types, ptr, kinds=kinds)
    assert _rel(sp2.energy_torch(pos.double(), ptr), z["total_energy_ref64"]) < 1e-6
    # without the override the raw-angle table (equilibrium angles in radians, beyond 1) is refused, not read as cosines,
    # and the rank-4 pair is read as the exported class with a warning
    assert float(tables["raw_angles"]["x_0"].abs().max()) > 1.0
    with pytest.raises(ValueError, match="cannot be a cosine"):
        prior.prior_tables_from_state_dict({k: v for k, v in sd.items() if ".raw_angles." in k})
    with pytest.warns(UserWarning, match="ShiftedPeriodicHarmonicImpropers"):
        got = prior.prior_tables_from_state_dict({k: v for k, v in sd.items() if ".shifted_impropers." in k})
    assert got["shifted_impropers"][0] == "impropers"
    with pytest.raises(ValueError):
        prior.prior_tables_from_state_dict(sd, {"raw_angles": "bonds", "shifted_impropers": "shifted_impropers"})
    with pytest.raises(KeyError):
        prior.prior_tables_from_state_dict(sd, dict(kinds, nobody="raw_angles"))


def test_prior_host_mirror_torch_path_matches_reference_vectors(golden_dir):
    """the plain-torch branch of the host mirror (what runs with --disable_optim / on the CPU)"""
    from flashmd_amd import prior

    z, pos, types, ptr, tables, maps = _load(golden_dir)
    sp = prior.StaticPriors.from_tables({k: (tables[k], maps[k]) for k in KINDS}, types, ptr)
    x = pos.double().requires_grad_(True)
    e = sp.energy_torch(x, ptr)
    (g,) = torch.autograd.grad(e.sum(), x)
    assert _rel(e.detach(), z["total_energy_ref64"]) < 1e-6 and _rel(-g, z["total_forces_ref64"]) < 1e-6
    with pytest.raises(RuntimeError):  # the HIP form has no CPU fallback
        sp.energy_forces(pos, ptr)


@pytest.mark.gpu
def test_prior_hip_vs_reference_vectors(golden_dir):
    from flashmd_amd import prior

    z, pos, types, ptr, tables, maps = _load(golden_dir)
    dev = "cuda:0"
    sp = prior.StaticPriors.from_tables({k: (tables[k], maps[k]) for k in KINDS}, types, ptr).to(dev)
    e, f = sp.energy_forces(pos.to(dev), ptr.to(dev))
    re, rf = _rel(e.cpu(), z["total_energy_ref64"]), _rel(f.cpu(), z["total_forces_ref64"])
    print(f"priors (HIP, fp32) vs reference fp64: energy {re:.2e} forces {rf:.2e}")
    assert re < 1e-5 and rf < 1e-5
    e2, f2 = sp.energy_forces(pos.to(dev), ptr.to(dev))
    assert torch.equal(e, e2) and torch.equal(f, f2)  # fixed summation order
    for k in KINDS:  # each kind alone
        one = prior.StaticPriors.from_tables({k: (tables[k], maps[k])}, types, ptr).to(dev)
        ek, fk = one.energy_forces(pos.to(dev), ptr.to(dev))
        assert _rel(ek.cpu(), z[k + "_energy_ref64"]) < 1e-5 and _rel(fk.cpu(), z[k + "_forces_ref64"]) < 2e-5, k


@pytest.mark.gpu
def test_more_prior_kinds_hip_vs_reference_vectors(golden_dir):
    """kinds 4-6 of fmd_prior_energy_forces against the reference classes' fp64 outputs, each alone and summed, and on
    269-bead chains against the fp64 oracle"""
    from flashmd_amd import prior, synthetic

    z, pos, types, ptr, tables, maps = _load(golden_dir, "case_priors_more.npz", MORE)
    dev = "cuda:0"
    sp = prior.StaticPriors.from_tables({k: (tables[k], maps[k]) for k in MORE}, types, ptr).to(dev)
    e, f = sp.energy_forces(pos.to(dev), ptr.to(dev))
    re, rf = _rel(e.cpu(), z["total_energy_ref64"]), _rel(f.cpu(), z["total_forces_ref64"])
    print(f"impropers + quartic + restricted angles (HIP, fp32) vs reference fp64: energy {re:.2e} forces {rf:.2e}")
    assert re < 1e-5 and rf < 2e-5
    e2, f2 = sp.energy_forces(pos.to(dev), ptr.to(dev))
    assert torch.equal(e, e2) and torch.equal(f, f2)
    for k in MORE:
        one = prior.StaticPriors.from_tables({k: (tables[k], maps[k])}, types, ptr).to(dev)
        ek, fk = one.energy_forces(pos.to(dev), ptr.to(dev))
        rek, rfk = _rel(ek.cpu(), z[k + "_energy_ref64"]), _rel(fk.cpu(), z[k + "_forces_ref64"])
        print(f"  {k}: energy {rek:.2e} forces {rfk:.2e}")
        assert rek < 1e-5 and rfk < 3e-5, k
    n_mol, A, T = 8, 269, 5
    pos, types, ptr = synthetic.chain_batch(n_mol, A, T)
    topo = synthetic.chain_topology(n_mol, A)
    big = {"impropers": topo["dihedrals"], "quartic_angles": topo["angles"], "restricted_angles": topo["angles"]}
    mol_of_bead = torch.arange(n_mol).repeat_interleave(A)
    terms = [(k, big[k], po.lookup(k, tables[k], types, big[k]).double()) for k in MORE]
    e_o, f_o = po.prior_energy_forces(pos.double(), terms, mol_of_bead, n_mol)
    sp = prior.StaticPriors.from_tables({k: (tables[k], big[k]) for k in MORE}, types, ptr).to(dev)
    e, f = sp.energy_forces(pos.to(dev), ptr.to(dev))
    assert _rel(e.cpu(), e_o) < 1e-5 and _rel(f.cpu(), f_o) < 3e-5


@pytest.mark.gpu
def test_unexported_prior_classes_hip_vs_reference_vectors(golden_dir):
    """kinds 7 and 8 of fmd_prior_energy_forces against the reference's fp64 outputs"""
    from flashmd_amd import prior

    z, pos, types, ptr, tables, maps = _load(golden_dir, "case_priors_unexported.npz", UNEXPORTED)
    dev = "cuda:0"
    sp = prior.StaticPriors.from_tables({k: (tables[k], maps[k]) for k in UNEXPORTED}, types, ptr).to(dev)
    e, f = sp.energy_forces(pos.to(dev), ptr.to(dev))
    re, rf = _rel(e.cpu(), z["total_energy_ref64"]), _rel(f.cpu(), z["total_forces_ref64"])
    print(f"raw angles + shifted impropers (HIP, fp32) vs reference fp64: energy {re:.2e} forces {rf:.2e}")
    assert re < 1e-5 and rf < 3e-5
    for k in UNEXPORTED:
        one = prior.StaticPriors.from_tables({k: (tables[k], maps[k])}, types, ptr).to(dev)
        ek, fk = one.energy_forces(pos.to(dev), ptr.to(dev))
        assert _rel(ek.cpu(), z[k + "_energy_ref64"]) < 1e-5 and _rel(fk.cpu(), z[k + "_forces_ref64"]) < 3e-5, k


@pytest.mark.gpu
def test_prior_hip_1enh_sized_batch_vs_oracle():
    """269-bead chains x 16 molecules (all four kinds, ~36k terms): HIP vs the fp64 oracle; empty term lists and
    molecules without terms are accepted."""
    from flashmd_amd import prior, synthetic

    dev = "cuda:0"
    n_mol, A, T = 16, 269, 5
    pos, types, ptr = synthetic.chain_batch(n_mol, A, T)
    topo = synthetic.chain_topology(n_mol, A, repulsion_stride=1)  # 35k pairs per chain: the partner-list path
    g = torch.Generator().manual_seed(3)
    tables = {"bonds": {"k": 2 + torch.rand(T, T, generator=g), "x_0": 3.6 + 0.4 * torch.rand(T, T, generator=g)},
              "angles": {"k": 1 + torch.rand(T, T, T, generator=g), "x_0": torch.rand(T, T, T, generator=g) - 0.5},
              "repulsion": {"sigma": 3 + torch.rand(T, T, generator=g)},
              "dihedrals": {"v_0": torch.rand(T, T, T, T, generator=g), "k1s": torch.rand(2, T, T, T, T, generator=g) - 0.5,
                            "k2s": torch.rand(2, T, T, T, T, generator=g) - 0.5}}
    mol_of_bead = torch.arange(n_mol).repeat_interleave(A)
    terms = [(k, topo[k], po.lookup(k, tables[k], types, topo[k])) for k in KINDS]
    e_o, f_o = po.prior_energy_forces(pos.double(), [(k, m, p.double()) for k, m, p in terms], mol_of_bead, n_mol)
    sp = prior.StaticPriors.from_tables({k: (tables[k], topo[k]) for k in KINDS}, types, ptr).to(dev)
    e, f = sp.energy_forces(pos.to(dev), ptr.to(dev))
    assert _rel(e.cpu(), e_o) < 1e-5 and _rel(f.cpu(), f_o) < 1e-5
    # bonds only in the first molecule: the others get zero
    first = topo["bonds"][:, : A - 1]
    sp1 = prior.StaticPriors.from_tables({"bonds": (tables["bonds"], first)}, types, ptr).to(dev)
    e1, f1 = sp1.energy_forces(pos.to(dev), ptr.to(dev))
    assert float(e1[1:].abs().max()) == 0.0 and float(f1[A:].abs().max()) == 0.0 and float(e1[0]) > 0
    empty = prior.StaticPriors.from_tables({"bonds": (tables["bonds"], first[:, :0])}, types, ptr).to(dev)
    e0, f0 = empty.energy_forces(pos.to(dev), ptr.to(dev))
    assert float(e0.abs().max()) == 0.0 and float(f0.abs().max()) == 0.0


@pytest.mark.gpu
def test_schnet_plus_priors_nve(golden_dir):
    """SumOut analogue: fused SchNet engine + condensed priors as one force field; NVE total energy is conserved
    (the prior forces are the gradient of the prior energies)."""
    import flashmd_amd.schnet as S
    from flashmd_amd import engine, prior, simulate, synthetic
    from oracle import schnet_oracle as so

    dev = "cuda:0"
    params = so.params_from_npz(os.path.join(golden_dir, "weights_L3_R50.npz"))
    n_mol, A, T = 4, 120, 5
    pos, types, ptr = synthetic.chain_batch(n_mol, A, T)
    topo = synthetic.chain_topology(n_mol, A)
    tables = {"bonds": {"k": torch.full((T, T), 20.0), "x_0": torch.full((T, T), 3.8)},
              "angles": {"k": torch.full((T, T, T), 5.0), "x_0": torch.full((T, T, T), -0.3)},
              "repulsion": {"sigma": torch.full((T, T), 3.5)},
              "dihedrals": {"v_0": torch.zeros(T, T, T, T), "k1s": 0.3 * torch.ones(2, T, T, T, T),
                            "k2s": -0.2 * torch.ones(2, T, T, T, T)}}
    sp = prior.StaticPriors.from_tables({k: (tables[k], topo[k]) for k in KINDS}, types, ptr).to(dev)
    eng = engine.FusedSchNetEngine(S.model_from_params(params).to(dev))
    P, Tt, Q = pos.to(dev), (types + 1).to(dev), ptr.to(dev)
    ff = prior.sum_out([lambda x: eng.energy_forces(x, Tt, Q, A), (sp, Q)])
    masses = torch.full((P.shape[0],), 12.0, device=dev)
    sim = simulate.NVESimulation(ff, P, masses, torch.full((n_mol,), 1.67, device=dev), A, dt=0.002)
    tot = []
    for _ in range(200):
        sim.timestep()
        tot.append((sim.energy + sim.kinetic_energy()).double().cpu())
    tot = torch.stack(tot)
    ke = float(sim.kinetic_energy().double().mean())
    drift = float(((tot.max(0).values - tot.min(0).values) / ke).max())
    print(f"SchNet + priors NVE: (max-min of E_tot)/KE = {drift:.2e}")
    assert drift < 2e-2
    assert int(sp.status[3]) == 0


def test_several_priors_of_one_kind_are_condensed_like_the_reference(golden_dir):
    """A SumOut may hold several models of one kind under different names (the reference's GeneralBonds /
    GeneralAngles); condense_all_priors_for_simulation concatenates their mappings and looked-up parameters and pads
    Fourier series to the longest expansion (simulation/specialize_prior.py:159-196).  Here: the fixture's bond terms
    split over two models, its dihedrals over a full-length and a one-term series -- the condensed energy / forces must
    be the sum of the models evaluated one by one (each of which the fixture pins), from tables and from a
    state_dict with two bond entries."""
    from flashmd_amd import prior

    z, pos, types, ptr, tables, maps = _load(golden_dir)
    mb, md = maps["bonds"], maps["dihedrals"]
    hb, hd = mb.shape[1] // 2, md.shape[1] // 3
    short = {"v_0": tables["dihedrals"]["v_0"], "k1s": tables["dihedrals"]["k1s"][:1], "k2s": tables["dihedrals"]["k2s"][:1]}
    assert tables["dihedrals"]["k1s"].shape[0] > 1

    def ef(sp):
        x = pos.double().requires_grad_(True)
        e = sp.energy_torch(x, ptr)
        (g,) = torch.autograd.grad(e.sum(), x)
        return e.detach(), -g

    parts = [prior.StaticPriors.from_tables({"bonds": (tables["bonds"], mb[:, :hb])}, types, ptr),
             prior.StaticPriors.from_tables({"bonds": (tables["bonds"], mb[:, hb:])}, types, ptr),
             prior.StaticPriors.from_tables({"dihedrals": (tables["dihedrals"], md[:, :hd])}, types, ptr),
             prior.StaticPriors.from_tables({"dihedrals": (short, md[:, hd:])}, types, ptr)]
    e_sum = sum(ef(p)[0] for p in parts)
    f_sum = sum(ef(p)[1] for p in parts)
    both = prior.StaticPriors.from_tables({"bonds": [(tables["bonds"], mb[:, :hb]), (tables["bonds"], mb[:, hb:])],
                                           "dihedrals": [(tables["dihedrals"], md[:, :hd]), (short, md[:, hd:])]},
                                          types, ptr)
    assert both.terms["bonds"][0].shape[1] == mb.shape[1] and both.terms["dihedrals"][1].shape[0] == \
        1 + 2 * tables["dihedrals"]["k1s"].shape[0]
    e, f = ef(both)
    assert _rel(e, e_sum) < 1e-12 and _rel(f, f_sum) < 1e-12
    # the split bonds alone reproduce the fixture's single bond model
    two = prior.StaticPriors.from_tables({"bonds": [(tables["bonds"], mb[:, :hb]), (tables["bonds"], mb[:, hb:])]}, types, ptr)
    assert _rel(ef(two)[0], z["bonds_energy_ref64"]) < 1e-6
    # the same container as a state_dict: two bond entries under different names
    sd = {"models.bonds_a.model.x_0": tables["bonds"]["x_0"], "models.bonds_a.model.k": tables["bonds"]["k"],
          "models.bonds_b.model.x_0": tables["bonds"]["x_0"], "models.bonds_b.model.k": tables["bonds"]["k"]}
    sp = prior.StaticPriors.from_state_dict(sd, {"bonds_a": mb[:, :hb], "bonds_b": mb[:, hb:]}, types, ptr)
    assert _rel(ef(sp)[0], z["bonds_energy_ref64"]) < 1e-6
    with pytest.raises(KeyError):
        prior.StaticPriors.from_state_dict(sd, {"bonds_a": mb[:, :hb]}, types, ptr)
