"""Generate tests/golden/case_priors.npz with the REFERENCE's own prior classes (build container only):
``HarmonicBonds``, ``HarmonicAngles``, ``Repulsion``, ``Dihedral`` from /root/reference/src/flashmd/prior,
evaluated in fp64 on synthetic chains; forces by autograd of the summed energies, as ``GradientsOut`` does.
The fixture holds inputs (positions, types, mappings, per-type parameter tables) and the reference's outputs.

    python oracle/make_golden_priors.py          # case_priors.npz
    python oracle/make_golden_priors.py more     # case_priors_more.npz: HarmonicImpropers, QuarticAngles, RestrictedQuartic
    python oracle/make_golden_priors.py unexported   # case_priors_unexported.npz: the two classes prior/__init__.py does not export

``more``: ``QuarticAngles`` (prior/polynomial.py:157-186) and ``RestrictedQuartic`` (prior/restricted_bending.py) run
through their own ``forward``.  ``HarmonicImpropers.forward`` cannot run as written -- its ``data2features`` is declared
``@staticmethod`` with a ``self`` parameter (prior/harmonic.py:309-312), so ``Harmonic.forward``'s call raises a
TypeError -- and the fixture is made from the pieces of the class that do run: ``data2parameters`` (the per-type
lookup), ``HarmonicImpropers.compute_features`` (= compute_torsions) and ``Harmonic.compute``, summed per molecule the
way ``Harmonic.forward`` does (:109-113).

``unexported``: ``ShiftedPeriodicHarmonicImpropers`` (prior/harmonic.py:325-400) through its own ``forward``.
``HarmonicAnglesRaw`` (:264-298) cannot even be constructed as written (its ``__init__`` calls ``Harmonic.__init__`` without
``order``: TypeError); its fixture is ``Harmonic(statistics, "raw_angles", order=3).data2parameters`` (the lookup),
``HarmonicAnglesRaw.compute_features`` (= compute_angles_raw) and ``Harmonic.compute``.
"""
import os
import sys
import types as pytypes

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "flash-molecular-dynamics_amd"))
from oracle import make_golden as mg  # noqa: E402  (stand-ins for nvtx / torch_geometric / torch_cluster)

mg._install_standins()
sys.path.insert(0, "/root/reference/src")
from flashmd.prior.fourier_series import Dihedral  # noqa: E402
from flashmd.prior.harmonic import HarmonicAngles, HarmonicBonds  # noqa: E402
from flashmd.prior.repulsion import Repulsion  # noqa: E402

from flashmd_amd import synthetic  # noqa: E402  (input generator only: no product math)


def main():
    torch.manual_seed(5)
    n_mol, A, T, n_degs = 3, 41, 5, 3
    pos, types, ptr = synthetic.chain_batch(n_mol, A, T)
    topo = synthetic.chain_topology(n_mol, A)
    g = torch.Generator().manual_seed(9)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)  # noqa: E731
    stats = {
        "bonds": {(a, b): {"k": 2 + 3 * r(1)[0], "x_0": 3.6 + 0.4 * r(1)[0]} for a in range(T) for b in range(T)},
        "angles": {(a, b, c): {"k": 1 + 4 * r(1)[0], "x_0": -0.6 + 1.2 * r(1)[0]}
                   for a in range(T) for b in range(T) for c in range(T)},
        "repulsion": {(a, b): {"sigma": 3.0 + 1.5 * r(1)[0]} for a in range(T) for b in range(T)},
        "dihedrals": {(a, b, c, d): {"v_0": r(1)[0], "k1s": {f"k1_{n}": r(1)[0] - 0.5 for n in range(1, n_degs + 1)},
                                     "k2s": {f"k2_{n}": r(1)[0] - 0.5 for n in range(1, n_degs + 1)}}
                      for a in range(T) for b in range(T) for c in range(T) for d in range(T)},
    }
    models = {"bonds": HarmonicBonds(stats["bonds"]), "angles": HarmonicAngles(stats["angles"]),
              "repulsion": Repulsion(stats["repulsion"]), "dihedrals": Dihedral(stats["dihedrals"], n_degs=n_degs)}
    mol_of_bead = torch.arange(n_mol).repeat_interleave(A)
    data = pytypes.SimpleNamespace(pos=pos.double().requires_grad_(True), atom_types=types, ptr=ptr, out={},
                                   neighbor_list={k: {"index_mapping": topo[k], "mapping_batch": mol_of_bead[topo[k][0]]}
                                                  for k in topo})
    out = {"pos": pos.numpy(), "types": types.numpy(), "ptr": ptr.numpy(), "n_degs": np.int64(n_degs)}
    total = 0
    for k, m in models.items():
        m = m.double()
        data = m(data)
        e = data.out[m.name]["energy"]
        (gr,) = torch.autograd.grad(e.sum(), data.pos, retain_graph=True)
        out[f"{k}_mapping"] = topo[k].numpy()
        out[f"{k}_energy_ref64"] = e.detach().numpy()
        out[f"{k}_forces_ref64"] = (-gr).numpy()
        total = total + e
        for name, buf in m.named_buffers():
            out[f"{k}_table_{name}"] = buf.numpy()
    (gr,) = torch.autograd.grad(total.sum(), data.pos)
    out["total_energy_ref64"] = total.detach().numpy()
    out["total_forces_ref64"] = (-gr).numpy()
    path = os.path.join(ROOT, "tests", "golden", "case_priors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def more():
    from flashmd.prior.harmonic import Harmonic, HarmonicImpropers
    from flashmd.prior.polynomial import QuarticAngles
    from flashmd.prior.restricted_bending import RestrictedQuartic

    n_mol, A, T, n_degs = 3, 41, 5, 4
    pos, types, ptr = synthetic.chain_batch(n_mol, A, T)
    topo = synthetic.chain_topology(n_mol, A)
    g = torch.Generator().manual_seed(21)
    r = lambda: torch.rand(1, generator=g, dtype=torch.float64)[0]  # noqa: E731
    tri = [(a, b, c) for a in range(T) for b in range(T) for c in range(T)]
    stats = {
        "impropers": {(a, b, c, d): {"k": 0.5 + 2 * r(), "x_0": -1.0 + 2.0 * r()}
                      for a in range(T) for b in range(T) for c in range(T) for d in range(T)},
        "quartic_angles": {key: {"v_0": r(), "ks": {f"k_{n}": 2 * r() - 1 for n in range(1, n_degs + 1)}} for key in tri},
        "restricted_angles": {key: {"a": 2 * r() - 1, "b": 2 * r() - 1, "c": 2 * r() - 1, "d": 2 * r() - 1,
                                    "k": 0.05 + 0.2 * r(), "v_0": r()} for key in tri},
    }
    mappings = {"impropers": topo["dihedrals"], "quartic_angles": topo["angles"], "restricted_angles": topo["angles"]}
    models = {"impropers": HarmonicImpropers(stats["impropers"]),
              "quartic_angles": QuarticAngles(stats["quartic_angles"], name="quartic_angles", n_degs=n_degs),
              "restricted_angles": RestrictedQuartic(stats["restricted_angles"], name="restricted_angles")}
    mol_of_bead = torch.arange(n_mol).repeat_interleave(A)
    nl = {k: {"index_mapping": mappings[k], "mapping_batch": mol_of_bead[mappings[k][0]]} for k in mappings}
    data = pytypes.SimpleNamespace(pos=pos.double().requires_grad_(True), atom_types=types, ptr=ptr, out={}, neighbor_list=nl)
    out = {"pos": pos.numpy(), "types": types.numpy(), "ptr": ptr.numpy(), "n_degs": np.int64(n_degs)}
    total = 0
    for k, m in models.items():
        m = m.double()
        if k == "impropers":
            try:
                m(data)
                raise AssertionError("HarmonicImpropers.forward runs now: make the fixture from it")
            except TypeError:
                pass
            prm = m.data2parameters(data)
            y = Harmonic.compute(HarmonicImpropers.compute_features(data.pos, mappings[k]).flatten(), **prm)
            e = torch.zeros(n_mol, dtype=torch.float64).index_add(0, nl[k]["mapping_batch"], y)
        else:
            data = m(data)
            e = data.out[m.name]["energy"]
        (gr,) = torch.autograd.grad(e.sum(), data.pos, retain_graph=True)
        out[f"{k}_mapping"] = mappings[k].numpy()
        out[f"{k}_energy_ref64"] = e.detach().numpy()
        out[f"{k}_forces_ref64"] = (-gr).numpy()
        total = total + e
        for name, buf in m.named_buffers():
            out[f"{k}_table_{name}"] = buf.numpy()
    (gr,) = torch.autograd.grad(total.sum(), data.pos)
    out["total_energy_ref64"] = total.detach().numpy()
    out["total_forces_ref64"] = (-gr).numpy()
    path = os.path.join(ROOT, "tests", "golden", "case_priors_more.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})
    print("energies:", {k: out[f"{k}_energy_ref64"] for k in models})


def unexported():
    from flashmd.prior.harmonic import Harmonic, HarmonicAnglesRaw, ShiftedPeriodicHarmonicImpropers

    n_mol, A, T = 3, 41, 5
    pos, types, ptr = synthetic.chain_batch(n_mol, A, T)
    topo = synthetic.chain_topology(n_mol, A)
    g = torch.Generator().manual_seed(33)
    r = lambda: torch.rand(1, generator=g, dtype=torch.float64)[0]  # noqa: E731
    stats = {"raw_angles": {(a, b, c): {"k": 0.5 + 2 * r(), "x_0": 1.2 + 1.2 * r()}
                            for a in range(T) for b in range(T) for c in range(T)},
             "shifted_impropers": {(a, b, c, d): {"k": 0.5 + 2 * r(), "x_0": -1.0 + 2.0 * r()}
                                   for a in range(T) for b in range(T) for c in range(T) for d in range(T)}}
    mappings = {"raw_angles": topo["angles"], "shifted_impropers": topo["dihedrals"]}
    try:
        HarmonicAnglesRaw(stats["raw_angles"], "raw_angles")
        raise AssertionError("HarmonicAnglesRaw can be constructed now: make the fixture from its forward")
    except TypeError:
        pass
    models = {"raw_angles": Harmonic(stats["raw_angles"], "raw_angles", order=3),
              "shifted_impropers": ShiftedPeriodicHarmonicImpropers(stats["shifted_impropers"])}
    models["shifted_impropers"].name = "shifted_impropers"   # (the neighbour-list key; the class default is "impropers")
    mol_of_bead = torch.arange(n_mol).repeat_interleave(A)
    nl = {k: {"index_mapping": mappings[k], "mapping_batch": mol_of_bead[mappings[k][0]]} for k in mappings}
    data = pytypes.SimpleNamespace(pos=pos.double().requires_grad_(True), atom_types=types, ptr=ptr, out={}, neighbor_list=nl)
    out = {"pos": pos.numpy(), "types": types.numpy(), "ptr": ptr.numpy()}
    total = 0
    for k, m in models.items():
        m = m.double()
        if k == "raw_angles":
            prm = m.data2parameters(data)
            y = Harmonic.compute(HarmonicAnglesRaw.compute_features(data.pos, mappings[k]).flatten(), **prm)
            e = torch.zeros(n_mol, dtype=torch.float64).index_add(0, nl[k]["mapping_batch"], y)
        else:
            data = m(data)
            e = data.out[m.name]["energy"]
        (gr,) = torch.autograd.grad(e.sum(), data.pos, retain_graph=True)
        out[f"{k}_mapping"] = mappings[k].numpy()
        out[f"{k}_energy_ref64"] = e.detach().numpy()
        out[f"{k}_forces_ref64"] = (-gr).numpy()
        total = total + e
        for name, buf in m.named_buffers():
            out[f"{k}_table_{name}"] = buf.numpy()
    (gr,) = torch.autograd.grad(total.sum(), data.pos)
    out["total_energy_ref64"] = total.detach().numpy()
    out["total_forces_ref64"] = (-gr).numpy()
    path = os.path.join(ROOT, "tests", "golden", "case_priors_unexported.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "energies:", {k: out[f"{k}_energy_ref64"] for k in models})


if __name__ == "__main__":
    arg = sys.argv[1:]
    unexported() if "unexported" in arg else more() if "more" in arg else main()
