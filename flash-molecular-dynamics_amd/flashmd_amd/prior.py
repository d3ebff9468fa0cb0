"""Classical prior terms (row N2): host mirror of the reference's prior classes and their condensed form.

``HarmonicBonds`` / ``HarmonicAngles`` / ``HarmonicImpropers`` / ``Repulsion`` / ``Dihedral`` / ``QuarticAngles`` /
``RestrictedQuartic`` keep the reference's static ``compute`` / ``compute_features`` (reference
prior/harmonic.py:116-123,221-225,258-262,301-322, prior/repulsion.py:112-122, prior/fourier_series.py:154-192,
prior/polynomial.py:127-186, prior/restricted_bending.py:200-238; features
geometry/internal_coordinates.py:72-223) as plain torch ops -- the --disable_optim / CPU path.
``StaticPriors`` is the condensed form the simulation uses (reference simulation/specialize_prior.py:76-207):
per-term parameter arrays, evaluated with energy and forces in one HIP kernel per kind
(``fmd_prior_energy_forces``); there is no CPU fallback for it.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Tuple

import torch

from . import _ffi
from ._ffi import call, ptr, stream

KIND = {"bonds": 0, "angles": 1, "repulsion": 2, "dihedrals": 3, "impropers": 4, "quartic_angles": 5,
        "restricted_angles": 6, "raw_angles": 7, "shifted_impropers": 8}
ORDER = {"bonds": 2, "angles": 3, "repulsion": 2, "dihedrals": 4, "impropers": 4, "quartic_angles": 3,
         "restricted_angles": 3, "raw_angles": 3, "shifted_impropers": 4}


def compute_distances(pos, mapping):
    return (pos[mapping[1]] - pos[mapping[0]]).norm(p=2, dim=1)


def compute_angles_cos(pos, mapping):
    dr1, dr2 = pos[mapping[0]] - pos[mapping[1]], pos[mapping[2]] - pos[mapping[1]]
    return (dr1 * dr2).sum(dim=1) / (dr1.norm(p=2, dim=1) * dr2.norm(p=2, dim=1))


def compute_angles_raw(pos, mapping):
    dr1, dr2 = pos[mapping[0]] - pos[mapping[1]], pos[mapping[2]] - pos[mapping[1]]
    return torch.atan2(torch.cross(dr1, dr2, dim=1).norm(p=2, dim=1), (dr1 * dr2).sum(dim=1))


def compute_torsions(pos, mapping):
    nrm = torch.nn.functional.normalize
    dr1 = nrm(pos[mapping[1]] - pos[mapping[0]], dim=1)
    dr2 = nrm(pos[mapping[2]] - pos[mapping[1]], dim=1)
    dr3 = nrm(pos[mapping[3]] - pos[mapping[2]], dim=1)
    n1, n2 = torch.cross(dr1, dr2, dim=1), torch.cross(dr2, dr3, dim=1)
    m1 = torch.cross(n1, dr2, dim=1)
    return torch.atan2(-(m1 * n2).sum(-1), (n1 * n2).sum(-1))


class HarmonicBonds:
    name, _order = "bonds", 2
    compute_features = staticmethod(compute_distances)

    @staticmethod
    def compute(x, x0, k, V0=0):
        return k * (x - x0) ** 2 + V0


class HarmonicAngles(HarmonicBonds):
    name, _order = "angles", 3
    compute_features = staticmethod(compute_angles_cos)


class HarmonicImpropers(HarmonicBonds):
    name, _order = "impropers", 4
    compute_features = staticmethod(compute_torsions)


class HarmonicAnglesRaw(HarmonicBonds):
    """Harmonic in the raw angle (reference prior/harmonic.py:264-298; not exported by the package)."""
    name, _order = "angles", 3
    compute_features = staticmethod(compute_angles_raw)


class ShiftedPeriodicHarmonicImpropers(HarmonicBonds):
    """reference prior/harmonic.py:325-400: torsions below zero are shifted by 2 pi, then pi is subtracted."""
    name, _order = "impropers", 4

    @staticmethod
    def compute_features(pos, mapping):
        f = compute_torsions(pos, mapping)
        pi32 = float(torch.tensor(torch.pi, dtype=torch.float32))   # the reference's torch_pi is a float32 tensor (:19)
        return torch.where(f < 0, f + 2 * pi32, f) - pi32


class QuarticAngles:
    """Polynomial of the angle's cosine (reference prior/polynomial.py:127-186): V = V0 + sum_n k_n x^n."""
    name, _order = "angles", 3
    compute_features = staticmethod(compute_angles_cos)

    @staticmethod
    def compute(x, ks, V0):
        V, x_pow = ks[0] * x, x
        for k in ks[1:]:
            x_pow = x_pow * x
            V = V + k * x_pow
        return V + V0


class RestrictedQuartic:
    """reference prior/restricted_bending.py:200-238, on the raw angle."""
    name, _order = "angles", 3
    compute_features = staticmethod(compute_angles_raw)

    @staticmethod
    def compute(x, a, b, c, d, k, v_0):
        cos, sin = torch.cos(x), torch.sin(x)
        return a * cos ** 4 + b * cos ** 3 + c * cos ** 2 + d * cos + k / (sin ** 2) + v_0


class Repulsion:
    name, _order = "repulsion", 2
    compute_features = staticmethod(compute_distances)

    @staticmethod
    def compute(x, sigma):
        rr = (sigma / x) * (sigma / x)
        return rr * rr * rr


class Dihedral:
    name, _order = "dihedrals", 4
    compute_features = staticmethod(compute_torsions)

    @staticmethod
    def compute(theta, v_0, k1s, k2s):
        n_degs = torch.arange(1, k1s.shape[1] + 1, dtype=theta.dtype, device=theta.device)
        angles = theta.view(-1, 1) * n_degs.view(1, -1)
        return (k1s * torch.sin(angles) + k2s * torch.cos(angles)).sum(dim=1) + v_0.view(-1)


def _lookup(kind: str, tables: Dict[str, torch.Tensor], types, mapping):
    """per-term parameters [P, M] from per-type tables (what data2parameters does in the reference)."""
    idx = tuple(types[mapping[i]] for i in range(mapping.shape[0]))
    if kind in ("bonds", "angles", "impropers", "raw_angles", "shifted_impropers"):
        return torch.stack([tables["k"][idx], tables["x_0"][idx]])
    if kind == "repulsion":
        return tables["sigma"][idx].unsqueeze(0)
    if kind == "quartic_angles":     # rows v_0, k_1 .. k_n  (Polynomial.ks is [n_degs, *types])
        return torch.cat([tables["v_0"][idx].unsqueeze(0)] + [tables["ks"][i][idx].unsqueeze(0)
                                                              for i in range(tables["ks"].shape[0])])
    if kind == "restricted_angles":
        return torch.stack([tables[n][idx] for n in ("a", "b", "c", "d", "k", "v_0")])
    n = tables["k1s"].shape[0]
    return torch.cat([tables["v_0"][idx].unsqueeze(0)] + [tables["k1s"][i][idx].unsqueeze(0) for i in range(n)]
                     + [tables["k2s"][i][idx].unsqueeze(0) for i in range(n)])


# kinds whose buffers look the same in a state_dict: the file alone cannot tell them apart
_AMBIGUOUS = {"angles": ("angles", "raw_angles"), "impropers": ("impropers", "shifted_impropers")}


def prior_tables_from_state_dict(sd, kinds: Dict[str, str] = None) -> Dict[str, Tuple[str, Dict[str, torch.Tensor]]]:
    """The prior models of a reference-format container, read from its ``state_dict``: name -> (kind, per-type tables).
    The reference's prior classes keep their parameters as BUFFERS indexed by bead types: ``Harmonic.x_0`` / ``.k``
    (prior/harmonic.py:60-77; order 2 = bonds, order 3 = angles), ``Repulsion.sigma`` (prior/repulsion.py:52-60),
    ``FourierSeries.k1s`` / ``.k2s`` / ``.v_0`` (prior/fourier_series.py:60-80; order 4 = dihedrals), ``Harmonic``
    of order 4 = impropers, ``Polynomial.ks`` / ``.v_0`` (prior/polynomial.py:62-75; order 3 = quartic angles),
    ``RestrictedQuartic.a`` .. ``.d`` / ``.k`` / ``.v_0`` (prior/restricted_bending.py:68-88), each under
    ``<prefix>.<name>.model.`` in ``SumOut.models`` (models/gradients.py:64-69).  The kind is told by the buffer set and
    its rank, the name is the ModuleDict key -- the key of ``data.neighbor_list`` the term's index mapping lives
    under.  The topology itself is not in the file: ``StaticPriors.from_state_dict`` takes it from the caller.

    AMBIGUITY: ``HarmonicAnglesRaw`` (harmonic in theta, prior/harmonic.py:264-298) keeps the same ``x_0`` / ``k`` buffers
    of rank 3 as ``HarmonicAngles`` (harmonic in cos theta), and ``ShiftedPeriodicHarmonicImpropers`` (:325-400) the same
    rank-4 pair as ``HarmonicImpropers``; the class is not in a ``state_dict``.  The default reading is the exported one
    (``angles``, ``impropers``); ``kinds={name: "raw_angles"}`` (or ``"shifted_impropers"``) overrides it per prior name.
    A rank-3 ``x_0`` outside [-1, 1] cannot be a cosine: that table raises unless the caller has said what it is."""
    import warnings

    kinds = dict(kinds or {})
    groups: Dict[str, Dict[str, torch.Tensor]] = {}
    for key, v in sd.items():
        head, _, leaf = key.rpartition(".")
        if leaf in ("x_0", "k", "sigma", "k1s", "k2s", "v_0", "ks", "a", "b", "c", "d"):
            groups.setdefault(head, {})[leaf] = v if isinstance(v, torch.Tensor) else torch.as_tensor(v)
    out = {}
    for head, bufs in groups.items():
        parts = head.split(".")
        name = parts[-2] if parts[-1] == "model" and len(parts) >= 2 else parts[-1]
        names = set(bufs)
        if names == {"x_0", "k"} and bufs["k"].dim() == 2:
            kind = "bonds"
        elif names == {"x_0", "k"} and bufs["k"].dim() == 3:
            kind = "angles"
        elif names == {"x_0", "k"} and bufs["k"].dim() == 4:
            kind = "impropers"
        elif names == {"ks", "v_0"} and bufs["v_0"].dim() == 3:
            kind = "quartic_angles"
        elif names == {"a", "b", "c", "d", "k", "v_0"} and bufs["v_0"].dim() == 3:
            kind = "restricted_angles"
        elif names == {"sigma"} and bufs["sigma"].dim() == 2:
            kind = "repulsion"
        elif names == {"k1s", "k2s", "v_0"} and bufs["v_0"].dim() == 4:
            kind = "dihedrals"
        else:
            raise ValueError(f"state_dict entry {head!r} looks like a prior ({sorted(names)}, rank "
                             f"{[int(b.dim()) for b in bufs.values()]}) of a kind this build does not evaluate")
        if name in kinds:
            want = kinds.pop(name)
            if want not in _AMBIGUOUS.get(kind, (kind,)):
                raise ValueError(f"prior {name!r} has the buffers of {_AMBIGUOUS.get(kind, (kind,))}, not of {want!r}")
            kind = want
        elif kind == "angles":
            x0 = bufs["x_0"][bufs["k"] != 0] if bufs["k"].shape == bufs["x_0"].shape else bufs["x_0"]
            if x0.numel() and float(x0.abs().max()) > 1.0 + 1e-6:
                raise ValueError(f"prior {name!r}: x_0 reaches {float(x0.abs().max()):.3f}, which cannot be a cosine -- "
                                 "a HarmonicAnglesRaw table?  Pass kinds={%r: 'raw_angles'} (or 'angles' to insist)" % name)
        elif kind == "impropers":
            warnings.warn(f"prior {name!r}: rank-4 Harmonic buffers are read as HarmonicImpropers; a "
                          "ShiftedPeriodicHarmonicImpropers model has the same buffers -- pass "
                          "kinds={%r: 'shifted_impropers'} if that is what the file holds" % name, stacklevel=2)
        out[name] = (kind, bufs)
    if kinds:
        raise KeyError(f"kinds names priors the state_dict does not hold: {sorted(kinds)}")
    return out


class StaticPriors:
    """All prior terms of a collated batch, condensed: for each kind ``mapping`` [order, M] (int64, global bead
    indices), ``params`` [P, M] (fp32) and the by-bead CSR of the fixed topology (``bead_ptr`` [N+1] int64,
    ``entries`` [order*M] int32 = term << 2 | slot) the gather kernel walks."""

    def __init__(self, terms: Dict[str, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]], max_beads: int):
        self.terms, self.max_beads = terms, int(max_beads)
        self.status = None
        self._checked = False

    @classmethod
    def from_tables(cls, spec: Dict[str, Tuple[Dict[str, torch.Tensor], torch.Tensor]], types, mol_ptr):
        """spec: kind -> (per-type parameter tables, mapping [order, M]) -- or a LIST of such pairs: several prior
        models of one kind (the reference's GeneralBonds / GeneralAngles exist so that a SumOut can hold more than one
        bond or angle model) are condensed into one term list the way condense_all_priors_for_simulation does it
        (simulation/specialize_prior.py:159-196): mappings concatenated, per-term parameters looked up model by model
        and concatenated, Fourier series padded with zero amplitudes to the longest expansion.
        types [N], mol_ptr [B+1] on the CPU."""
        n_beads = int(mol_ptr[-1])
        terms = {}
        for kind, members in spec.items():
            if isinstance(members, tuple):
                members = [members]
            assert kind in KIND and members
            looked = []
            for tables, mp in members:
                assert mp.shape[0] == ORDER[kind] and mp.dtype == torch.int64
                looked.append(_lookup(kind, tables, types, mp).to(torch.float32))
            if kind == "dihedrals" and len(looked) > 1:   # rows: v_0, k1s[0..n), k2s[0..n) -> pad n to the largest
                n_max = max((q.shape[0] - 1) // 2 for q in looked)
                padded = []
                for q in looked:
                    n = (q.shape[0] - 1) // 2
                    z = torch.zeros(n_max - n, q.shape[1])
                    padded.append(torch.cat([q[:1 + n], z, q[1 + n:], z]))
                looked = padded
            if kind == "quartic_angles" and len(looked) > 1:   # rows: v_0, k_1..k_n -> zero coefficients up to the largest n
                rows = max(q.shape[0] for q in looked)
                looked = [torch.cat([q, torch.zeros(rows - q.shape[0], q.shape[1])]) for q in looked]
            params = torch.cat(looked, dim=1)
            mapping = torch.cat([mp for _, mp in members], dim=1)
            order, n_terms = mapping.shape
            assert n_terms < (1 << 29)
            # CSR by bead: entries (term << 2 | slot) sorted by (bead, term, slot)
            ent = (torch.arange(n_terms).view(1, -1) * 4 + torch.arange(order).view(-1, 1)).flatten()
            bead = mapping.flatten()
            perm = torch.argsort(bead * (4 * max(n_terms, 1)) + ent)
            bead_ptr = torch.zeros(n_beads + 1, dtype=torch.int64)
            bead_ptr[1:] = torch.cumsum(torch.bincount(bead, minlength=n_beads), 0)
            ents = ent[perm]
            pair = None
            if order == 2:  # partner-list form of the same CSR (used for long pair lists)
                t, slot = ents >> 2, ents & 3
                mol_start = mol_ptr[torch.bucketize(mapping[0][t], mol_ptr[1:], right=True)]
                other = mapping[1 - slot, t] - mol_start
                word = (other | ((slot == 0).to(torch.int64) << 31)).to(torch.int64)
                pair = (word.to(torch.uint32) if hasattr(torch, "uint32") else word, params[:, t].contiguous())
            terms[kind] = (mapping.contiguous(), params.contiguous(), bead_ptr, ents.to(torch.int32).contiguous(), pair)
        return cls(terms, int((mol_ptr[1:] - mol_ptr[:-1]).max()))

    @classmethod
    def from_state_dict(cls, sd, topology: Dict[str, torch.Tensor], types, mol_ptr, kinds: Dict[str, str] = None):
        """Row N4: the condensed priors of a reference-format ``SumOut`` container (what
        ``condense_all_priors_for_simulation`` builds from the model, simulation/specialize_prior.py:76-207) from its
        ``state_dict`` -- a dict, an ``.npz`` or a weights-only ``torch.load`` file -- plus the index mappings
        ``topology[name]`` [order, M] (global bead indices of the collated batch: ``data.neighbor_list[name]
        ["index_mapping"]``), bead types [N] and mol_ptr [B+1] on the CPU.  Every prior in the file needs its mapping;
        several priors of one kind (two bond models under different names, ...) are condensed into one term list as the
        reference does (``from_tables``).  ``kinds``: per prior name, which of two classes with identical buffers the
        file holds (``raw_angles`` for HarmonicAnglesRaw, ``shifted_impropers`` for ShiftedPeriodicHarmonicImpropers):
        see ``prior_tables_from_state_dict``."""
        if isinstance(sd, str):
            from .schnet import _state_dict_from_file

            sd = _state_dict_from_file(sd)
        found = prior_tables_from_state_dict(sd, kinds)
        spec = {}
        for name, (kind, tables) in found.items():
            if name not in topology:
                raise KeyError(f"prior {name!r} ({kind}) is in the state_dict but topology has no index mapping for it")
            spec.setdefault(kind, []).append((tables, topology[name].to(torch.int64)))
        return cls.from_tables(spec, types, mol_ptr)

    def to(self, device):
        mv = lambda t: None if t is None else (tuple(u.to(device) for u in t) if isinstance(t, tuple) else t.to(device))  # noqa: E731
        self.terms = {k: tuple(mv(t) for t in v) for k, v in self.terms.items()}
        self.status = torch.zeros(4, dtype=torch.int32, device=device)
        return self

    def add_energy_forces(self, pos, mol_ptr, energy, forces):
        """energy[B] += U_prior per molecule, forces[N,3] += -dU/dpos (in place, current stream)."""
        _ffi.require_gpu(pos, mol_ptr, energy, forces)
        assert pos.dtype == torch.float32 and pos.is_contiguous() and forces.is_contiguous()
        n_mol = mol_ptr.numel() - 1
        for kind, (mapping, params, bead_ptr, entries, pair) in self.terms.items():
            # partner-list form for long pair lists (> 16 entries per bead on average)
            if pair is not None and entries.numel() > 16 * pos.shape[0]:
                call("fmd_prior_pairs_energy_forces", ptr(pos), ptr(mol_ptr), n_mol, self.max_beads, KIND[kind],
                     ptr(bead_ptr), ptr(pair[0]), ptr(pair[1]), entries.numel(), ptr(energy), ptr(forces),
                     ptr(self.status), stream())
                continue
            call("fmd_prior_energy_forces", ptr(pos), ptr(mol_ptr), n_mol, self.max_beads, KIND[kind], ptr(mapping),
                 ptr(params), params.shape[0], ptr(bead_ptr), ptr(entries), mapping.shape[1], ptr(energy),
                 ptr(forces), ptr(self.status), stream())
        if not self._checked:   # first evaluation: one sync to catch a wrong bead bound / foreign term early
            self.check_status()
        return energy, forces

    def check_status(self):
        """status[3] (device): bit 0 = a term reaches outside its molecule, bit 1 = a molecule has more beads than
        ``max_beads`` (the kernels skipped it).  One sync; raises when either is set."""
        self._checked = True
        flag = int(self.status[3])
        if flag:
            raise RuntimeError(f"StaticPriors: device status {flag} (1 = term outside its molecule, 2 = molecule "
                               f"larger than max_beads={self.max_beads}): prior energies/forces are invalid")

    def energy_forces(self, pos, mol_ptr):
        energy = torch.zeros(mol_ptr.numel() - 1, dtype=torch.float32, device=pos.device)
        forces = torch.zeros_like(pos)
        return self.add_energy_forces(pos, mol_ptr, energy, forces)

    def energy_torch(self, pos, mol_ptr):
        """The same energies as plain torch ops (differentiable; --disable_optim / CPU path)."""
        n_mol = mol_ptr.numel() - 1
        mol_of_bead = torch.repeat_interleave(torch.arange(n_mol, device=pos.device), mol_ptr[1:] - mol_ptr[:-1])
        energy = torch.zeros(n_mol, dtype=pos.dtype, device=pos.device)
        for kind, (mapping, params, *_) in self.terms.items():
            p = params.to(pos.dtype)
            if kind == "bonds":
                e = HarmonicBonds.compute(compute_distances(pos, mapping), p[1], p[0])
            elif kind == "angles":
                e = HarmonicAngles.compute(compute_angles_cos(pos, mapping), p[1], p[0])
            elif kind == "repulsion":
                e = Repulsion.compute(compute_distances(pos, mapping), p[0])
            elif kind == "impropers":
                e = HarmonicImpropers.compute(compute_torsions(pos, mapping), p[1], p[0])
            elif kind == "raw_angles":
                e = HarmonicAnglesRaw.compute(compute_angles_raw(pos, mapping), p[1], p[0])
            elif kind == "shifted_impropers":
                e = ShiftedPeriodicHarmonicImpropers.compute(ShiftedPeriodicHarmonicImpropers.compute_features(pos, mapping),
                                                             p[1], p[0])
            elif kind == "quartic_angles":
                e = QuarticAngles.compute(compute_angles_cos(pos, mapping), p[1:], p[0])
            elif kind == "restricted_angles":
                e = RestrictedQuartic.compute(compute_angles_raw(pos, mapping), *p)
            else:
                n = (p.shape[0] - 1) // 2
                e = Dihedral.compute(compute_torsions(pos, mapping), p[0], p[1:1 + n].t(), p[1 + n:].t())
            energy = energy.index_add(0, mol_of_bead[mapping[0]], e)
        return energy


def sum_out(force_fns):
    """force_fn(pos) -> (energy[B], forces[N,3]) summing several models, the role of the reference's SumOut
    (models/gradients.py:71-152).  ``force_fns``: first a callable returning fresh (energy, forces) -- e.g. the
    fused SchNet engine -- then objects with ``add_energy_forces(pos, mol_ptr, energy, forces)`` such as
    StaticPriors, given as (obj, mol_ptr) pairs."""
    first, rest = force_fns[0], force_fns[1:]

    def fn(pos):
        energy, forces = first(pos)
        for obj, mol_ptr in rest:
            obj.add_energy_forces(pos, mol_ptr, energy, forces)
        return energy, forces

    def check_status():
        for f in [first] + [obj for obj, _ in rest]:
            chk = getattr(f, "check_status", None) or getattr(getattr(f, "__self__", None), "check_status", None)
            if chk is not None:
                chk()

    fn.check_status = check_status
    return fn
