"""CPU restatement of the reference's classical prior terms (row N2 of SURVEY.md section 8f).  TEST
INFRASTRUCTURE ONLY: imported by tests/, never by the product package.

Follows, term by term:
  features   geometry/internal_coordinates.py:72-101 (distances), :152-178 (cos of angles), :181-223 (torsions)
  bonds      prior/harmonic.py:91-123, 200-225      U = k (d - x0)^2
  angles     prior/harmonic.py:228-262              U = k (cos(theta) - x0)^2
  repulsion  prior/repulsion.py:80-122              U = (sigma / d)^6
  dihedrals  prior/fourier_series.py:97-141,154-192 U = v0 + sum_n k1_n sin(n phi) + k2_n cos(n phi)
  impropers  prior/harmonic.py:301-322              U = k (phi - x0)^2            (Harmonic.compute on the torsion)
  quartic_angles     prior/polynomial.py:127-186    U = v0 + sum_n k_n cos(theta)^n
  restricted_angles  prior/restricted_bending.py:200-238  U = a c^4 + b c^3 + c c^2 + d c + k / sin^2 + v0 on the raw
                     angle theta = atan2(|r1 x r2|, r1 . r2) (geometry/internal_coordinates.py:104-135)
Per-term parameters are looked up by bead type exactly as `data2parameters` does (table[types of the term]);
energies are summed per molecule (`scatter(..., mapping_batch)`), forces are -dU/dpos by autograd.
Pinned against the reference's own classes by oracle/make_golden_priors.py -> tests/golden/case_priors.npz and
case_priors_more.npz (the last three kinds).
"""
import torch


def distances(pos, mapping):
    return (pos[mapping[1]] - pos[mapping[0]]).norm(p=2, dim=1)


def angles_cos(pos, mapping):
    dr1 = pos[mapping[0]] - pos[mapping[1]]
    dr2 = pos[mapping[2]] - pos[mapping[1]]
    return (dr1 * dr2).sum(dim=1) / (dr1.norm(p=2, dim=1) * dr2.norm(p=2, dim=1))


def angles_raw(pos, mapping):
    dr1 = pos[mapping[0]] - pos[mapping[1]]
    dr2 = pos[mapping[2]] - pos[mapping[1]]
    return torch.atan2(torch.cross(dr1, dr2, dim=1).norm(p=2, dim=1), (dr1 * dr2).sum(dim=1))


def torsions(pos, mapping):
    n = torch.nn.functional.normalize
    dr1 = n(pos[mapping[1]] - pos[mapping[0]], dim=1)
    dr2 = n(pos[mapping[2]] - pos[mapping[1]], dim=1)
    dr3 = n(pos[mapping[3]] - pos[mapping[2]], dim=1)
    n1 = torch.cross(dr1, dr2, dim=1)
    n2 = torch.cross(dr2, dr3, dim=1)
    m1 = torch.cross(n1, dr2, dim=1)
    return torch.atan2(-(m1 * n2).sum(-1), (n1 * n2).sum(-1))


def term_energies(kind, pos, mapping, params):
    """params: [P, M] per-term parameters (the Static* form, simulation/specialize_prior.py:76-207)."""
    if kind == "bonds":
        return params[0] * (distances(pos, mapping) - params[1]) ** 2          # k, x0
    if kind == "angles":
        return params[0] * (angles_cos(pos, mapping) - params[1]) ** 2         # k, x0
    if kind == "repulsion":
        rr = (params[0] / distances(pos, mapping)) ** 2                        # sigma
        return rr * rr * rr
    if kind == "dihedrals":
        n = (params.shape[0] - 1) // 2                                         # v0, k1[1..n], k2[1..n]
        phi = torsions(pos, mapping)
        deg = torch.arange(1, n + 1, dtype=pos.dtype).view(-1, 1)
        ang = phi.view(1, -1) * deg
        return params[0] + (params[1:1 + n] * torch.sin(ang) + params[1 + n:] * torch.cos(ang)).sum(0)
    if kind == "impropers":
        return params[0] * (torsions(pos, mapping) - params[1]) ** 2           # k, x0
    if kind == "raw_angles":        # prior/harmonic.py:264-298 (HarmonicAnglesRaw): k, x0 on the raw angle
        return params[0] * (angles_raw(pos, mapping) - params[1]) ** 2
    if kind == "shifted_impropers":  # prior/harmonic.py:325-400: torsions < 0 shifted by 2 pi, then pi subtracted
        phi = torsions(pos, mapping)
        pi32 = float(torch.tensor(torch.pi, dtype=torch.float32))   # the reference's torch_pi is a float32 tensor (harmonic.py:19)
        phi = torch.where(phi < 0, phi + 2 * pi32, phi) - pi32
        return params[0] * (phi - params[1]) ** 2
    if kind == "quartic_angles":                                               # v0, k_1 .. k_n
        x = angles_cos(pos, mapping)
        v, xp = params[1] * x, x
        for q in range(2, params.shape[0]):
            xp = xp * x
            v = v + params[q] * xp
        return v + params[0]
    if kind == "restricted_angles":                                            # a, b, c, d, k, v0
        th = angles_raw(pos, mapping)
        c, s = torch.cos(th), torch.sin(th)
        return params[0] * c ** 4 + params[1] * c ** 3 + params[2] * c ** 2 + params[3] * c + params[4] / s ** 2 + params[5]
    raise ValueError(kind)


def lookup(kind, tables, types, mapping):
    """per-term parameters from per-type tables, as data2parameters does."""
    idx = tuple(types[mapping[i]] for i in range(mapping.shape[0]))
    if kind in ("bonds", "angles", "impropers", "raw_angles", "shifted_impropers"):
        return torch.stack([tables["k"][idx], tables["x_0"][idx]])
    if kind == "repulsion":
        return tables["sigma"][idx].unsqueeze(0)
    if kind == "quartic_angles":
        return torch.cat([tables["v_0"][idx].unsqueeze(0), torch.stack([tables["ks"][i][idx] for i in range(tables["ks"].shape[0])])])
    if kind == "restricted_angles":
        return torch.stack([tables[n][idx] for n in ("a", "b", "c", "d", "k", "v_0")])
    n = tables["k1s"].shape[0]
    return torch.cat([tables["v_0"][idx].unsqueeze(0), torch.stack([tables["k1s"][i][idx] for i in range(n)]),
                      torch.stack([tables["k2s"][i][idx] for i in range(n)])])


def prior_energy_forces(pos, terms, mol_of_bead, n_mol):
    """terms: list of (kind, mapping [order, M], params [P, M]).  Returns energy [n_mol] and forces [N, 3]."""
    pos = pos.detach().clone().requires_grad_(True)
    energy = torch.zeros(n_mol, dtype=pos.dtype)
    for kind, mapping, params in terms:
        e = term_energies(kind, pos, mapping, params.to(pos.dtype))
        energy = energy.index_add(0, mol_of_bead[mapping[0]], e)
    (g,) = torch.autograd.grad(energy.sum(), pos)
    return energy.detach(), -g
