"""CPU oracle for the CGSchNet energy+force path (TEST INFRASTRUCTURE, not product code).

This file restates, in plain torch on the CPU, the arithmetic of the reference's
``--disable_optim`` PyTorch path.  It is imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` -- never by the
product package ``flashmd_amd`` -- and it is never the thing that is measured or shipped.

Pinning: ``oracle/make_golden.py`` ran the reference's own ``StandardSchNet`` +
``GradientsOut`` (imported from /root/reference with in-process stand-ins for the missing
``nvtx`` / ``torch_geometric`` / ``torch_cluster`` wheels) on the seeded inputs stored under
``tests/golden/`` and wrote its energies/forces there; ``tests/test_oracle.py`` checks this
restatement against those vectors and against the closed-form cutoff end points the
reference tests hold (reference ``tests/models/test_cutoff.py:27-42``).
The W16A16 variant (``w16a16=True``) follows the rounding points listed in SURVEY.md row A9 and is
pinned to the reference's own Triton kernels: ``oracle/make_golden_triton.py`` executed them under
``TRITON_INTERPRET=1`` (stage vectors ``tests/golden/triton_stages_1x37.npz``: K6 / K7 bit-identical, K4 / K5
within one fp16 ulp on < 0.3 % of the elements) and the reference's W16A16 model end to end
(``tests/golden/case_w16a16_ref.npz``); ``tests/test_oracle.py`` checks both.

Every function cites the reference file:line it follows (paths relative to
/root/reference/src/flashmd).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------


def xavier_uniform(gen: torch.Generator, fan_out: int, fan_in: int) -> torch.Tensor:
    """Xavier-uniform [out,in] weight, the distribution of models/_module_init.py:24-26."""
    bound = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(fan_out, fan_in, generator=gen) * 2.0 - 1.0) * bound


def make_params(
    hidden: int = 128,
    filters: int = 128,
    num_rbf: int = 50,
    num_interactions: int = 3,
    out_widths: Tuple[int, ...] = (128, 64),
    embedding_size: int = 100,
    cutoff: float = 5.0,
    seed: int = 0,
    bias_scale: float = 0.0,
) -> Dict:
    """Random StandardSchNet-shaped weights (models/schnet.py:778-841): N(0,1) embedding,
    xavier-uniform linears, zero biases (``bias_scale`` > 0 draws non-zero biases so tests
    exercise the bias terms).  Weights use torch.nn.Linear's [out, in] layout."""
    g = torch.Generator().manual_seed(seed)

    def bias(n):
        return (torch.rand(n, generator=g) * 2 - 1) * bias_scale

    p = {
        "hidden": hidden,
        "filters": filters,
        "num_rbf": num_rbf,
        "cutoff": float(cutoff),
        "emb": torch.randn(embedding_size, hidden, generator=g),
        "blocks": [],
        "out": [],
    }
    for _ in range(num_interactions):
        p["blocks"].append(
            {
                "lin1_w": xavier_uniform(g, filters, hidden),  # CFConv.lin1, no bias (schnet.py:584)
                "f0_w": xavier_uniform(g, filters, num_rbf),  # filter MLP layer 0 (schnet.py:813-817)
                "f0_b": bias(filters),
                "f1_w": xavier_uniform(g, filters, filters),  # filter MLP layer 1, last_bias=False
                "lin2_w": xavier_uniform(g, hidden, filters),  # CFConv.lin2 (schnet.py:585)
                "lin2_b": bias(hidden),
                "lin_w": xavier_uniform(g, hidden, hidden),  # InteractionBlock.lin (schnet.py:479)
                "lin_b": bias(hidden),
            }
        )
    widths = [hidden] + list(out_widths) + [1]  # schnet.py:829-834
    for k, (w_in, w_out) in enumerate(zip(widths[:-1], widths[1:])):
        last = k == len(widths) - 2
        p["out"].append(
            {"w": xavier_uniform(g, w_out, w_in), "b": None if last else bias(w_out)}
        )
    return p


def params_to(p: Dict, dtype: torch.dtype) -> Dict:
    def cv(v):
        if isinstance(v, torch.Tensor):
            return v.to(dtype) if v.is_floating_point() else v
        if isinstance(v, dict):
            return {k: cv(x) for k, x in v.items()}
        if isinstance(v, list):
            return [cv(x) for x in v]
        return v

    return cv(p)


# --------------------------------------------------------------------------------------
# synthetic 1ENH-shaped inputs
# --------------------------------------------------------------------------------------


def synth_molecule(n_beads: int, cutoff: float, mol_index: int) -> torch.Tensor:
    """n_beads points uniform in a ball of radius 1.44*rc * (n/269)^(1/3), rejecting points
    closer than 0.15*rc to an earlier one; seed 1000+mol_index.  Gives ~55 neighbours/bead at
    n=269, rc=5 -- the density of the reference's 1ENH batch (kernels/csr_kernels.py:127,519)."""
    g = torch.Generator().manual_seed(1000 + mol_index)
    rs = 1.44 * cutoff * (n_beads / 269.0) ** (1.0 / 3.0)
    dmin2 = (0.15 * cutoff) ** 2
    pts: List[torch.Tensor] = []
    while len(pts) < n_beads:
        c = (torch.rand(3, generator=g) * 2 - 1) * rs
        if float((c * c).sum()) > rs * rs:
            continue
        if pts:
            q = torch.stack(pts)
            if float(((q - c) ** 2).sum(1).min()) < dmin2:
                continue
        pts.append(c)
    return torch.stack(pts).float()


def synth_batch(n_mol: int, n_beads: int = 269, cutoff: float = 5.0, first_mol: int = 0):
    """(pos[N,3] f32, atom_types[N] i64, mol_ptr[B+1] i64).  One type sequence (seed 7,
    uniform in [1,25)) shared by all molecules, as simulation/base.py:937-946 requires."""
    pos = torch.cat([synth_molecule(n_beads, cutoff, first_mol + m) for m in range(n_mol)])
    g = torch.Generator().manual_seed(7)
    types1 = torch.randint(1, 25, (n_beads,), generator=g)
    types = types1.repeat(n_mol)
    ptr = torch.arange(n_mol + 1, dtype=torch.int64) * n_beads
    return pos, types, ptr


# --------------------------------------------------------------------------------------
# neighbour list (A1)
# --------------------------------------------------------------------------------------


def radius_graph(pos: torch.Tensor, mol_ptr: torch.Tensor, cutoff: float):
    """All ordered pairs (i, j), i != j, same molecule, |r_j - r_i| < cutoff (strict), rows
    sorted by centre i then neighbour j; returns (edge_src=i, edge_dst=j) int64.

    Follows neighbor_list/torch_impl.py:216-224 (radius_graph, flow="target_to_source",
    loop=False, no truncation below 1000 neighbours).  torch_cluster itself is not in the
    container: strictness and order are fixed here by definition.  The test is
    made on the squared distance ((dx*dx + dy*dy) + dz*dz, each op rounded in the input
    dtype) against cutoff*cutoff rounded to the input dtype -- the GPU kernel uses the same
    unfused sequence so the edge set is bit-identical."""
    srcs, dsts = [], []
    rc2 = torch.tensor(cutoff, dtype=pos.dtype) * torch.tensor(cutoff, dtype=pos.dtype)
    for m in range(mol_ptr.numel() - 1):
        a, b = int(mol_ptr[m]), int(mol_ptr[m + 1])
        p = pos[a:b]
        dx = p[None, :, 0] - p[:, None, 0]  # [i, j] = x_j - x_i
        dy = p[None, :, 1] - p[:, None, 1]
        dz = p[None, :, 2] - p[:, None, 2]
        d2 = (dx * dx + dy * dy) + dz * dz
        mask = d2 < rc2
        mask.fill_diagonal_(False)
        ij = mask.nonzero()  # row-major: sorted by i then j
        srcs.append(ij[:, 0] + a)
        dsts.append(ij[:, 1] + a)
    if not srcs:
        z = torch.zeros(0, dtype=torch.int64)
        return z, z.clone()
    return torch.cat(srcs), torch.cat(dsts)


# --------------------------------------------------------------------------------------
# per-edge geometry (A2)
# --------------------------------------------------------------------------------------


def compute_distances(pos, edge_src, edge_dst):
    """geometry/internal_coordinates.py:96-101: |pos[dst] - pos[src]|_2."""
    dr = pos[edge_dst] - pos[edge_src]
    return dr.norm(p=2, dim=1)


def cosine_cutoff(d: torch.Tensor, cutoff: float) -> torch.Tensor:
    """models/cutoff.py:137-145 (cutoff_lower == 0 branch)."""
    c = 0.5 * (torch.cos(d * math.pi / cutoff) + 1.0)
    return c * (d < cutoff).to(d.dtype)


def rbf_centers(num_rbf: int, cutoff: float, dtype=torch.float32):
    """models/radial_basis/gaussian.py:71-75: linspace centres and gamma=-0.5/spacing^2
    (computed in fp32 like the reference buffer, then cast)."""
    offset = torch.linspace(0.0, cutoff, num_rbf)
    coeff = -0.5 / (offset[1] - offset[0]) ** 2
    return offset.to(dtype), float(coeff)


def gaussian_rbf(d: torch.Tensor, centers: torch.Tensor, gamma: float, cutoff: float):
    """models/radial_basis/gaussian.py:98-101: exp(gamma (d-c_k)^2) * C(d)."""
    dd = d.unsqueeze(-1)
    return torch.exp(gamma * torch.pow(dd - centers, 2)) * cosine_cutoff(dd, cutoff)


# --------------------------------------------------------------------------------------
# CSR (A3) -- integer oracle
# --------------------------------------------------------------------------------------


def build_csr(keys: torch.Tensor, num_nodes: int):
    """kernels/csr_kernels.py:88-169 / :229-294: ptr = [0, cumsum(bincount(keys))] int64 and
    perm[k] = original edge id, edges of one key contiguous.  The reference fills segments
    through atomic cursors, so the order inside a segment is whatever the hardware produced
    (SURVEY.md F6); the canonical member of that output set -- ascending edge id inside each
    segment, i.e. the stable counting sort -- is the oracle."""
    counts = torch.bincount(keys, minlength=num_nodes)
    ptr = torch.zeros(num_nodes + 1, dtype=torch.int64)
    ptr[1:] = counts.cumsum(0)
    perm = torch.sort(keys, stable=True).indices.to(torch.int64)
    return ptr, perm


# --------------------------------------------------------------------------------------
# fp16 rounding helpers for the W16A16 variant (A9)
# --------------------------------------------------------------------------------------


def _h(x: torch.Tensor) -> torch.Tensor:
    """round to fp16 and return in the working dtype (a 'storage rounding point')."""
    return x.to(torch.float16).to(x.dtype)


def _tanh_w16(x: torch.Tensor) -> torch.Tensor:
    """kernels/cfconv_kernels.py:449-454: (e^{2x}-1)/(e^{2x}+1) with x clamped to +-10."""
    xc = x.clamp(-10.0, 10.0)
    e = torch.exp(2.0 * xc)
    return (e - 1.0) / (e + 1.0)


class _RoundGradFP16(torch.autograd.Function):
    """identity forward; rounds the incoming gradient to fp16 (the reference stores
    grad_filter_out / grad of fp16 activations as fp16: cfconv_kernels.py:252-253,1198)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _h(g)


class _RoundFP16STE(torch.autograd.Function):
    """fp16 storage rounding in forward, pass-through gradient."""

    @staticmethod
    def forward(ctx, x):
        return _h(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _LinTanhW16(torch.autograd.Function):
    """K4 forward (cfconv_kernels.py:657-720) and K6 backward (:973-1035):
    y = f16(tanh(f16(x) @ w_h + b_h)); gx = f16(g * (1 - y^2)) @ w_h^T (fp32 accumulate)."""

    @staticmethod
    def forward(ctx, x, w_h, b_h):  # w_h [K,N] already fp16-rounded values
        z = _h(x) @ w_h
        if b_h is not None:
            z = z + b_h
        y = _h(_tanh_w16(z))
        ctx.save_for_backward(y, w_h)
        return y

    @staticmethod
    def backward(ctx, g):
        y, w_h = ctx.saved_tensors
        gz = _h(_h(g) * (1.0 - y * y))
        return gz @ w_h.t(), None, None


class _LinW16(torch.autograd.Function):
    """K5 forward (cfconv_kernels.py:780-832, 908-952) and K7 backward (:1174-1208):
    y = x_h @ w_h (stored f16 or f32); gx = f16( f16(g) @ w_h^T )."""

    @staticmethod
    def forward(ctx, x, w_h, out_half: bool):
        y = _h(x) @ w_h
        ctx.save_for_backward(w_h)
        return _h(y) if out_half else y

    @staticmethod
    def backward(ctx, g):
        (w_h,) = ctx.saved_tensors
        return _h(_h(g) @ w_h.t()), None, None


# --------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------


def filter_network(rbf, blk, w16a16: bool = False):
    """models/mlp.py:41-57 as built at schnet.py:813-817: tanh(rbf W0^T + b0) W1^T.
    W16A16 twin: models/gptq.py:92-130 (weights = Linear.weight.t().half(), :178-184)."""
    if not w16a16:
        h = torch.tanh(rbf @ blk["f0_w"].t() + blk["f0_b"])
        return h @ blk["f1_w"].t()
    h = _LinTanhW16.apply(rbf, _h(blk["f0_w"].t()), _h(blk["f0_b"]))
    f = _LinW16.apply(h, _h(blk["f1_w"].t()), True)
    return _RoundGradFP16.apply(f)  # grad_filter_out is produced in fp16 (cfconv_kernels.py:252)


def cfconv(x, edge_src, edge_dst, d, rbf, blk, cutoff, w16a16=False, stages=None):
    """models/schnet.py:603-720, PyTorch branch :706-719:
    x1 = lin1(x); W = filter(rbf) * C(d); agg[dst] += x1[src] * W; out = lin2(agg)."""
    x1 = x @ blk["lin1_w"].t()
    W = filter_network(rbf, blk, w16a16) * cosine_cutoff(d, cutoff).view(-1, 1)
    msg = x1.index_select(0, edge_src) * W  # message(): x_j * W, x_j = x[edge_index[0]]
    agg = torch.zeros_like(x1).index_add_(0, edge_dst, msg)  # aggregate at edge_index[1]
    if stages is not None:
        stages.append({"x1": x1.detach(), "agg": agg.detach()})
    return agg @ blk["lin2_w"].t() + blk["lin2_b"]


def interaction_block(x, edge_src, edge_dst, d, rbf, blk, cutoff, w16a16=False, stages=None):
    """models/schnet.py:497-548: lin(tanh(cfconv(x)))."""
    c = cfconv(x, edge_src, edge_dst, d, rbf, blk, cutoff, w16a16, stages)
    return torch.tanh(c) @ blk["lin_w"].t() + blk["lin_b"]


def output_network(x, p, w16a16=False):
    """models/mlp.py:41-57 as built at schnet.py:829-834 (tanh hidden layers, last layer no
    bias).  W16A16 twin: models/gptq.py:266-306 (K4, K4, K5 -> fp32)."""
    layers = p["out"]
    if not w16a16:
        for lay in layers[:-1]:
            x = torch.tanh(x @ lay["w"].t() + lay["b"])
        return x @ layers[-1]["w"].t()
    for lay in layers[:-1]:
        x = _LinTanhW16.apply(x, _h(lay["w"].t()), _h(lay["b"]))
    return _LinW16.apply(x, _h(layers[-1]["w"].t()), False)


def schnet_energy(
    p: Dict,
    pos: torch.Tensor,
    atom_types: torch.Tensor,
    mol_ptr: torch.Tensor,
    edge_src: Optional[torch.Tensor] = None,
    edge_dst: Optional[torch.Tensor] = None,
    w16a16: bool = False,
    stages: Optional[list] = None,
) -> torch.Tensor:
    """models/schnet.py:178-369 with every MLCG_USE_* gate off: energy[B]."""
    rc = p["cutoff"]
    if edge_src is None:
        edge_src, edge_dst = radius_graph(pos.detach(), mol_ptr, rc)
    x = p["emb"][atom_types]  # schnet.py:203
    d = compute_distances(pos, edge_src, edge_dst)  # schnet.py:287
    centers, gamma = rbf_centers(p["num_rbf"], rc, pos.dtype)
    rbf = gaussian_rbf(d, centers, gamma, rc)  # schnet.py:299
    if stages is not None:
        stages.append({"d": d.detach(), "rbf": rbf.detach()})
    for blk in p["blocks"]:  # schnet.py:336-341
        x = x + interaction_block(x, edge_src, edge_dst, d, rbf, blk, rc, w16a16, stages)
    e_atom = output_network(x, p, w16a16)  # schnet.py:346
    n_mol = mol_ptr.numel() - 1
    batch = torch.repeat_interleave(torch.arange(n_mol), mol_ptr[1:] - mol_ptr[:-1])
    energy = torch.zeros(n_mol, dtype=e_atom.dtype).index_add_(0, batch, e_atom.flatten())
    return energy  # schnet.py:356-357


def schnet_energy_forces(p, pos, atom_types, mol_ptr, edge_src=None, edge_dst=None,
                         w16a16=False, stages=None):
    """models/gradients.py:249-280: forces = -d(sum_b E_b)/d pos."""
    pos = pos.detach().clone().requires_grad_(True)
    e = schnet_energy(p, pos, atom_types, mol_ptr, edge_src, edge_dst, w16a16, stages)
    (g,) = torch.autograd.grad(e.sum(), pos)
    return e.detach(), -g


# --------------------------------------------------------------------------------------
# integrator and replica exchange oracles (rows "next" N1 and (e))
# --------------------------------------------------------------------------------------


def langevin_baoab_pre(x, v, f, masses, noise, dt, friction, beta):
    """simulation/langevin.py:136-157: B, A, O, A with the noise tensor given
    (vscale=exp(-friction dt), noisescale=sqrt(1-vscale^2), :76-77; noise scale
    sqrt(1/(beta m)), :211-215)."""
    vscale = math.exp(-friction * dt)
    noisescale = math.sqrt(1.0 - vscale * vscale)
    v = v + 0.5 * dt * f / masses[:, None]
    x = x + v * dt * 0.5
    bm = torch.sqrt(1.0 / (beta * masses))[:, None]
    v = v * vscale + noisescale * (bm * noise)
    x = x + v * dt * 0.5
    return x, v


def langevin_baoab_post(v, f, masses, dt):
    """simulation/langevin.py:169: second B half-step."""
    return v + 0.5 * dt * f / masses[:, None]


def velocity_verlet_pre(x, v, f, masses, dt):
    """simulation/velocity_verlet.py:38-41: half kick, full drift."""
    v = v + 0.5 * dt * f / masses[:, None]
    return x + dt * v, v


def pt_exchange_decisions(u_a, u_b, beta_a, beta_b, rand_u):
    """simulation/parallel_tempering.py:368-413: accept when rand < exp((U_a-U_b)(beta_a-beta_b))."""
    p = torch.exp((u_a - u_b) * (beta_a - beta_b))
    return rand_u < p


# --------------------------------------------------------------------------------------
# fixture I/O
# --------------------------------------------------------------------------------------


def params_from_npz(path: str) -> Dict:
    """Read the flat weight fixture written by oracle/make_golden.py (numpy, no pickle)."""
    import numpy as np

    z = np.load(path)
    hidden, filters, num_rbf, n_blocks, n_out = [int(v) for v in z["meta"]]
    p = {"hidden": hidden, "filters": filters, "num_rbf": num_rbf,
         "cutoff": float(z["cutoff"][0]), "emb": torch.from_numpy(z["emb"]),
         "blocks": [], "out": []}
    keys = ("lin1_w", "f0_w", "f0_b", "f1_w", "lin2_w", "lin2_b", "lin_w", "lin_b")
    for l in range(n_blocks):
        p["blocks"].append({k: torch.from_numpy(z[f"b{l}_{k}"]) for k in keys})
    for l in range(n_out):
        b = torch.from_numpy(z[f"o{l}_b"]) if f"o{l}_b" in z.files else None
        p["out"].append({"w": torch.from_numpy(z[f"o{l}_w"]), "b": b})
    return p
