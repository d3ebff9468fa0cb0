"""Host-side mirror of the reference's SchNet model for the force path (reference
src/flashmd/models/schnet.py, mlp.py, cutoff.py, radial_basis/gaussian.py, gradients.py).

Module and parameter names follow the reference so a reference ``state_dict`` loads unchanged
(``embedding_layer``, ``interaction_blocks.{l}.conv.{lin1,lin2,filter_network.layers.{0,2}}``,
``interaction_blocks.{l}.lin``, ``rbf_layer.{offset,coeff}``, ``output_network.layers.*``).
torch_geometric / torch_cluster are not used: the graph is built by the HIP neighbour kernel
(or, with every gate off, by plain torch) and message passing is gather + index_add.

Gates (read at import, same names and meaning as reference models/schnet.py:52-56):
  MLCG_USE_TRITON_MESSAGE_PASSING, MLCG_USE_FUSED_RBF, MLCG_USE_FUSED_TANH_LINEAR, MLCG_USE_CSR,
  MLCG_USE_SRC_CSR_GRAD_X      "1" (default) -> hand-written HIP operators, "0" -> PyTorch ops.
  MLCG_USE_FUSED_ENGINE        "1" (default) -> the whole energy+force evaluation runs in the
                               fused HIP engine when the model qualifies (flashmd_amd.engine).
With any HIP gate on, tensors must live on the GPU: there is no silent CPU fallback
(``--disable_optim`` / all gates "0" selects the PyTorch path, which is what runs on CPU).
"""
from __future__ import annotations

import math
import os
from typing import Final, List, Optional

import torch

from . import kernels as K

USE_TRITON_MESSAGE_PASSING = os.environ.get("MLCG_USE_TRITON_MESSAGE_PASSING", "1") == "1"
USE_FUSED_RBF = os.environ.get("MLCG_USE_FUSED_RBF", "1")
USE_FUSED_TANH_LINEAR = os.environ.get("MLCG_USE_FUSED_TANH_LINEAR", "1") == "1"
USE_CSR = os.environ.get("MLCG_USE_CSR", "1") == "1"
USE_SRC_CSR_GRAD_X = os.environ.get("MLCG_USE_SRC_CSR_GRAD_X", "1") == "1"
USE_FUSED_ENGINE = os.environ.get("MLCG_USE_FUSED_ENGINE", "1") == "1"

ENERGY_KEY: Final[str] = "energy"  # reference data/_keys.py:42-43
FORCE_KEY: Final[str] = "forces"


def any_hip_gate_on() -> bool:
    return (USE_TRITON_MESSAGE_PASSING or USE_FUSED_RBF == "1" or USE_FUSED_TANH_LINEAR or USE_CSR
            or USE_SRC_CSR_GRAD_X)


def _need_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: the HIP operator gates are on but the tensors are on the CPU. There is no CPU "
            "fallback for the optimized path; run with --disable_optim (all MLCG_USE_* = 0) for the "
            "PyTorch path.")


# --------------------------------------------------------------------------------------------------
# small modules (math spec of the path)
# --------------------------------------------------------------------------------------------------


def init_xavier_uniform(module: torch.nn.Module, zero_bias: bool = True) -> None:
    """reference models/_module_init.py:4-28"""
    if isinstance(module, torch.nn.Linear):
        torch.nn.init.xavier_uniform_(module.weight)
        if module.bias is not None and zero_bias:
            torch.nn.init.constant_(module.bias, 0.0)


class CosineCutoff(torch.nn.Module):
    """0.5 (cos(pi d / r_hi) + 1) [d < r_hi]  (reference models/cutoff.py:75-145)."""

    def __init__(self, cutoff_lower: float = 0.0, cutoff_upper: float = 5.0):
        super().__init__()
        self.cutoff_lower, self.cutoff_upper = cutoff_lower, cutoff_upper
        if cutoff_upper < cutoff_lower:
            raise ValueError(f"Upper cutoff {cutoff_upper} is less than lower cutoff {cutoff_lower}")

    def forward(self, distances: torch.Tensor) -> torch.Tensor:
        if self.cutoff_lower > 0:
            c = 0.5 * (torch.cos(math.pi * (2 * (distances - self.cutoff_lower)
                                            / (self.cutoff_upper - self.cutoff_lower) + 1.0)) + 1.0)
            c = c * (distances < self.cutoff_upper).to(distances.dtype)
            return c * (distances > self.cutoff_lower).to(distances.dtype)
        c = 0.5 * (torch.cos(distances * math.pi / self.cutoff_upper) + 1.0)
        return c * (distances < self.cutoff_upper).to(distances.dtype)


class GaussianBasis(torch.nn.Module):
    """exp(coeff (d - offset_k)^2) * cutoff(d)  (reference models/radial_basis/gaussian.py:9-102)."""

    def __init__(self, cutoff: CosineCutoff, num_rbf: int = 50):
        super().__init__()
        self.cutoff, self.num_rbf = cutoff, num_rbf
        offset = torch.linspace(cutoff.cutoff_lower, cutoff.cutoff_upper, num_rbf)
        self.register_buffer("coeff", -0.5 / (offset[1] - offset[0]) ** 2)
        self.register_buffer("offset", offset)

    def forward(self, dist: torch.Tensor) -> torch.Tensor:
        dist = dist.unsqueeze(-1)
        return torch.exp(self.coeff * torch.pow(dist - self.offset, 2)) * self.cutoff(dist)


class MLP(torch.nn.Module):
    """reference models/mlp.py:6-57"""

    def __init__(self, layer_widths: List[int], activation_func=None, last_bias: bool = True):
        super().__init__()
        activation_func = activation_func or torch.nn.Tanh()
        layers = []
        for w_in, w_out in zip(layer_widths[:-2], layer_widths[1:-1]):
            layers += [torch.nn.Linear(w_in, w_out, bias=True), activation_func]
        layers.append(torch.nn.Linear(layer_widths[-2], layer_widths[-1], bias=last_bias))
        self.layers = torch.nn.Sequential(*layers)
        for layer in self.layers:
            init_xavier_uniform(layer)

    def forward(self, x, data=None):
        return self.layers(x)


# --------------------------------------------------------------------------------------------------
# graph construction
# --------------------------------------------------------------------------------------------------


def radius_graph_torch(pos: torch.Tensor, ptr: torch.Tensor, cutoff: float) -> torch.Tensor:
    """PyTorch branch of the neighbour list (what torch_cluster.radius_graph does for the
    reference, neighbor_list/torch_impl.py:216-224): strict <, same molecule, no self pairs,
    rows [centre, neighbour] sorted by centre then neighbour."""
    out = []
    rc2 = torch.tensor(cutoff, dtype=pos.dtype, device=pos.device) ** 2
    bounds = ptr.tolist()
    for a, b in zip(bounds[:-1], bounds[1:]):
        p = pos[a:b]
        diff = p[None, :, :] - p[:, None, :]
        d2 = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
        mask = d2 < rc2
        mask.fill_diagonal_(False)
        out.append(mask.nonzero().t() + a)
    if not out:
        return torch.zeros((2, 0), dtype=torch.int64, device=pos.device)
    return torch.cat(out, dim=1)


def _mol_ptr(data) -> torch.Tensor:
    ptr = getattr(data, "ptr", None)
    if ptr is not None:
        return ptr
    counts = torch.bincount(data.batch)
    ptr = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=data.batch.device)
    ptr[1:] = counts.cumsum(0)
    return ptr


# --------------------------------------------------------------------------------------------------
# CFConv / InteractionBlock / SchNet
# --------------------------------------------------------------------------------------------------


class CFConv(torch.nn.Module):
    """reference models/schnet.py:551-740"""

    def __init__(self, filter_network, cutoff, in_channels=128, out_channels=128, num_filters=128):
        super().__init__()
        self.lin1 = torch.nn.Linear(in_channels, num_filters, bias=False)
        self.lin2 = torch.nn.Linear(num_filters, out_channels)
        self.filter_network, self.cutoff = filter_network, cutoff
        init_xavier_uniform(self.lin1)
        init_xavier_uniform(self.lin2)

    def forward(self, x, edge_index, edge_weight, edge_attr, csr_data: dict = None):
        x = self.lin1(x)
        if csr_data is not None and "dst_ptr" in csr_data:  # reference :648-676
            filter_out = self.filter_network(edge_attr)
            x = K.fused_csr_cfconv_autograd(
                x.contiguous(), filter_out.contiguous(), edge_weight.contiguous(),
                csr_data["edge_src"], csr_data["edge_dst"], csr_data["dst_ptr"], csr_data["csr_perm"],
                x.shape[0], self.cutoff.cutoff_upper, csr_data.get("src_ptr"), csr_data.get("src_perm"))
        elif USE_TRITON_MESSAGE_PASSING:  # reference :677-705
            _need_gpu(x, "CFConv")
            filter_out = self.filter_network(edge_attr)
            x = K.fused_cutoff_gather_multiply_scatter_autograd(
                x.contiguous(), filter_out.contiguous(), edge_weight.contiguous(),
                edge_index[0].contiguous(), edge_index[1].contiguous(), x.shape[0],
                self.cutoff.cutoff_upper,
                csr_data.get("src_ptr") if csr_data else None,
                csr_data.get("src_perm") if csr_data else None)
        else:  # reference :706-715 (PyTorch path: the parity oracle's branch)
            C = self.cutoff(edge_weight)
            W = self.filter_network(edge_attr) * C.view(-1, 1)
            msg = x.index_select(0, edge_index[0]) * W
            x = torch.zeros_like(x).index_add_(0, edge_index[1], msg)
        return self.lin2(x)


class InteractionBlock(torch.nn.Module):
    """reference models/schnet.py:453-548"""

    def __init__(self, cfconv_layer, hidden_channels=128, activation=None):
        super().__init__()
        self.conv = cfconv_layer
        self.activation = activation or torch.nn.Tanh()
        self.lin = torch.nn.Linear(hidden_channels, hidden_channels)
        self.use_fused_tanh_linear = USE_FUSED_TANH_LINEAR and isinstance(self.activation, torch.nn.Tanh)
        init_xavier_uniform(self.lin)

    def forward(self, x, edge_index, edge_weight, edge_attr, *args, csr_data: dict = None):
        x = self.conv(x, edge_index, edge_weight, edge_attr, csr_data=csr_data)
        if self.use_fused_tanh_linear:  # reference :534-541
            _need_gpu(x, "InteractionBlock")
            return K.fused_tanh_linear_autograd(x.contiguous(), self.lin.weight.t().contiguous(),
                                                self.lin.bias)
        return self.lin(self.activation(x))


class SchNet(torch.nn.Module):
    """reference models/schnet.py:94-450 (forward :178-369)"""

    name: Final[str] = "SchNet"

    def __init__(self, embedding_layer, interaction_blocks, rbf_layer, output_network,
                 max_num_neighbors: int = 1000):
        super().__init__()
        self.embedding_layer, self.rbf_layer = embedding_layer, rbf_layer
        self.max_num_neighbors = max_num_neighbors
        self.interaction_blocks = torch.nn.Sequential(*interaction_blocks)
        self.output_network = output_network
        self._cached_gamma = None
        self._engine = None

    def is_nl_compatible(self, nl) -> bool:
        """reference models/schnet.py:371-380 (+ neighbor_list/neighbor_list.py:131-163): a neighbour list the
        batch brings along is used when it is an order-2 list without self interaction at this model's cutoff."""
        if not isinstance(nl, dict):
            return False
        need = {"tag": str, "order": int, "index_mapping": torch.Tensor, "cell_shifts": torch.Tensor,
                "rcut": float, "self_interaction": bool}
        if not all(isinstance(nl.get(k), t) for k, t in need.items()):
            return False
        return nl["order"] == 2 and nl["self_interaction"] is False \
            and nl["rcut"] == self.rbf_layer.cutoff.cutoff_upper

    # -- neighbour list: a compatible list of the batch is used as it is (reference :216-245); otherwise it is
    #    rebuilt every forward, like the reference does in practice
    def _edges(self, data, cutoff: float) -> torch.Tensor:
        nl = getattr(data, "neighbor_list", None)
        nl = nl.get(self.name) if nl else None
        if self.is_nl_compatible(nl):
            return nl["index_mapping"]
        ptr = _mol_ptr(data)
        mb = int(getattr(data, "max_beads_per_mol", 0))
        if mb <= 0:
            mb = int((ptr[1:] - ptr[:-1]).max().item()) if ptr.numel() > 1 else 1
        if self.max_num_neighbors < mb - 1:
            # torch_cluster would keep an implementation-defined subset of max_num_neighbors neighbours
            # (neighbor_list/torch_impl.py:216-224); this build has no such truncation and refuses to guess
            raise NotImplementedError(
                f"max_num_neighbors={self.max_num_neighbors} can truncate the neighbourhoods of a {mb}-bead "
                "molecule; the MI355X path builds the full within-cutoff graph only")
        if any_hip_gate_on():
            _need_gpu(data.pos, "SchNet neighbour list")
            return K.radius_graph(data.pos, cutoff, data.batch, ptr_=ptr,
                                  max_beads_per_mol=int(getattr(data, "max_beads_per_mol", 0)))
        return radius_graph_torch(data.pos.detach(), ptr, cutoff)

    def forward(self, data):
        x = self.embedding_layer(data.atom_types)
        cutoff_upper = self.rbf_layer.cutoff.cutoff_upper
        edge_index = self._edges(data, cutoff_upper)
        # the CSR indices are built first (the reference builds them after the RBF expansion, :309-334): with both at
        # hand the fused RBF operator's backward needs no float atomics
        csr_data = None
        if USE_CSR or USE_SRC_CSR_GRAD_X:  # reference :309-334
            _need_gpu(x, "SchNet CSR")
            edge_src, edge_dst = edge_index[0].contiguous(), edge_index[1].contiguous()
            csr_data = {"edge_src": edge_src, "edge_dst": edge_dst}
            if USE_CSR:
                csr_data["dst_ptr"], csr_data["csr_perm"] = K.build_csr_index(edge_dst, x.shape[0])
            if USE_SRC_CSR_GRAD_X:
                csr_data["src_ptr"], csr_data["src_perm"] = K.build_src_csr_index(edge_src, x.shape[0])

        if USE_FUSED_RBF == "1":  # reference :261-283
            _need_gpu(data.pos, "SchNet fused RBF")
            if self._cached_gamma is None:
                self._cached_gamma = self.rbf_layer.coeff.item()
            if csr_data is not None and "dst_ptr" in csr_data and "src_ptr" in csr_data:
                distances, rbf_expansion = K.fused_distance_gaussian_rbf_cutoff_with_csr(
                    data.pos.contiguous(), csr_data["edge_src"], csr_data["edge_dst"], self.rbf_layer.offset,
                    self._cached_gamma, cutoff_upper, csr_data["dst_ptr"], csr_data["csr_perm"],
                    csr_data["src_ptr"], csr_data["src_perm"])
            else:
                distances, rbf_expansion = K.fused_distance_gaussian_rbf_cutoff_autograd(
                    data.pos.contiguous(), edge_index[0].contiguous(), edge_index[1].contiguous(),
                    self.rbf_layer.offset, self._cached_gamma, cutoff_upper)
        else:  # reference :287-299
            dr = data.pos[edge_index[1]] - data.pos[edge_index[0]]
            distances = dr.norm(p=2, dim=1)
            rbf_expansion = self.rbf_layer(distances)

        for block in self.interaction_blocks:  # reference :336-341
            x = x + block(x, edge_index, distances, rbf_expansion, csr_data=csr_data)

        energy = self.output_network(x, data)  # reference :346
        num_graphs = _mol_ptr(data).numel() - 1
        out = torch.zeros(num_graphs, energy.shape[1], dtype=energy.dtype, device=energy.device)
        energy = out.index_add_(0, data.batch, energy).flatten()  # reference :356-357
        data.out[self.name] = {ENERGY_KEY: energy}
        return data


class StandardSchNet(SchNet):
    """reference models/schnet.py:743-841"""

    def __init__(self, rbf_layer, cutoff, output_hidden_layer_widths: List[int],
                 hidden_channels: int = 128, embedding_size: int = 100, num_filters: int = 128,
                 num_interactions: int = 3, activation=None, max_num_neighbors: int = 1000):
        if num_interactions < 1:
            raise ValueError("At least one interaction block must be specified")
        activation = activation or torch.nn.Tanh()
        embedding_layer = torch.nn.Embedding(embedding_size, hidden_channels)
        blocks = []
        for _ in range(num_interactions):
            filter_network = MLP([rbf_layer.num_rbf, num_filters, num_filters], activation,
                                 last_bias=False)
            conv = CFConv(filter_network, cutoff, hidden_channels, hidden_channels, num_filters)
            blocks.append(InteractionBlock(conv, hidden_channels, activation))
        output_network = MLP([hidden_channels] + list(output_hidden_layer_widths) + [1], activation,
                             last_bias=False)
        super().__init__(embedding_layer, blocks, rbf_layer, output_network, max_num_neighbors)


class GradientsOut(torch.nn.Module):
    """forces = -d(sum E)/d(pos)  (reference models/gradients.py:194-293)."""

    def __init__(self, model: torch.nn.Module, targets: str = FORCE_KEY):
        super().__init__()
        self.model, self.name = model, model.name
        self.targets = [targets] if isinstance(targets, str) else list(targets)

    def forward(self, data):
        eng = engine_for(self.model, data)
        if eng is not None:  # fused HIP engine: energy and forces in one pass, no autograd graph
            energy, forces = eng.energy_forces(data.pos, data.atom_types, _mol_ptr(data),
                                               int(getattr(data, "max_beads_per_mol", 0)))
            data.out[self.name] = {ENERGY_KEY: energy, FORCE_KEY: forces}
            data.pos = data.pos.detach()
            return data
        data.pos.requires_grad_(True)
        data = self.model(data)
        if FORCE_KEY in self.targets:
            y = data.out[self.name][ENERGY_KEY]
            dy_dr = torch.autograd.grad(y.sum(), data.pos, create_graph=self.training)[0]
            data.out[self.name][FORCE_KEY] = -dy_dr
        data.pos = data.pos.detach()
        return data


def engine_for(model, data):
    """The fused engine when the gate is on and the model/device qualify, else None."""
    if not (USE_FUSED_ENGINE and any_hip_gate_on()):
        return None
    from . import engine  # deferred: engine imports the C library

    return engine.maybe_engine(model, data)


class Batch:
    """Minimal stand-in for the collated AtomicData the reference passes around
    (reference data/atomic_data.py:21-237): just the attributes the force path reads."""

    def __init__(self, pos, atom_types, ptr, masses=None, max_beads_per_mol: int = 0):
        self.pos, self.atom_types, self.ptr = pos, atom_types, ptr
        n_mol = ptr.numel() - 1
        self.batch = torch.repeat_interleave(
            torch.arange(n_mol, device=pos.device), (ptr[1:] - ptr[:-1]).to(pos.device))
        self.masses = masses
        self.neighbor_list, self.out = {}, {}
        self.max_beads_per_mol = max_beads_per_mol


def _state_dict_from_file(path: str) -> dict:
    """Name -> tensor from a ``state_dict`` file.  ``.npz``: numpy arrays under the reference's parameter names;
    anything else: ``torch.load(..., weights_only=True)`` -- a plain ``state_dict`` (or ``{"state_dict": ...}``), never
    a pickled module (the reference's published ``model_and_prior.pt`` is one: it needs the ``mlcg`` package to
    unpickle and is out of scope; export ``model.state_dict()`` where that package exists)."""
    if path.endswith(".npz"):
        import numpy as np

        z = np.load(path)  # allow_pickle stays False
        return {k: torch.from_numpy(z[k]) for k in z.files}
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    if not isinstance(sd, dict) or not all(isinstance(v, torch.Tensor) for v in sd.values()):
        raise ValueError(f"{path}: expected a state_dict (name -> tensor)")
    return sd


def find_schnet_prefix(sd) -> str:
    """Prefix under which a StandardSchNet sits in a state_dict: "" for the bare model, "model." inside a
    ``GradientsOut``, "models.SchNet.model." inside the reference's ``SumOut`` container (models/gradients.py:19-152:
    ``SumOut.models[name] = GradientsOut(model)``) -- whatever precedes ``embedding_layer.weight``."""
    hits = [k[: -len("embedding_layer.weight")] for k in sd if k.endswith("embedding_layer.weight")]
    hits = [p for p in hits if p + "rbf_layer.offset" in sd and p + "interaction_blocks.0.conv.lin1.weight" in sd]
    if len(hits) != 1:
        raise KeyError(f"state_dict holds {len(hits)} StandardSchNet-shaped sub-models (prefixes {hits}); pass prefix=")
    return hits[0]


def model_from_state_dict(sd, prefix: Optional[str] = None, cutoff: Optional[float] = None,
                          max_num_neighbors: int = 1000) -> StandardSchNet:
    """Row N4: a ``StandardSchNet`` from a state_dict keyed by the reference's parameter names (reference
    models/schnet.py:778-841; the same names this mirror uses: ``embedding_layer.weight``, ``rbf_layer.offset`` /
    ``.coeff``, ``interaction_blocks.<l>.conv.{lin1,lin2,filter_network.layers.<k>}``, ``interaction_blocks.<l>.lin``,
    ``output_network.layers.<k>``).  ``sd``: a dict or a path (``.npz`` or a weights-only ``torch.load`` file);
    ``prefix``: where the SchNet sits (None: found by ``find_schnet_prefix``, so a ``GradientsOut`` / ``SumOut``
    container's state_dict loads as it is).

    A state_dict does not encode every hyper-parameter.  Read off the shapes: widths, number of blocks, basis size,
    embedding size.  NOT in the file and therefore fixed or to be supplied: the activation (tanh, the reference's
    default and the only one the HIP path implements), the cutoff function (``CosineCutoff`` with lower bound 0), and
    the cutoff radius: ``cutoff=None`` takes the last Gaussian centre, which is right for the reference's untrained
    basis (``offset = linspace(0, cutoff, R)``, ``coeff = -0.5 / (offset[1] - offset[0])**2``,
    radial_basis/gaussian.py:56-75) and is VERIFIED against exactly that form -- a trained basis
    (``trainable=True`` registers both as Parameters) does not pass and needs ``cutoff=`` spelled out."""
    if isinstance(sd, str):
        sd = _state_dict_from_file(sd)
    if prefix is None:
        prefix = find_schnet_prefix(sd)
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    need = ["embedding_layer.weight", "rbf_layer.offset", "interaction_blocks.0.conv.lin1.weight"]
    missing = [k for k in need if k not in sd]
    if missing:
        raise KeyError(f"state_dict lacks {missing}: not a StandardSchNet under the reference's names")
    emb = sd["embedding_layer.weight"]
    offset = sd["rbf_layer.offset"].flatten().to(torch.float32)
    n_blocks = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("interaction_blocks."))
    filt = sd["interaction_blocks.0.conv.filter_network.layers.0.weight"]
    out_idx = sorted({int(k.split(".")[2]) for k in sd if k.startswith("output_network.layers.") and k.endswith(".weight")})
    widths = [int(sd[f"output_network.layers.{i}.weight"].shape[0]) for i in out_idx[:-1]]
    if cutoff is None:
        rc = float(offset[-1])
        lin = torch.linspace(0.0, rc, offset.numel())
        coeff = sd.get("rbf_layer.coeff")
        want = -0.5 / float(lin[1] - lin[0]) ** 2
        if float(offset[0]) != 0.0 or not torch.allclose(offset, lin, rtol=1e-6, atol=1e-6 * rc) \
                or coeff is None or abs(float(coeff) - want) > 1e-5 * abs(want):
            raise ValueError("rbf_layer.offset / .coeff are not the untrained basis linspace(0, cutoff, R), "
                             "-0.5 / spacing^2 (cutoff_lower != 0 or a trained basis): the cutoff radius cannot be "
                             "read off the weights, pass cutoff=")
    else:
        rc = float(cutoff)
    m = StandardSchNet(GaussianBasis(CosineCutoff(0.0, rc), int(offset.numel())), CosineCutoff(0.0, rc), widths,
                       hidden_channels=int(emb.shape[1]), embedding_size=int(emb.shape[0]),
                       num_filters=int(filt.shape[0]), num_interactions=n_blocks, max_num_neighbors=max_num_neighbors)
    m.load_state_dict({k: v.to(torch.float32) for k, v in sd.items()}, strict=True)
    return m


def model_from_params(p: dict) -> StandardSchNet:
    """Build a StandardSchNet from the flat weight dict used by the fixtures (tests/golden)."""
    rc = p["cutoff"]
    widths = [o["w"].shape[0] for o in p["out"][:-1]]
    m = StandardSchNet(GaussianBasis(CosineCutoff(0.0, rc), p["num_rbf"]), CosineCutoff(0.0, rc),
                       widths, hidden_channels=p["hidden"], embedding_size=p["emb"].shape[0],
                       num_filters=p["filters"], num_interactions=len(p["blocks"]))
    with torch.no_grad():
        m.embedding_layer.weight.copy_(p["emb"])
        for blk, b in zip(m.interaction_blocks, p["blocks"]):
            blk.conv.lin1.weight.copy_(b["lin1_w"])
            blk.conv.filter_network.layers[0].weight.copy_(b["f0_w"])
            blk.conv.filter_network.layers[0].bias.copy_(b["f0_b"])
            blk.conv.filter_network.layers[2].weight.copy_(b["f1_w"])
            blk.conv.lin2.weight.copy_(b["lin2_w"])
            blk.conv.lin2.bias.copy_(b["lin2_b"])
            blk.lin.weight.copy_(b["lin_w"])
            blk.lin.bias.copy_(b["lin_b"])
        lins = [l for l in m.output_network.layers if isinstance(l, torch.nn.Linear)]
        for lin, o in zip(lins, p["out"]):
            lin.weight.copy_(o["w"])
            if o["b"] is not None:
                lin.bias.copy_(o["b"])
    return m
