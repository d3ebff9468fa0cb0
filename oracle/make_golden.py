"""Generate tests/golden/*.npz by running the REFERENCE's own SchNet code (build container only).

Run once, here, with /root/reference present:   python oracle/make_golden.py
The GPU box never runs this (it has no /root/reference); it only reads the .npz fixtures.

What runs: the reference's ``StandardSchNet`` / ``InteractionBlock`` / ``CFConv`` (PyTorch
branch, all MLCG_USE_* = "0") / ``MLP`` / ``GaussianBasis`` / ``CosineCutoff`` /
``compute_distances`` / ``GradientsOut``, imported from /root/reference/src.  Three wheels the
reference needs are not installed in the container (nvtx, torch_geometric, torch_cluster);
tiny in-process stand-ins for exactly the symbols the path touches are registered in
``sys.modules`` first.  No reference source text is copied anywhere.

Fixtures are DATA: inputs (weights, positions, types) and the reference's outputs (fp64
energies / forces, and for the small case the per-stage tensors), plus this script.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

for k in ("MLCG_USE_TRITON_MESSAGE_PASSING", "MLCG_USE_FUSED_RBF", "MLCG_USE_FUSED_TANH_LINEAR",
          "MLCG_USE_CSR", "MLCG_USE_SRC_CSR_GRAD_X"):
    os.environ[k] = "0"


def _install_standins():
    # ---- nvtx: annotate usable as decorator and as context manager
    nvtx = types.ModuleType("nvtx")

    class annotate:
        def __init__(self, *a, **k):
            pass

        def __call__(self, fn):
            return fn

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    nvtx.annotate = annotate
    sys.modules["nvtx"] = nvtx

    # ---- torch_geometric: MessagePassing.propagate (gather edge_index[0] -> message ->
    #      sum at edge_index[1]), utils.scatter(sum), data.Data, data.collate
    tg = types.ModuleType("torch_geometric")
    tg_nn = types.ModuleType("torch_geometric.nn")
    tg_utils = types.ModuleType("torch_geometric.utils")
    tg_data = types.ModuleType("torch_geometric.data")
    tg_collate = types.ModuleType("torch_geometric.data.collate")

    class MessagePassing(torch.nn.Module):
        def __init__(self, aggr="add", **kw):
            super().__init__()
            assert aggr == "add"

        def propagate(self, edge_index, size=None, **kwargs):
            x = kwargs.pop("x")
            x_j = x.index_select(0, edge_index[0])
            msg = self.message(x_j=x_j, **kwargs)
            out = torch.zeros(x.shape[0], msg.shape[1], dtype=msg.dtype)
            return out.index_add_(0, edge_index[1], msg)

    def scatter(src, index, dim=0, reduce="sum", dim_size=None):
        assert reduce == "sum" and dim == 0
        n = int(index.max()) + 1 if dim_size is None else dim_size
        out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
        return out.index_add_(0, index, src)

    class Data:
        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        def __contains__(self, k):
            return hasattr(self, k)

        def __getitem__(self, k):
            return getattr(self, k)

        def __setitem__(self, k, v):
            setattr(self, k, v)

    tg_nn.MessagePassing = MessagePassing
    tg_utils.scatter = scatter
    tg_data.Data = Data
    tg_collate.collate = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    tg.nn, tg.utils, tg.data = tg_nn, tg_utils, tg_data
    tg_data.collate = tg_collate
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tg_nn,
                        "torch_geometric.utils": tg_utils, "torch_geometric.data": tg_data,
                        "torch_geometric.data.collate": tg_collate})

    # ---- torch_cluster: brute-force radius_graph (strict <, same molecule, no self loops,
    #      rows [centre, neighbour] sorted by centre then neighbour for flow=target_to_source)
    tc = types.ModuleType("torch_cluster")

    def radius_graph(pos, r, batch=None, loop=False, max_num_neighbors=32,
                     flow="source_to_target", num_workers=1):
        from oracle.schnet_oracle import radius_graph as rg
        n = pos.shape[0]
        if batch is None:
            batch = torch.zeros(n, dtype=torch.long)
        counts = torch.bincount(batch)
        ptr = torch.zeros(counts.numel() + 1, dtype=torch.long)
        ptr[1:] = counts.cumsum(0)
        i, j = rg(pos.detach(), ptr, r)
        assert not loop and flow == "target_to_source"
        return torch.stack([i, j])

    tc.radius_graph = radius_graph
    tc.radius = None
    sys.modules["torch_cluster"] = tc


def _reference_model(num_rbf, L, rc, out_widths, hidden=128, filters=128, seed=0):
    from flashmd.models.schnet import StandardSchNet
    from flashmd.models.radial_basis.gaussian import GaussianBasis
    from flashmd.models.cutoff import CosineCutoff

    torch.manual_seed(seed)
    model = StandardSchNet(
        GaussianBasis(CosineCutoff(0.0, rc), num_rbf), CosineCutoff(0.0, rc), list(out_widths),
        hidden_channels=hidden, num_filters=filters, num_interactions=L)
    return model


def _params_from_reference(model, rc, bias_seed=None):
    """Pull the reference module's tensors into the oracle's flat dict (fp32)."""
    if bias_seed is not None:  # reference zero-initialises biases; perturb so tests cover them
        g = torch.Generator().manual_seed(bias_seed)
        with torch.no_grad():
            for name, prm in model.named_parameters():
                if name.endswith("bias"):
                    prm.copy_((torch.rand(prm.shape, generator=g) * 2 - 1) * 0.1)
    p = {"hidden": model.embedding_layer.embedding_dim, "num_rbf": model.rbf_layer.num_rbf,
         "cutoff": float(rc), "emb": model.embedding_layer.weight.detach().clone(),
         "blocks": [], "out": []}
    for blk in model.interaction_blocks:
        fl = [m for m in blk.conv.filter_network.layers if isinstance(m, torch.nn.Linear)]
        p["blocks"].append({
            "lin1_w": blk.conv.lin1.weight.detach().clone(),
            "f0_w": fl[0].weight.detach().clone(), "f0_b": fl[0].bias.detach().clone(),
            "f1_w": fl[1].weight.detach().clone(),
            "lin2_w": blk.conv.lin2.weight.detach().clone(),
            "lin2_b": blk.conv.lin2.bias.detach().clone(),
            "lin_w": blk.lin.weight.detach().clone(), "lin_b": blk.lin.bias.detach().clone()})
    p["filters"] = p["blocks"][0]["f1_w"].shape[0]
    ol = [m for m in model.output_network.layers if isinstance(m, torch.nn.Linear)]
    for m in ol:
        p["out"].append({"w": m.weight.detach().clone(),
                         "b": None if m.bias is None else m.bias.detach().clone()})
    return p


def flatten_params(p):
    d = {"meta": np.array([p["hidden"], p["filters"], p["num_rbf"], len(p["blocks"]),
                           len(p["out"])], dtype=np.int64),
         "cutoff": np.array([p["cutoff"]], dtype=np.float64), "emb": p["emb"].numpy()}
    for l, b in enumerate(p["blocks"]):
        for k, v in b.items():
            d[f"b{l}_{k}"] = v.numpy()
    for l, o in enumerate(p["out"]):
        d[f"o{l}_w"] = o["w"].numpy()
        if o["b"] is not None:
            d[f"o{l}_b"] = o["b"].numpy()
    return d


def _run_reference(model, pos, types_, ptr, dtype):
    from flashmd.models.gradients import GradientsOut
    from torch_geometric.data import Data

    m = GradientsOut(model.to(dtype), targets="forces").eval()
    n_mol = ptr.numel() - 1
    batch = torch.repeat_interleave(torch.arange(n_mol), ptr[1:] - ptr[:-1])
    data = Data(pos=pos.to(dtype).clone(), atom_types=types_, batch=batch, ptr=ptr,
                neighbor_list={}, out={})
    data = m(data)
    return data.out["SchNet"]["energy"].detach(), data.out["SchNet"]["forces"].detach()


def case_config1(model, p, so, rc):
    """BASELINE configs[0] at its stated size: 4 molecules x 269 beads, the reference's --disable_optim path."""
    pos, types_, ptr = so.synth_batch(4, 269, rc)
    e64, f64 = _run_reference(model, pos, types_, ptr, torch.float64)
    e32, f32 = _run_reference(model, pos, types_, ptr, torch.float32)
    oe, of = so.schnet_energy_forces(so.params_to(p, torch.float64), pos.double(), types_, ptr)
    assert float((of - f64).norm() / f64.norm()) < 1e-12 and float((oe - e64).abs().max() / e64.abs().max()) < 1e-12
    print(f"[1enh x4] reference fp32 vs fp64: forces {float((f32.double() - f64).norm() / f64.norm()):.2e}")
    np.savez_compressed(
        os.path.join(GOLD, "case_1enh_4x269.npz"), pos=pos.numpy(), types=types_.numpy(), ptr=ptr.numpy(),
        energy_ref64=e64.numpy(), forces_ref64=f64.numpy(), energy_ref32=e32.numpy(), forces_ref32=f32.numpy())


def main():
    sys.path.insert(0, "/root/reference/src")
    _install_standins()
    from oracle import schnet_oracle as so

    os.makedirs(GOLD, exist_ok=True)
    rc, R, L = 5.0, 50, 3

    # ---- weights: reference-initialised StandardSchNet (xavier, N(0,1) embedding), biases
    #      perturbed to +-0.1 so that every bias term is exercised
    model = _reference_model(R, L, rc, (128, 64))
    p = _params_from_reference(model, rc, bias_seed=11)
    if "--only-config1" in sys.argv:   # adds case_1enh_4x269.npz without rewriting the other fixtures
        case_config1(model, p, so, rc)
        return
    np.savez_compressed(os.path.join(GOLD, "weights_L3_R50.npz"), **flatten_params(p))

    # ---- case A: 3 small molecules (37 beads), all per-stage tensors, fp64 reference
    pos, types_, ptr = so.synth_batch(3, 37, rc)
    e64, f64 = _run_reference(model, pos, types_, ptr, torch.float64)
    e32, f32 = _run_reference(model, pos, types_, ptr, torch.float32)
    src, dst = so.radius_graph(pos, ptr, rc)
    # restatement vs reference, same dtype
    oe, of = so.schnet_energy_forces(so.params_to(p, torch.float64), pos.double(), types_, ptr)
    err_e = float((oe - e64).abs().max() / e64.abs().max())
    err_f = float((of - f64).norm() / f64.norm())
    print(f"[small] oracle vs reference fp64: energy {err_e:.2e}  forces {err_f:.2e}")
    assert err_e < 1e-12 and err_f < 1e-12
    st = []
    so.schnet_energy_forces(so.params_to(p, torch.float64), pos.double(), types_, ptr, stages=st)
    np.savez_compressed(
        os.path.join(GOLD, "case_small_3x37.npz"), pos=pos.numpy(), types=types_.numpy(),
        ptr=ptr.numpy(), edge_src=src.numpy(), edge_dst=dst.numpy(),
        energy_ref64=e64.numpy(), forces_ref64=f64.numpy(),
        energy_ref32=e32.numpy(), forces_ref32=f32.numpy(),
        d=st[0]["d"].numpy(), rbf=st[0]["rbf"].numpy().astype(np.float32),
        agg0=st[1]["agg"].numpy(), agg1=st[2]["agg"].numpy(), agg2=st[3]["agg"].numpy())

    # ---- case B: 2 x 269 beads (1ENH-shaped), fp64 reference energies/forces only
    pos, types_, ptr = so.synth_batch(2, 269, rc)
    e64, f64 = _run_reference(model, pos, types_, ptr, torch.float64)
    e32, f32 = _run_reference(model, pos, types_, ptr, torch.float32)
    src, dst = so.radius_graph(pos, ptr, rc)
    oe, of = so.schnet_energy_forces(so.params_to(p, torch.float64), pos.double(), types_, ptr)
    err_e = float((oe - e64).abs().max() / e64.abs().max())
    err_f = float((of - f64).norm() / f64.norm())
    print(f"[1enh x2] oracle vs reference fp64: energy {err_e:.2e}  forces {err_f:.2e}; "
          f"E/N = {src.numel() / pos.shape[0]:.1f}")
    print(f"[1enh x2] reference fp32 vs fp64: forces "
          f"{float((f32.double() - f64).norm() / f64.norm()):.2e}")
    assert err_e < 1e-12 and err_f < 1e-12
    np.savez_compressed(
        os.path.join(GOLD, "case_1enh_2x269.npz"), pos=pos.numpy(), types=types_.numpy(),
        ptr=ptr.numpy(), n_edges=np.array([src.numel()]),
        energy_ref64=e64.numpy(), forces_ref64=f64.numpy(),
        energy_ref32=e32.numpy(), forces_ref32=f32.numpy())

    # ---- case C: ragged batch (molecules of 5, 64, 1, 33 beads; a 1-bead molecule has no edges)
    parts = [so.synth_molecule(n, rc, 50 + k) for k, n in enumerate((5, 64, 1, 33))]
    pos = torch.cat(parts)
    ptr = torch.tensor([0, 5, 69, 70, 103])
    g = torch.Generator().manual_seed(3)
    types_ = torch.randint(1, 25, (103,), generator=g)
    e64, f64 = _run_reference(model, pos, types_, ptr, torch.float64)
    oe, of = so.schnet_energy_forces(so.params_to(p, torch.float64), pos.double(), types_, ptr)
    assert float((of - f64).norm() / f64.norm()) < 1e-12
    np.savez_compressed(
        os.path.join(GOLD, "case_ragged.npz"), pos=pos.numpy(), types=types_.numpy(),
        ptr=ptr.numpy(), energy_ref64=e64.numpy(), forces_ref64=f64.numpy())
    case_config1(model, p, so, rc)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
