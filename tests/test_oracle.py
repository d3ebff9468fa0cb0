"""CPU: the oracle against the reference's own vectors (tests/golden, written by
oracle/make_golden.py from the reference's StandardSchNet) and closed-form end points."""
import os

import numpy as np
import pytest
import torch

from oracle import schnet_oracle as so


@pytest.fixture(scope="module")
def params(golden_dir):
    return so.params_from_npz(os.path.join(golden_dir, "weights_L3_R50.npz"))


def test_cutoff_endpoints():
    # reference tests/models/test_cutoff.py:27-42: C(0) == 1, C(10) == 0 for (0, 5)
    d = torch.linspace(0, 10, 100)
    c = so.cosine_cutoff(d, 5.0)
    assert c[0] == 1 and c[-1] == 0
    assert float(so.cosine_cutoff(torch.tensor([5.0]), 5.0)) == 0.0


def test_rbf_centers_match_reference_definition():
    c, gamma = so.rbf_centers(50, 5.0)
    assert c[0] == 0 and abs(float(c[-1]) - 5.0) < 1e-6
    assert abs(gamma + 0.5 / (5.0 / 49) ** 2) < 1e-3


@pytest.mark.parametrize("case", ["case_small_3x37", "case_1enh_2x269", "case_ragged"])
def test_energy_forces_vs_reference_fp64(params, golden_dir, case):
    z = np.load(os.path.join(golden_dir, case + ".npz"))
    pos, types, ptr = (torch.from_numpy(z[k]) for k in ("pos", "types", "ptr"))
    e, f = so.schnet_energy_forces(so.params_to(params, torch.float64), pos.double(), types, ptr)
    np.testing.assert_allclose(e.numpy(), z["energy_ref64"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(f.numpy(), z["forces_ref64"], rtol=1e-10, atol=1e-12)


def test_fp32_oracle_within_budget_of_fp64_reference(params, golden_dir):
    z = np.load(os.path.join(golden_dir, "case_1enh_2x269.npz"))
    pos, types, ptr = (torch.from_numpy(z[k]) for k in ("pos", "types", "ptr"))
    e, f = so.schnet_energy_forces(params, pos, types, ptr)
    rel = np.linalg.norm(f.numpy() - z["forces_ref64"]) / np.linalg.norm(z["forces_ref64"])
    assert rel < 1e-5  # fp32 summation-order noise; the parity budget is 1e-4


def test_edges_and_stages_small(params, golden_dir):
    z = np.load(os.path.join(golden_dir, "case_small_3x37.npz"))
    pos, types, ptr = (torch.from_numpy(z[k]) for k in ("pos", "types", "ptr"))
    src, dst = so.radius_graph(pos, ptr, 5.0)
    assert np.array_equal(src.numpy(), z["edge_src"]) and np.array_equal(dst.numpy(), z["edge_dst"])
    # symmetric graph, sorted by (centre, neighbour), no self loops, no cross-molecule pairs
    key = src * pos.shape[0] + dst
    assert torch.all(key[1:] > key[:-1]) and torch.all(src != dst)
    assert set(zip(src.tolist(), dst.tolist())) == set(zip(dst.tolist(), src.tolist()))
    d = so.compute_distances(pos.double(), src, dst)
    np.testing.assert_allclose(d.numpy(), z["d"], rtol=1e-12)
    assert float(d.max()) < 5.0


def test_csr_oracle_properties():
    g = torch.Generator().manual_seed(0)
    keys = torch.randint(0, 40, (1000,), generator=g)
    ptr, perm = so.build_csr(keys, 50)
    assert ptr[0] == 0 and ptr[-1] == 1000 and ptr.dtype == torch.int64
    assert torch.equal(ptr[1:] - ptr[:-1], torch.bincount(keys, minlength=50))
    assert torch.equal(torch.sort(perm).values, torch.arange(1000))
    for n in range(50):
        seg = perm[ptr[n]:ptr[n + 1]]
        assert torch.all(keys[seg] == n) and torch.all(seg[1:] > seg[:-1])
    ptr0, perm0 = so.build_csr(torch.zeros(0, dtype=torch.int64), 7)
    assert ptr0.tolist() == [0] * 8 and perm0.numel() == 0


def test_w16a16_oracle_close_to_fp32(params, golden_dir):
    z = np.load(os.path.join(golden_dir, "case_small_3x37.npz"))
    pos, types, ptr = (torch.from_numpy(z[k]) for k in ("pos", "types", "ptr"))
    e, f = so.schnet_energy_forces(params, pos, types, ptr, w16a16=True)
    rel = np.linalg.norm(f.numpy() - z["forces_ref64"]) / np.linalg.norm(z["forces_ref64"])
    assert 1e-6 < rel < 5e-2  # fp16 storage points: not within 1e-4, but the same function
