"""GPU parity, operator granularity, against the REFERENCE'S OWN Triton kernels: tests/golden/
triton_stages_1x37.npz holds inputs and the outputs of the reference kernels K1-K15 executed under
TRITON_INTERPRET=1 by oracle/make_golden_triton.py (one listed autotune configuration each, recorded in the
fixture).  Every HIP operator is called through the C ABI (flashmd_amd.kernels -> ctypes).
Bar: integers bit-exact; fp32 within the tolerance at each assert (path budget 1e-4); fp16-storing kernels:
equal except where the last bit of an fp32 sum or tanh moves an fp16 rounding (fraction bounded below)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def K():
    import flashmd_amd.kernels as kernels

    assert torch.cuda.is_available(), "these tests need the MI355X"
    return kernels


@pytest.fixture(scope="module")
def tri(golden_dir):
    z = np.load(os.path.join(golden_dir, "triton_stages_1x37.npz"))
    return {k: (torch.from_numpy(z[k]).to(DEV) if z[k].dtype.kind in "fi" and z[k].ndim else z[k])
            for k in z.files}


def rel(a, b):
    a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def flips(a, b):
    """fraction of fp16 outputs that differ, and the largest difference in units of the reference's ulp"""
    a, b = a.float().cpu(), b.float().cpu()
    ulp = torch.maximum(b.abs(), torch.tensor(2.0 ** -14)).log2().floor().exp2() * 2.0 ** -10
    return float((a != b).float().mean()), float(((a - b).abs() / ulp).max())


def test_k14_distance_rbf_cutoff_forward(K, tri):
    d, rbf = K.fused_distance_gaussian_rbf_cutoff(tri["pos"], tri["edge_src"], tri["edge_dst"], tri["centers"],
                                                  float(tri["gamma"][0]), float(tri["cutoff"][0]))
    assert rel(d, tri["k14_d"]) < 2e-7 and rel(rbf, tri["k14_rbf"]) < 2e-6


def test_k3_k15_fp32_linears(K, tri):
    mg = int(tri["mg"][0])
    assert rel(K.fused_linear_tanh(tri["k14_rbf"][:mg].contiguous(), tri["k3_w"], tri["k3_b"]), tri["k3_y"]) < 2e-6
    assert rel(K.fused_tanh_linear(tri["k15_x"], tri["k15_w"], tri["k15_b"]), tri["k15_y"]) < 2e-6


def test_k4_k5_w16a16_forward(K, tri):
    hid_ref = tri["k4_y"]
    hid = K.fused_linear_tanh_fp16(tri["k14_rbf"], tri["k4_w"], tri["k4_b"])
    fr, mx = flips(hid, hid_ref)
    print(f"K4: {fr:.2e} of the fp16 outputs differ, max {mx:.1f} ulp")
    assert hid.dtype == torch.float16 and fr < 1e-2 and mx <= 1.0
    fr, mx = flips(K.fused_linear_tanh_fp16(hid_ref[:37].contiguous(), tri["k4h_w"], tri["k4h_b"]), tri["k4h_y"])
    assert fr < 1e-2 and mx <= 1.0
    y16 = K.linear_fp16(hid_ref, tri["k5_w"], out_dtype=torch.float16)
    fr, mx = flips(y16, tri["k5_y16"])
    print(f"K5->f16: {fr:.2e} differ, max {mx:.1f} ulp")
    assert y16.dtype == torch.float16 and fr < 1e-2 and mx <= 1.0
    assert rel(K.linear_fp16(hid_ref, tri["k5_w"]), tri["k5_y32"]) < 1e-6


def test_k6_k7_w16a16_backward(K, tri):
    """The reference's K6 rounds g*(1-y^2) to fp16 before the dot (cfconv_kernels.py:1019-1023); K7 rounds A
    (:1198).  Inputs to the dots are then identical fp16 values, so only the fp32 accumulation order is left."""
    mg = int(tri["mg"][0])
    gx = K.fused_tanh_backward_grad_x_v2(tri["k6_g"], tri["k4_y"][:mg].contiguous(), tri["k6_wt"])
    r6 = rel(gx, tri["k6_gx"])
    c = K.matmul_fp32_fp16_to_fp16_v2(tri["k7_a"], tri["k7_bt"])
    fr, mx = flips(c, tri["k7_c"])
    print(f"K6 rel {r6:.2e}; K7 {fr:.2e} differ, max {mx:.1f} ulp")
    assert gx.dtype == torch.float32 and r6 < 2e-6
    assert c.dtype == torch.float16 and fr < 1e-2 and mx <= 1.0


def test_k11_k12_csr_bit_exact(K, tri):
    n = tri["pos"].shape[0]
    ptr, perm = K.build_csr_index(tri["edge_dst"], n)
    assert torch.equal(ptr, tri["k11_ptr"]) and torch.equal(perm, tri["k11_perm"])
    ptr, perm = K.build_src_csr_index(tri["edge_src"], n)
    assert torch.equal(ptr, tri["k12_ptr"]) and torch.equal(perm, tri["k12_perm"])


@pytest.mark.parametrize("tag", ["32", "16"])
def test_k1_k9_k10_k2_cfconv(K, tri, tag):
    n, rc = tri["pos"].shape[0], float(tri["cutoff"][0])
    x1, g, d, filt = tri["cf_x"], tri["cf_g"], tri["k14_d"], tri["k5_y" + tag]
    src, dst = tri["edge_src"], tri["edge_dst"]
    out = K.fused_csr_cfconv(x1, filt, d, src, tri["k11_ptr"], tri["k11_perm"], n, rc)
    assert rel(out, tri["k9_out" + tag]) < 1e-6
    if tag == "32":
        assert rel(K.fused_cutoff_gather_multiply_scatter(x1, filt, d, src, dst, n, rc), tri["k1_out32"]) < 1e-6
    gx = K.fused_src_csr_grad_x(g, filt, d, dst, tri["k12_ptr"], tri["k12_perm"], n, rc)
    assert rel(gx, tri["k10_gx" + tag]) < 1e-6
    gf = K.fused_grad_filter_out(x1, g, d, src, dst, rc, out_dtype=filt.dtype)
    assert gf.dtype == filt.dtype
    if tag == "32":
        assert rel(gf, tri["k2_gf32"]) < 1e-6
    else:
        fr, mx = flips(gf, tri["k2_gf16"])
        assert fr < 1e-2 and mx <= 1.0
