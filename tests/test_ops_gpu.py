"""GPU parity tests, operator granularity: every HIP operator of flashmd_amd.kernels (called through
the C ABI) against the CPU oracle on the same seeded inputs.  Integer outputs bit-exact; fp32 outputs
within the tolerance written at each assert (budget of the path: 1e-4 relative)."""
import os

import numpy as np
import pytest
import torch

from oracle import schnet_oracle as so

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def K():
    import flashmd_amd.kernels as kernels

    assert torch.cuda.is_available(), "these tests need the MI355X"
    return kernels


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _graph(n_mol=3, n_beads=37):
    pos, types, ptr = so.synth_batch(n_mol, n_beads, 5.0)
    src, dst = so.radius_graph(pos, ptr, 5.0)
    return pos, types, ptr, src, dst


# ---------------------------------------------------------------------------------------- A1
@pytest.mark.parametrize("shape", [(3, 37), (2, 269), (1, 1), (5, 2), (1, 1300)])
def test_radius_graph_bit_exact(K, shape):
    pos, types, ptr = so.synth_batch(shape[0], shape[1], 5.0)
    src, dst = so.radius_graph(pos, ptr, 5.0)
    ei = K.radius_graph(pos.to(DEV), 5.0, None, ptr_=ptr.to(DEV), max_beads_per_mol=shape[1]).cpu()
    assert ei.dtype == torch.int64 and ei.shape == (2, src.numel())
    assert torch.equal(ei[0], src) and torch.equal(ei[1], dst)


def test_radius_graph_ragged_and_batch_arg(K, golden_dir):
    z = np.load(os.path.join(golden_dir, "case_ragged.npz"))
    pos, ptr = torch.from_numpy(z["pos"]), torch.from_numpy(z["ptr"])
    src, dst = so.radius_graph(pos, ptr, 5.0)
    batch = torch.repeat_interleave(torch.arange(ptr.numel() - 1), ptr[1:] - ptr[:-1])
    ei = K.radius_graph(pos.to(DEV), 5.0, batch.to(DEV)).cpu()  # ptr derived from batch, bound unknown
    assert torch.equal(ei[0], src) and torch.equal(ei[1], dst)


def test_radius_graph_cutoff_boundary_is_strict(K):
    # beads exactly rc apart (representable in fp32) must NOT be neighbours; just inside must be
    pos = torch.tensor([[0.0, 0, 0], [5.0, 0, 0], [0, 4.9999995, 0], [3.0, 4.0, 0]])
    ptr = torch.tensor([0, 4])
    src, dst = so.radius_graph(pos, ptr, 5.0)
    ei = K.radius_graph(pos.to(DEV), 5.0, None, ptr_=ptr.to(DEV)).cpu()
    assert torch.equal(ei[0], src) and torch.equal(ei[1], dst)
    pairs = set(zip(ei[0].tolist(), ei[1].tolist()))
    assert (0, 1) not in pairs and (0, 3) not in pairs and (0, 2) in pairs


def test_radius_graph_pairs_within_an_ulp_of_the_cutoff(K, golden_dir):
    """480 beads on a sphere of radius 5 (1 +- 2e-7) around bead 0: hundreds of squared distances within one or two
    ulps of rc^2, where a fused d2 = dx dx + fma(dz, dz, dy dy) and the separately rounded (dx dx + dy dy) + dz dz of
    the oracle disagree.  The operator's edge list and the engine's pair count must both match the oracle exactly
    (the kernels had been contracted once: the __fmul_rn / __fadd_rn intrinsics do not stop it)."""
    import flashmd_amd.schnet as S
    from flashmd_amd import engine

    g = torch.Generator().manual_seed(11)
    n = 480
    u = torch.randn(n, 3, generator=g, dtype=torch.float64)
    u = u / u.norm(dim=1, keepdim=True)
    r = 5.0 * (1.0 + 2e-7 * (2 * torch.rand(n, 1, generator=g, dtype=torch.float64) - 1))
    pos = (u * r).float()
    pos[0] = 0.0
    ptr = torch.tensor([0, n])
    src, dst = so.radius_graph(pos, ptr, 5.0)
    d2 = (pos[1:] ** 2).sum(1)
    near = int(((d2 - 25.0).abs() <= 4e-6).sum())
    inside = int((src == 0).sum())
    assert near > 200 and 50 < inside < n - 50   # the case is what it claims to be
    ei = K.radius_graph(pos.to(DEV), 5.0, None, ptr_=ptr.to(DEV), max_beads_per_mol=n).cpu()
    assert ei.shape == (2, src.numel()) and torch.equal(ei[0], src) and torch.equal(ei[1], dst)
    params = so.params_from_npz(os.path.join(golden_dir, "weights_L3_R50.npz"))
    eng = engine.FusedSchNetEngine(S.model_from_params(params).to(DEV))
    types = torch.randint(1, 20, (n,), generator=g)
    eng.energy_forces(pos.to(DEV), types.to(DEV), ptr.to(DEV), n)
    assert int(eng.status[0]) == 0 and int(eng.status[2]) == src.numel() // 2


# ---------------------------------------------------------------------------------------- A2
@pytest.mark.parametrize("R", [50, 7, 64, 100])   # 100: more columns than one staging pass of the backward kernel (61)
def test_distance_rbf_forward_and_backward(K, R):
    pos, _, ptr, src, dst = _graph()
    centers, gamma = so.rbf_centers(R, 5.0)
    p64 = pos.double().requires_grad_(True)
    d_ref = so.compute_distances(p64, src, dst)
    rbf_ref = so.gaussian_rbf(d_ref, centers.double(), gamma, 5.0)
    g = torch.Generator().manual_seed(1)
    gd, gr = torch.randn(d_ref.shape, generator=g), torch.randn(rbf_ref.shape, generator=g)
    (gp_ref,) = torch.autograd.grad((d_ref * gd.double()).sum() + (rbf_ref * gr.double()).sum(), p64)

    pg = pos.to(DEV).requires_grad_(True)
    d, rbf = K.fused_distance_gaussian_rbf_cutoff_autograd(pg, src.to(DEV), dst.to(DEV),
                                                           centers.to(DEV), gamma, 5.0)
    assert rel(d.detach().cpu(), d_ref.detach()) < 1e-6
    assert rel(rbf.detach().cpu(), rbf_ref.detach()) < 2e-6
    (gp,) = torch.autograd.grad((d * gd.to(DEV)).sum() + (rbf * gr.to(DEV)).sum(), pg)
    assert rel(gp.cpu(), gp_ref) < 1e-5
    # the same operator with both CSR indices at hand: backward without float atomics (edge pass + by-node gather),
    # same values as the fp64 oracle and bit-reproducible from call to call
    n = pos.shape[0]
    dst_ptr, dst_perm = K.build_csr_index(dst.to(DEV), n)
    src_ptr, src_perm = K.build_src_csr_index(src.to(DEV), n)
    grads = []
    for _ in range(2):
        pc = pos.to(DEV).requires_grad_(True)
        d2, rbf2 = K.fused_distance_gaussian_rbf_cutoff_with_csr(pc, src.to(DEV), dst.to(DEV), centers.to(DEV), gamma,
                                                                 5.0, dst_ptr, dst_perm, src_ptr, src_perm)
        assert torch.equal(d2, d) and torch.equal(rbf2, rbf)
        grads.append(torch.autograd.grad((d2 * gd.to(DEV)).sum() + (rbf2 * gr.to(DEV)).sum(), pc)[0])
    assert rel(grads[0].cpu(), gp_ref) < 1e-5 and torch.equal(grads[0], grads[1])
    # zero edges: empty outputs (reference cfconv_kernels.py:1630-1631)
    e0 = torch.zeros(0, dtype=torch.int64, device=DEV)
    d0, r0 = K.fused_distance_gaussian_rbf_cutoff(pos.to(DEV), e0, e0, centers.to(DEV), gamma, 5.0)
    assert d0.shape == (0,) and r0.shape == (0, R)


# ---------------------------------------------------------------------------------------- A3
@pytest.mark.parametrize("shape", [(3, 37), (2, 269)])
def test_csr_build_bit_exact(K, shape):
    pos, types, ptr = so.synth_batch(*shape, 5.0)
    src, dst = so.radius_graph(pos, ptr, 5.0)
    n = pos.shape[0]
    for keys in (dst, src):  # dst: unsorted keys; src: sorted keys (identity fast path)
        ptr_ref, perm_ref = so.build_csr(keys, n)
        p, q = K._build_csr(keys.to(DEV), n)
        assert p.dtype == torch.int64 and q.dtype == torch.int64
        assert torch.equal(p.cpu(), ptr_ref) and torch.equal(q.cpu(), perm_ref)
    p, q, s = K.build_csr_index(dst.to(DEV), n, src.to(DEV))
    assert torch.equal(s.cpu(), src[so.build_csr(dst, n)[1]])


def test_csr_build_edge_cases(K):
    g = torch.Generator().manual_seed(5)
    keys = torch.randint(0, 7, (5000,), generator=g)  # long segments (> 64), empty nodes 7..19
    ptr_ref, perm_ref = so.build_csr(keys, 20)
    p, q = K.build_src_csr_index(keys.to(DEV), 20)
    assert torch.equal(p.cpu(), ptr_ref) and torch.equal(q.cpu(), perm_ref)
    p, q = K.build_csr_index(torch.zeros(0, dtype=torch.int64, device=DEV), 9)
    assert p.tolist() == [0] * 10 and q.numel() == 0


# ---------------------------------------------------------------------------------------- A5/A6
def _cfconv_ref(x, filt, d, src, dst, n, rc):
    W = filt * so.cosine_cutoff(d, rc).view(-1, 1)
    return torch.zeros(n, x.shape[1], dtype=x.dtype).index_add_(0, dst, x[src] * W)


@pytest.mark.parametrize("fdtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("variant", ["csr", "csr_nosrc", "atomic"])
def test_cfconv_forward_backward(K, fdtype, variant):
    pos, _, ptr, src, dst = _graph()
    n, F, rc = pos.shape[0], 128, 5.0
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, F, generator=g)
    filt = torch.randn(src.numel(), F, generator=g).to(fdtype)
    d = so.compute_distances(pos, src, dst)
    gout = torch.randn(n, F, generator=g)

    xr, fr, dr = (x.double().requires_grad_(True), filt.double().requires_grad_(True),
                  d.double().requires_grad_(True))
    out_ref = _cfconv_ref(xr, fr, dr, src, dst, n, rc)
    gx_ref, gf_ref, gd_ref = torch.autograd.grad((out_ref * gout.double()).sum(), (xr, fr, dr))

    xg, fg, dg = (x.to(DEV).requires_grad_(True), filt.to(DEV).requires_grad_(True),
                  d.to(DEV).requires_grad_(True))
    s, t = src.to(DEV), dst.to(DEV)
    dst_ptr, csr_perm = K.build_csr_index(t, n)
    src_ptr, src_perm = K.build_src_csr_index(s, n)
    if variant == "csr":
        out = K.fused_csr_cfconv_autograd(xg, fg, dg, s, t, dst_ptr, csr_perm, n, rc, src_ptr, src_perm)
    elif variant == "csr_nosrc":
        out = K.fused_csr_cfconv_autograd(xg, fg, dg, s, t, dst_ptr, csr_perm, n, rc)
    else:
        out = K.fused_cutoff_gather_multiply_scatter_autograd(xg, fg, dg, s, t, n, rc, src_ptr, src_perm)
    assert out.dtype == torch.float32
    assert rel(out.detach().cpu(), out_ref.detach()) < 2e-6
    gx, gf, gd = torch.autograd.grad((out * gout.to(DEV)).sum(), (xg, fg, dg))
    assert gf.dtype == fdtype  # grad_filter_out matches filter_out's dtype (reference csr_kernels.py:909)
    assert rel(gx.cpu(), gx_ref) < 2e-6
    assert rel(gf.float().cpu(), gf_ref) < (2e-6 if fdtype == torch.float32 else 1e-3)
    assert rel(gd.cpu(), gd_ref) < 1e-5  # the d(cutoff)/d(distance) term


def test_cfconv_odd_feature_width_and_segment_reduce(K):
    pos, _, ptr, src, dst = _graph(2, 20)
    n, F = pos.shape[0], 50
    g = torch.Generator().manual_seed(3)
    x, filt = torch.randn(n, F, generator=g), torch.randn(src.numel(), F, generator=g)
    d = so.compute_distances(pos, src, dst)
    ref = _cfconv_ref(x.double(), filt.double(), d.double(), src, dst, n, 5.0)
    dst_ptr, csr_perm = K.build_csr_index(dst.to(DEV), n)
    out = K.fused_csr_cfconv(x.to(DEV), filt.to(DEV), d.to(DEV), src.to(DEV), dst_ptr, csr_perm, n, 5.0)
    assert rel(out.cpu(), ref) < 2e-6
    for F2 in (128, 6):
        msg = torch.randn(src.numel(), F2, generator=g)
        ref2 = torch.zeros(n, F2).index_add_(0, dst, msg)
        out2 = K.csr_segment_reduce(msg.to(DEV), dst_ptr, csr_perm, n)
        assert rel(out2.cpu(), ref2) < 2e-6
    e0 = torch.zeros(0, dtype=torch.int64, device=DEV)
    z = K.fused_csr_cfconv(x.to(DEV), torch.zeros(0, F, device=DEV), torch.zeros(0, device=DEV), e0,
                           torch.zeros(n + 1, dtype=torch.int64, device=DEV), e0, n, 5.0)
    assert z.shape == (n, F) and float(z.abs().max()) == 0.0


# ---------------------------------------------------------------------------------------- A7, GEMMs
@pytest.mark.parametrize("m,k,n", [(999, 128, 128), (64, 50, 128), (130, 128, 1), (1, 7, 200)])
def test_gemm_f32_variants(K, m, k, n):
    g = torch.Generator().manual_seed(4)
    a, w, b = torch.randn(m, k, generator=g), torch.randn(k, n, generator=g) * 0.2, torch.randn(n, generator=g)
    A, W, B = a.to(DEV), w.to(DEV), b.to(DEV)
    ref = torch.tanh(a.double()) @ w.double() + b.double()
    assert rel(K.fused_tanh_linear(A, W, B).cpu(), ref) < 2e-6
    assert rel(K.fused_linear_tanh(A, W, B).cpu(), torch.tanh(a.double() @ w.double() + b.double())) < 2e-6
    assert rel(K.gemm_f32(A, W.t().contiguous(), B, b_is_nk=True).cpu(),
               a.double() @ w.double() + b.double()) < 2e-6
    # autograd of tanh-linear
    ar = a.double().requires_grad_(True)
    go = torch.randn(m, n, generator=g)
    (ga_ref,) = torch.autograd.grad(((torch.tanh(ar) @ w.double() + b.double()) * go.double()).sum(), ar)
    Ag = A.clone().requires_grad_(True)
    (ga,) = torch.autograd.grad((K.fused_tanh_linear_autograd(Ag, W, B) * go.to(DEV)).sum(), Ag)
    assert rel(ga.cpu(), ga_ref) < 2e-6


def _h(x):
    return x.half().float()


@pytest.mark.parametrize("m,k,n", [(1000, 50, 128), (257, 128, 128), (300, 64, 1)])
def test_w16a16_gemms(K, m, k, n):
    g = torch.Generator().manual_seed(6)
    x = torch.rand(m, k, generator=g)
    w = (torch.randn(k, n, generator=g) * 0.2).half()
    b = (torch.randn(n, generator=g) * 0.1).half()
    X, W, B = x.to(DEV), w.to(DEV), b.to(DEV)
    # K4: y = f16(tanh_c(f16(x) @ w + b))
    z = _h(x).double() @ w.double() + b.double()
    y_ref = so._tanh_w16(z)
    y = K.fused_linear_tanh_fp16(X, W, B)
    assert y.dtype == torch.float16
    assert float((y.float().cpu() - y_ref.float()).abs().max()) < 1.5e-3  # one fp16 ulp at |y|<=1
    y16 = K.fused_linear_tanh_fp16(X.half(), W, B)  # fp16 input accepted (cfconv_kernels.py:735)
    assert float((y16.float().cpu() - y_ref.float()).abs().max()) < 1.5e-3
    # K5
    xh = x.half()
    ref = xh.double() @ w.double()
    assert rel(K.linear_fp16(xh.to(DEV), W).cpu(), ref) < 1e-6
    assert rel(K.linear_fp16(xh.to(DEV), W, out_dtype=torch.float16).float().cpu(), ref) < 1e-3
    # K7 and K6 (w_t is [k_out, n_in]: reuse w as the pre-transposed operand)
    a = torch.randn(m, k, generator=g)
    assert rel(K.matmul_fp32_fp16_to_fp16_v2(a.to(DEV), W).float().cpu(), _h(a).double() @ w.double()) < 1e-3
    go, yy = torch.randn(m, k, generator=g).half(), (torch.rand(m, k, generator=g) * 2 - 1).half()
    gz = _h(go.float() * (1 - yy.float() ** 2))
    # g*(1-y^2) is formed by separately rounded fp32 operations, then rounded to fp16, as the reference's K6 does
    # (pinned by its interpreter-executed vector, tests/test_reference_kernel_vectors_gpu.py): the dot's inputs are
    # then identical fp16 values and only the fp32 accumulation order is left
    assert rel(K.fused_tanh_backward_grad_x_v2(go.to(DEV), yy.to(DEV), W).cpu(), gz.double() @ w.double()) < 2e-6


# ---------------------------------------------------------------------------------------- N1, (e)
def test_langevin_and_pt_swap(K):
    from flashmd_amd import simulate

    g = torch.Generator().manual_seed(8)
    n_mol, A = 6, 11
    n = n_mol * A
    x, v, f, noise = (torch.randn(n, 3, generator=g) for _ in range(4))
    masses = torch.rand(n, generator=g) + 0.5
    beta_mol = torch.tensor([1.67, 1.67, 1.42, 1.42, 1.16, 1.16])
    beta_node = beta_mol.repeat_interleave(A)
    x_ref, v_ref = so.langevin_baoab_pre(x, v, f, masses, noise, 0.004, 1.0, beta_node)
    X, V = x.to(DEV).clone(), v.to(DEV).clone()
    simulate.langevin_pre(X, V, f.to(DEV), masses.to(DEV), beta_node.to(DEV), noise.to(DEV), 0.004, 1.0)
    assert rel(X.cpu(), x_ref) < 1e-6 and rel(V.cpu(), v_ref) < 1e-6
    v2_ref = so.langevin_baoab_post(v_ref, f, masses, 0.004)
    simulate.langevin_post(V, f.to(DEV), masses.to(DEV), 0.004)
    assert rel(V.cpu(), v2_ref) < 1e-6
    # NVE velocity Verlet first half (the second half-kick is langevin_post)
    xv_ref, vv_ref = so.velocity_verlet_pre(x, v, f, masses, 0.004)
    Xv, Vv = x.to(DEV).clone(), v.to(DEV).clone()
    simulate.verlet_pre(Xv, Vv, f.to(DEV), masses.to(DEV), 0.004)
    assert rel(Xv.cpu(), xv_ref) < 1e-6 and rel(Vv.cpu(), vv_ref) < 1e-6
    # swap molecules (0,2) accepted, (1,3) rejected
    mol_a, mol_b = torch.tensor([0, 1]), torch.tensor([2, 3])
    acc = torch.tensor([1, 0], dtype=torch.uint8)
    xs, vs = x_ref.clone().view(n_mol, A, 3), v2_ref.clone().view(n_mol, A, 3)
    xe, ve = xs.clone(), vs.clone()
    xe[0], xe[2] = xs[2], xs[0]
    ve[0] = vs[2] * (beta_mol[0] / beta_mol[2]).sqrt()
    ve[2] = vs[0] * (beta_mol[2] / beta_mol[0]).sqrt()
    simulate.pt_swap_local(X, V, mol_a.to(DEV), mol_b.to(DEV), acc.to(DEV), beta_mol.to(DEV), A)
    assert rel(X.cpu().view(n_mol, A, 3), xe) < 1e-6 and rel(V.cpu().view(n_mol, A, 3), ve) < 1e-6
