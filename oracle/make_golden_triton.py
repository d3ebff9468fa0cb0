"""Generate tests/golden/triton_*.npz by EXECUTING the reference's own Triton kernels on the CPU
(build container only; TEST INFRASTRUCTURE, never imported by the product package).

    python -O oracle/make_golden_triton.py            (re-launches itself with -O when run without)

How the reference's GPU-only code runs here, with no line of it changed or copied:
  * ``TRITON_INTERPRET=1`` (set below, before triton is imported): every ``@triton.jit`` function of
    /root/reference/src/flashmd/kernels becomes an ``InterpretedFunction`` that executes the kernel
    body program by program on numpy arrays (triton 3.6.0, ``tl.dot`` = ``np.matmul(..., dtype=f32)``).
  * ``python -O``: the reference's host wrappers guard their inputs with ``assert x.is_cuda``
    (kernels/cfconv_kernels.py:134-138,735-738,859-863; csr_kernels.py:441-446,770-775); -O drops
    asserts, so the wrappers (output allocation, grid, strides) and the autograd Functions that chain
    the kernels (FusedLinearTanhFP16Function :1305-1361, LinearFP16ToFP16Function :1402-1434,
    LinearFP16Function :1364-1399) run as they are.
  * ``triton.autotune`` would time its configurations on a GPU; each Autotuner is pinned here to ONE of
    the configurations the reference lists (``kernel.configs = [that one]``) -- the block sizes are
    recorded in the fixture.  BLOCK_K only fixes the association of the fp32 partial sums.
  * the two GPTQ modules refuse CPU input with a RuntimeError (models/gptq.py:115-119,289-293): a
    forward-pre-hook hands them the same tensor viewed as a ``torch.Tensor`` subclass whose
    ``is_cuda`` reads True (``Tensor.as_subclass``: same storage, same autograd graph).

What is produced:
  1. ``triton_stages_1x37.npz`` -- for molecule 0 of the small case (37 beads, weights_L3_R50): inputs and the
     reference kernels' outputs, stage by stage: K14 distance+RBF+cutoff forward, K3 fp32
     linear+tanh, K4 / K5 (f16 and f32 out) / K6 / K7 (the W16A16 GEMMs of SURVEY row A9), K15
     tanh+linear, K1 atomic CFConv, K9 CSR CFConv (f32 and f16 filter), K10 src-CSR grad_x, K2
     grad_filter_out (f32 and f16), K11/K12 CSR build.
  2. ``case_w16a16_ref.npz`` -- BASELINE config 3 end to end: the reference's ``StandardSchNet`` after its
     own ``apply_gptq_w16a16_to_model`` (models/gptq.py:374-443), evaluated by its own ``GradientsOut``
     with the MLCG_USE_* gates "0" (PyTorch CFConv branch = the gradient of its own energy):
     filter and output networks run K4/K5 forward and K6/K7 backward in the interpreter.  Energies and
     forces for the three golden cases.
Everything written is data (inputs and outputs); this script is the provenance.
"""
import json
import os
import subprocess
import sys

if __debug__ and __name__ == "__main__":
    # the asserts of the reference's host wrappers must be compiled out: run again with -O
    sys.exit(subprocess.call([sys.executable, "-O", os.path.abspath(__file__)] + sys.argv[1:]))

os.environ["TRITON_INTERPRET"] = "1"
for _k in ("MLCG_USE_TRITON_MESSAGE_PASSING", "MLCG_USE_FUSED_RBF", "MLCG_USE_FUSED_TANH_LINEAR",
           "MLCG_USE_CSR", "MLCG_USE_SRC_CSR_GRAD_X"):
    os.environ[_k] = "0"

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

PINNED = {}
_LISTED = {}


def pin(module, name, **meta):
    """Restrict an Autotuner to the one listed configuration with these block sizes."""
    kern = getattr(module, name)
    listed = _LISTED.setdefault(name, list(kern.configs))
    for c in listed:
        if all(c.kwargs.get(k) == v for k, v in meta.items()):
            kern.configs = [c]
            PINNED[name] = dict(c.kwargs)
            return
    raise KeyError((name, meta))


class HostAsDevice(torch.Tensor):
    """A CPU tensor whose ``is_cuda`` reads True (for the GPTQ modules' device check only)."""

    @property
    def is_cuda(self):
        return True


def _pre(_mod, args):
    return (args[0].as_subclass(HostAsDevice),) + tuple(args[1:])


def _post(_mod, _args, out):
    return out.as_subclass(torch.Tensor)


def np_(t):
    return t.detach().cpu().numpy()


def stage_vectors(ck, cs, so, p):
    """Molecule 0 of the small case (37 beads, its directed edges); the GEMM stages that feed nothing
    else use the first MG edge rows to keep the fixture small."""
    z = np.load(os.path.join(GOLD, "case_small_3x37.npz"))
    n = int(z["ptr"][1])
    keep = torch.from_numpy(z["edge_src"]) < n
    pos = torch.from_numpy(z["pos"])[:n].contiguous()
    src = torch.from_numpy(z["edge_src"])[keep].contiguous()
    dst = torch.from_numpy(z["edge_dst"])[keep].contiguous()
    e, MG = src.numel(), 400
    rc = p["cutoff"]
    centers, gamma = so.rbf_centers(p["num_rbf"], rc)
    blk = p["blocks"][0]
    out = {"pos": np_(pos), "edge_src": np_(src), "edge_dst": np_(dst), "cutoff": np.array([rc]),
           "gamma": np.array([gamma]), "centers": np_(centers), "mg": np.array([MG])}

    # K14 forward (cfconv_kernels.py:1470-1656)
    d, rbf = ck.fused_distance_gaussian_rbf_cutoff(pos, src, dst, centers.contiguous(), gamma, rc)
    out["k14_d"], out["k14_rbf"] = np_(d), np_(rbf)

    # K3 fp32 linear+tanh (:457-593), weight [K,N] = Linear.weight.t(); input = k14_rbf[:MG]
    w0 = blk["f0_w"].t().contiguous()
    out["k3_w"], out["k3_b"] = np_(w0), np_(blk["f0_b"])
    out["k3_y"] = np_(ck.fused_linear_tanh(rbf[:MG].contiguous(), w0, blk["f0_b"]))

    # K4 / K5: the W16A16 filter network forward (gptq.py:92-130); input = k14_rbf, then k4_y
    w0h, b0h = w0.half(), blk["f0_b"].half()
    w1h = blk["f1_w"].t().contiguous().half()
    hid = ck.fused_linear_tanh_fp16(rbf, w0h, b0h)
    f16 = ck.linear_fp16(hid, w1h, out_dtype=torch.float16)
    f32 = ck.linear_fp16(hid, w1h, out_dtype=torch.float32)
    out.update(k4_w=np_(w0h), k4_b=np_(b0h), k4_y=np_(hid), k5_w=np_(w1h), k5_y16=np_(f16), k5_y32=np_(f32))
    # K4 with an fp16 input (second layer of the output network, gptq.py:301); input = k4_y[:n]
    o1h, c1h = p["out"][1]["w"].t().contiguous().half(), p["out"][1]["b"].half()
    out.update(k4h_w=np_(o1h), k4h_b=np_(c1h),
               k4h_y=np_(ck.fused_linear_tanh_fp16(hid[:n].contiguous(), o1h, c1h)))

    # K6 / K7: their backward (cfconv_kernels.py:973-1064, 1165-1227); y = k4_y[:MG]
    g = torch.Generator().manual_seed(21)
    g16 = (torch.randn(MG, 128, generator=g) * 0.05).half()
    w0h_t = w0h.t().contiguous()  # [N_out=128, K_in=50]
    out.update(k6_g=np_(g16), k6_wt=np_(w0h_t),
               k6_gx=np_(ck.fused_tanh_backward_grad_x_v2(g16, hid[:MG].contiguous(), w0h_t)))
    a32 = torch.randn(MG, 128, generator=g) * 0.05
    w1h_t = w1h.t().contiguous()
    out.update(k7_a=np_(a32), k7_bt=np_(w1h_t), k7_c=np_(ck.matmul_fp32_fp16_to_fp16_v2(a32, w1h_t)))

    # K15 tanh+linear (:1758-1898)
    xn = torch.randn(n, 128, generator=g)
    wl = blk["lin_w"].t().contiguous()
    out.update(k15_x=np_(xn), k15_w=np_(wl), k15_b=np_(blk["lin_b"]),
               k15_y=np_(ck.fused_tanh_linear(xn, wl, blk["lin_b"])))

    # K11 / K12 CSR build (csr_kernels.py:88-169, 229-294)
    dst_ptr, perm = cs.build_csr_index(dst, n)
    src_ptr, sperm = cs.build_src_csr_index(src, n)
    out.update(k11_ptr=np_(dst_ptr), k11_perm=np_(perm), k12_ptr=np_(src_ptr), k12_perm=np_(sperm))

    # K1 / K9 CFConv forward, K10 / K2 backward (cfconv_kernels.py:93-170, 261-337; csr_kernels.py:399-482,
    # 727-815); filters = k5_y32 / k5_y16, distances = k14_d
    x1 = torch.randn(n, 128, generator=g)
    gout = torch.randn(n, 128, generator=g)
    out.update(cf_x=np_(x1), cf_g=np_(gout))
    out["k1_out32"] = np_(ck.fused_cutoff_gather_multiply_scatter(x1, f32, d, src, dst, n, rc))
    for tag, filt in (("32", f32), ("16", f16)):
        out["k9_out" + tag] = np_(cs.fused_csr_cfconv(x1, filt, d, src, dst_ptr, perm, n, rc))
        out["k10_gx" + tag] = np_(cs.fused_src_csr_grad_x(gout, filt, d, dst, src_ptr, sperm, n, rc))
        out["k2_gf" + tag] = np_(ck.fused_grad_filter_out(x1, gout, d, src, dst, rc, out_dtype=filt.dtype))
    return out


def w16a16_end_to_end(mg, so, p, tag=""):
    """Reference StandardSchNet + its own GPTQ swap + GradientsOut on the three golden cases."""
    from flashmd.models import gptq

    res = {}
    model = mg._reference_model(p["num_rbf"], len(p["blocks"]), p["cutoff"], (128, 64))
    # same weights as tests/golden/weights_L3_R50.npz (make_golden.py perturbs the biases with seed 11)
    p_chk = mg._params_from_reference(model, p["cutoff"], bias_seed=11)
    for a, b in zip(p_chk["blocks"], p["blocks"]):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    gptq.apply_gptq_w16a16_to_model(model)
    gptq.validate_gptq_w16a16(model)
    for m in model.modules():
        if isinstance(m, (gptq.GPTQW16A16FilterNetwork, gptq.GPTQW16A16OutputNetwork)):
            m.register_forward_pre_hook(_pre)
            m.register_forward_hook(_post)
    for case in ("case_small_3x37", "case_1enh_2x269", "case_ragged"):
        z = np.load(os.path.join(GOLD, case + ".npz"))
        pos, types_, ptr = (torch.from_numpy(z[k]) for k in ("pos", "types", "ptr"))
        e, f = mg._run_reference(model, pos, types_, ptr, torch.float32)
        e64, f64 = z["energy_ref64"], z["forces_ref64"]
        de = float(np.abs(np_(e).astype(np.float64) - e64).max() / np.abs(e64).max())
        df = float(np.linalg.norm(np_(f).astype(np.float64) - f64) / np.linalg.norm(f64))
        oe, of = so.schnet_energy_forces(p, pos, types_, ptr, w16a16=True)
        oe_ = float((oe - e).abs().max() / e.abs().max())
        of_ = float((of - f).norm() / f.norm())
        print(f"[{case}] reference W16A16 vs reference fp64: energy {de:.2e} forces {df:.2e}; "
              f"oracle(w16a16) vs reference W16A16: energy {oe_:.2e} forces {of_:.2e}")
        res[case + "_energy" + tag] = np_(e)
        res[case + "_forces" + tag] = np_(f)
    return res


def main():
    import oracle.make_golden as mg
    from oracle import schnet_oracle as so

    mg._install_standins()
    import flashmd.kernels.cfconv_kernels as ck
    import flashmd.kernels.csr_kernels as cs

    # one listed configuration per autotuned kernel (cfconv_kernels.py:457-467,644-656,766-778,896-906,
    # 963-972,1165-1172,1470-1479,1758-1768)
    pin(ck, "fused_linear_tanh_kernel", BLOCK_M=64, BLOCK_N=64, BLOCK_K=32)
    pin(ck, "fused_linear_tanh_fp16_kernel", BLOCK_M=64, BLOCK_N=128, BLOCK_K=64)
    pin(ck, "linear_fp16_kernel", BLOCK_M=64, BLOCK_N=128, BLOCK_K=32)
    pin(ck, "linear_fp16_to_fp16_kernel", BLOCK_M=64, BLOCK_N=128, BLOCK_K=32)
    pin(ck, "fused_tanh_backward_grad_x_v2_kernel", BLOCK_M=128, BLOCK_N=64, BLOCK_K=128)
    pin(ck, "matmul_fp32_fp16_to_fp16_v2_kernel", BLOCK_M=128, BLOCK_N=128, BLOCK_K=128)
    for name in ("fused_distance_gaussian_rbf_cutoff_kernel", "fused_tanh_linear_kernel"):
        kern = getattr(ck, name)
        kern.configs = [kern.configs[0]]
        PINNED[name] = dict(kern.configs[0].kwargs)

    p = so.params_from_npz(os.path.join(GOLD, "weights_L3_R50.npz"))
    st = stage_vectors(ck, cs, so, p)
    st["pinned_configs"] = np.array(json.dumps(PINNED))
    np.savez_compressed(os.path.join(GOLD, "triton_stages_1x37.npz"), **st)
    print("stage vectors:", {k: v.shape for k, v in st.items() if k.startswith("k")})

    if "--stages-only" in sys.argv:
        return
    res = w16a16_end_to_end(mg, so, p)
    res["pinned_configs"] = np.array(json.dumps(PINNED))
    # The same evaluation with ANOTHER of the reference's listed configurations for every GEMM whose
    # list offers a different BLOCK_K (the association of the fp32 partial sums changes, nothing else):
    # the distance between the two runs is the spread of the reference's own W16A16 result.
    pin(ck, "linear_fp16_kernel", BLOCK_M=32, BLOCK_N=128, BLOCK_K=128)
    pin(ck, "linear_fp16_to_fp16_kernel", BLOCK_M=32, BLOCK_N=128, BLOCK_K=128)
    pin(ck, "fused_tanh_backward_grad_x_v2_kernel", BLOCK_M=128, BLOCK_N=64, BLOCK_K=64)
    pin(ck, "matmul_fp32_fp16_to_fp16_v2_kernel", BLOCK_M=128, BLOCK_N=128, BLOCK_K=64)
    res.update(w16a16_end_to_end(mg, so, p, tag="_cfgB"))
    res["pinned_configs_cfgB"] = np.array(json.dumps(PINNED))
    for case in ("case_small_3x37", "case_1enh_2x269", "case_ragged"):
        fa, fb = res[case + "_forces"], res[case + "_forces_cfgB"]
        ea, eb = res[case + "_energy"], res[case + "_energy_cfgB"]
        print(f"[{case}] reference W16A16, two listed autotune configurations: energy "
              f"{np.abs(ea - eb).max() / np.abs(ea).max():.2e} forces "
              f"{np.linalg.norm(fa - fb) / np.linalg.norm(fa):.2e}")
    np.savez_compressed(os.path.join(GOLD, "case_w16a16_ref.npz"), **res)
    print("written to", GOLD)


if __name__ == "__main__":
    main()
