"""CPU, build container only (skipped where /root/reference is absent, e.g. on the GPU box): the drop-in boundary
checked against the reference itself.
  1. every name of the reference's ``flashmd.kernels.__all__`` exists in ``flashmd_amd.kernels`` with the same
     ``inspect.signature`` (parameter names, order, defaults);
  2. INTEGRATION.md section 1 applied for real: the import block of the reference's ``models/schnet.py`` (and of
     ``models/gptq.py``) is rewritten to ``flashmd_amd.kernels`` and the module is imported -- the names resolve, the
     module reports the optimised operators as available, and its classes construct.
Nothing of the reference is copied: its source is read, patched in memory and executed."""
import importlib
import inspect
import os
import re
import sys
import types

import pytest
import torch

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "flashmd")), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref_kernels():
    sys.path.insert(0, REF)
    try:
        yield importlib.import_module("flashmd.kernels")
    finally:
        sys.path.remove(REF)


def _sig(obj):
    """(name, kind, default) of every parameter: what a positional or keyword call site depends on (annotations are
    documentation and differ in spelling: the reference annotates with classes, this package with strings)."""
    return [(p.name, p.kind, None if p.default is inspect.Parameter.empty else repr(p.default))
            for p in inspect.signature(obj).parameters.values()]


def test_signatures_of_all_exports_match(ref_kernels):
    import flashmd_amd.kernels as K

    assert len(ref_kernels.__all__) == 23
    for name in ref_kernels.__all__:
        assert hasattr(K, name), name
        ref_obj, our_obj = getattr(ref_kernels, name), getattr(K, name)
        if inspect.isfunction(ref_obj):
            assert inspect.isfunction(our_obj), name
            assert _sig(our_obj) == _sig(ref_obj), f"{name}: {_sig(our_obj)} != {_sig(ref_obj)}"
        else:  # the six *_kernel objects: Triton launchables there, C-ABI launch handles here
            assert name.endswith("_kernel") and not inspect.isfunction(our_obj), name
    # the names the reference's callers import without going through __all__ (models/schnet.py:28-48, gptq.py:18-27)
    for name in ("fused_cutoff_gather_multiply_scatter_autograd", "fused_distance_gaussian_rbf_cutoff_autograd",
                 "fused_tanh_linear_autograd", "build_csr_index", "build_src_csr_index", "fused_csr_cfconv_autograd",
                 "fused_linear_tanh_fp16_autograd", "linear_fp16_autograd"):
        assert _sig(getattr(K, name)) == _sig(getattr(ref_kernels, name)), name


def _forget(saved):
    """drop what the test imported from the reference tree and the stand-in wheels (nothing of torch's own)"""
    for k in list(sys.modules):
        if k not in saved and k.split(".")[0] in ("flashmd", "nvtx", "torch_geometric", "torch_cluster"):
            del sys.modules[k]


def _standins():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle.make_golden as mg

    mg._install_standins()


def _import_patched(module_name, rel_path, patch):
    path = os.path.join(REF, "flashmd", rel_path)
    src = patch(open(path).read())
    mod = types.ModuleType(module_name)
    mod.__file__, mod.__package__ = path, module_name.rpartition(".")[0]
    exec(compile(src, path, "exec"), mod.__dict__)
    return mod


def test_integration_import_swap_resolves(ref_kernels, monkeypatch):
    import flashmd_amd.kernels as K

    for k in ("MLCG_USE_TRITON_MESSAGE_PASSING", "MLCG_USE_FUSED_RBF", "MLCG_USE_FUSED_TANH_LINEAR", "MLCG_USE_CSR",
              "MLCG_USE_SRC_CSR_GRAD_X"):
        monkeypatch.setenv(k, "1")
    saved = dict(sys.modules)
    sys.path.insert(0, REF)
    try:
        _standins()

        def patch_schnet(src):  # INTEGRATION.md section 1, first diff
            src, n1 = re.subn(r"from flashmd\.kernels\.cfconv_kernels import \(", "from flashmd_amd.kernels import (", src)
            src, n2 = re.subn(r"\n    \)\n    from flashmd\.kernels\.csr_kernels import \(", "", src)
            assert n1 == 1 and n2 == 1
            return src

        m = _import_patched("flashmd.models._schnet_patched", "models/schnet.py", patch_schnet)
        assert m.TRITON_AVAILABLE is True
        for name in ("fused_cutoff_gather_multiply_scatter_autograd", "fused_distance_gaussian_rbf_cutoff_autograd",
                     "fused_tanh_linear_autograd", "build_csr_index", "build_src_csr_index",
                     "fused_csr_cfconv_autograd"):
            assert getattr(m, name) is getattr(K, name), name
        # the reference's classes still construct on top of the swapped operators
        from flashmd.models.cutoff import CosineCutoff
        from flashmd.models.radial_basis.gaussian import GaussianBasis

        net = m.StandardSchNet(GaussianBasis(CosineCutoff(0.0, 5.0), 20), CosineCutoff(0.0, 5.0), [32],
                               hidden_channels=128, num_filters=128, num_interactions=1)
        assert sum(p.numel() for p in net.parameters()) > 0

        def patch_gptq(src):  # second diff
            src, n = re.subn(r"from flashmd\.kernels import \(", "from flashmd_amd.kernels import (", src)
            assert n == 1
            return src

        g = _import_patched("flashmd.models._gptq_patched", "models/gptq.py", patch_gptq)
        assert g.TRITON_FP16_AVAILABLE is True
        assert g.fused_linear_tanh_fp16_autograd is K.fused_linear_tanh_fp16_autograd
        assert g.linear_fp16_autograd is K.linear_fp16_autograd
        fn = g.GPTQW16A16FilterNetwork(20, 128, 128)
        with pytest.raises(RuntimeError):  # the reference's own "no CPU fallback" guard still fires first
            fn(torch.zeros(4, 20))
    finally:
        sys.path.remove(REF)
        _forget(saved)


def test_state_dict_names_are_the_references(ref_kernels, monkeypatch):
    """Row N4: the mirror's parameter / buffer names and shapes are the reference StandardSchNet's, so a state_dict
    exported from the reference loads into ``flashmd_amd.schnet.model_from_state_dict`` (reference
    models/schnet.py:778-841)."""
    import flashmd_amd.schnet as S

    for k in ("MLCG_USE_TRITON_MESSAGE_PASSING", "MLCG_USE_FUSED_RBF", "MLCG_USE_FUSED_TANH_LINEAR", "MLCG_USE_CSR",
              "MLCG_USE_SRC_CSR_GRAD_X"):
        monkeypatch.setenv(k, "0")
    saved = dict(sys.modules)
    sys.path.insert(0, REF)
    try:
        _standins()
        import oracle.make_golden as mg

        ref = mg._reference_model(50, 3, 5.0, (128, 64))
        sd = ref.state_dict()
        mine = S.model_from_state_dict({k: v.clone() for k, v in sd.items()})
        ours = mine.state_dict()
        assert list(ours) == list(sd)
        for k in sd:
            assert ours[k].shape == sd[k].shape and torch.equal(ours[k], sd[k].to(ours[k].dtype)), k
    finally:
        sys.path.remove(REF)
        _forget(saved)
