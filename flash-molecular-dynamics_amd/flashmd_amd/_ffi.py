"""ctypes binding of the C ABI declared in include/flashmd_hip.h.

The shared library is built in-tree by ``__graft_entry__.build()`` (``make -C csrc``).  There is no
fallback: if the library is missing, or a tensor is not on a GPU, the operators raise.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libflashmd_hip.so")

P, I64, I32, F32 = c_void_p, c_int64, c_int, c_float

# name -> (restype, argtypes): one row per symbol in include/flashmd_hip.h
SIGNATURES = {
    "fmd_strerror": (ctypes.c_char_p, [I32]),
    "fmd_abi_version": (I32, []),
    "fmd_radius_graph_count": (I32, [P, P, I64, I64, I64, F32, P, P, P]),
    "fmd_radius_graph_fill": (I32, [P, P, I64, I64, I64, F32, P, P, P, P]),
    "fmd_distance_rbf_cutoff_fwd": (I32, [P, P, P, P, F32, F32, I64, I64, P, P, P]),
    "fmd_distance_rbf_cutoff_bwd": (I32, [P, P, P, P, F32, F32, I64, I64, I64, P, P, P, P, P]),
    "fmd_distance_rbf_cutoff_bwd_csr": (I32, [P, P, P, P, F32, F32, I64, I64, I64, P, P, P, P, P, P, P, P, P, P]),
    "fmd_build_csr_workspace_bytes": (I64, [I64, I64]),
    "fmd_build_csr": (I32, [P, I64, I64, P, P, P, P]),
    "fmd_csr_cfconv_fwd": (I32, [P, P, I32, P, P, P, P, I64, I64, F32, P, P]),
    "fmd_cfconv_scatter_atomic": (I32, [P, P, I32, P, P, P, I64, I64, I64, F32, P, P]),
    "fmd_cfconv_grad_edge": (I32, [P, P, P, I32, P, P, P, I64, I64, F32, P, P, P]),
    "fmd_csr_segment_reduce": (I32, [P, P, P, I64, I64, P, P]),
    "fmd_gemm_f32": (I32, [P, P, P, P, P, I64, I64, I64, I32, I32, I32, P]),
    "fmd_linear_tanh_f16": (I32, [P, I32, P, P, P, I64, I64, I64, P]),
    "fmd_linear_f16": (I32, [P, P, P, I32, I64, I64, I64, P]),
    "fmd_tanh_bwd_grad_x_f16": (I32, [P, I32, P, P, P, I64, I64, I64, P]),
    "fmd_matmul_f32_f16_to_f16": (I32, [P, P, P, I64, I64, I64, P]),
    "fmd_schnet_pack_bytes": (I64, [P]),
    "fmd_schnet_pack": (I32, [P, P, P, P]),
    "fmd_schnet_workspace_bytes": (I64, [P, I64, I64, I64]),
    "fmd_schnet_energy_forces": (I32, [P, P, P, P, P, I64, I64, I64, I64, P, P, P, P, P]),
    "fmd_schnet_kernel_count": (I32, [P]),
    "fmd_schnet_kernel_name": (ctypes.c_char_p, [P, I32]),
    "fmd_schnet_energy_forces_timed": (I32, [P, P, P, P, P, I64, I64, I64, I64, P, P, P, P, P, P, I64]),
    "fmd_langevin_pre": (I32, [P, P, P, P, P, P, I64, F32, F32, F32, P]),
    "fmd_langevin_post": (I32, [P, P, P, I64, F32, P]),
    "fmd_verlet_pre": (I32, [P, P, P, P, I64, F32, P]),
    "fmd_overdamped_step": (I32, [P, P, P, P, I64, F32, F32, P]),
    "fmd_prior_energy_forces": (I32, [P, P, I64, I64, I32, P, P, I32, P, P, I64, P, P, P, P]),
    "fmd_prior_pairs_energy_forces": (I32, [P, P, I64, I64, I32, P, P, P, I64, P, P, P, P]),
    "fmd_pt_swap_local": (I32, [P, P, P, P, P, P, I64, I64, P]),
}



class SchnetDesc(ctypes.Structure):
    """fmd_schnet_desc (include/flashmd_hip.h)"""
    _fields_ = [("hidden", ctypes.c_int32), ("num_rbf", ctypes.c_int32), ("num_blocks", ctypes.c_int32),
                ("num_out", ctypes.c_int32), ("out_width", ctypes.c_int32 * 4),
                ("embedding_size", ctypes.c_int32), ("precision", ctypes.c_int32),
                ("n_split", ctypes.c_int32), ("cutoff", ctypes.c_float), ("gamma", ctypes.c_float),
                ("flags", ctypes.c_int32)]


FLAG_NO_FILTER_STASH = 1   # FMD_FLAG_NO_FILTER_STASH
FLAG_NO_MOLECULE_GROUPS = 2   # FMD_FLAG_NO_MOLECULE_GROUPS
ABI_VERSION = 6            # what this binding (and engine.py's handling of the status words) was written for


class SchnetBlockWeights(ctypes.Structure):
    """fmd_schnet_block_weights: host pointers, nn.Linear [out,in] layout"""
    _fields_ = [(n, c_void_p) for n in ("lin1_w", "f0_w", "f0_b", "f1_w", "lin2_w", "lin2_b", "lin_w",
                                        "lin_b")]


class SchnetWeights(ctypes.Structure):
    """fmd_schnet_weights"""
    _fields_ = [("embedding", c_void_p), ("blocks", ctypes.POINTER(SchnetBlockWeights)),
                ("out_w", c_void_p * 4), ("out_b", c_void_p * 4)]


_lib = None


def lib() -> ctypes.CDLL:
    """Load (once) and return the library; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"flashmd_amd: {LIB_PATH} is missing -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                "There is no CPU fallback for the HIP operators.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype, fn.argtypes = res, args
        got = handle.fmd_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(
                f"flashmd_amd: {LIB_PATH} implements ABI {got}, this package expects ABI {ABI_VERSION} (same signatures, "
                "different status-word and workspace rules): rebuild it with "
                "`python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = handle
    return _lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def strerror(rc: int) -> str:
    return lib().fmd_strerror(int(rc)).decode()


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"flashmd_amd.{what}: {lib().fmd_strerror(rc).decode()} (code {rc})")


def call(name: str, *args):
    check(getattr(lib(), name)(*args), name)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "flashmd_amd HIP operators need tensors on an AMD GPU (got a CPU tensor); "
                "set the MLCG_USE_* gates to 0 / pass --disable_optim for the PyTorch path")
