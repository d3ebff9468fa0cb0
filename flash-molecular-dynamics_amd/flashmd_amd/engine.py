"""Fused whole-path engine (energy + forces of a batch in one HIP pipeline).  Placeholder until the
fused kernels land: qualifies no model, so GradientsOut takes the operator path."""
from __future__ import annotations


def maybe_engine(model, data):
    if not data.pos.is_cuda:
        raise RuntimeError(
            "flashmd_amd: the HIP gates are on but the batch is on the CPU. There is no CPU fallback "
            "for the optimized path; run with --disable_optim (all MLCG_USE_* = 0).")
    return None
