"""W16A16 module swap for the filter and output networks (reference src/flashmd/models/gptq.py).

"GPTQ W16A16" in the reference is a plain fp16 cast of ``Linear.weight.t()`` plus fp16-storage /
fp32-accumulate kernels; the same here, on the MFMA fp16 kernels of csrc/fmd_gemm.hip.  As in the
reference there is no CPU path: the modules raise on CPU input (gptq.py:115-119).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .kernels import fused_linear_tanh_fp16_autograd, linear_fp16_autograd


def _linears(mlp: nn.Module):
    return [m for m in mlp.layers if isinstance(m, nn.Linear)]


class GPTQW16A16FilterNetwork(nn.Module):
    """Linear+tanh (fp16) -> Linear (fp16 out)  (reference gptq.py:30-189)."""

    def __init__(self, in_features: int, hidden_features: int, out_features: int):
        super().__init__()
        self.in_features, self.hidden_features, self.out_features = in_features, hidden_features, out_features
        self.weight0 = nn.Parameter(torch.empty(in_features, hidden_features, dtype=torch.float16))
        self.bias0 = nn.Parameter(torch.empty(hidden_features, dtype=torch.float16))
        self.weight1 = nn.Parameter(torch.empty(hidden_features, out_features, dtype=torch.float16))
        nn.init.xavier_uniform_(self.weight0)
        nn.init.zeros_(self.bias0)
        nn.init.xavier_uniform_(self.weight1)

    def forward(self, x: torch.Tensor, data=None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("GPTQ W16A16 filter network requires a GPU input. "
                               "No CPU fallback is provided to prevent silent bugs.")
        x = fused_linear_tanh_fp16_autograd(x.contiguous(), self.weight0, self.bias0)
        return linear_fp16_autograd(x, self.weight1, out_dtype=torch.float16)

    @classmethod
    def from_mlp(cls, mlp: nn.Module) -> "GPTQW16A16FilterNetwork":
        l0, l1 = _linears(mlp)[0], _linears(mlp)[-1]
        q = cls(l0.in_features, l0.out_features, l1.out_features)
        with torch.no_grad():  # Linear weight is [out,in]; the kernels take [in,out] (gptq.py:175-184)
            q.weight0.copy_(l0.weight.t().half())
            q.bias0.copy_(l0.bias.half()) if l0.bias is not None else q.bias0.zero_()
            q.weight1.copy_(l1.weight.t().half())
        return q.to(l0.weight.device)


class GPTQW16A16OutputNetwork(nn.Module):
    """Linear+tanh, Linear+tanh (fp16) -> Linear (fp32 out)  (reference gptq.py:192-371)."""

    def __init__(self, in_features, hidden1_features, hidden2_features, out_features):
        super().__init__()
        self.weight0 = nn.Parameter(torch.empty(in_features, hidden1_features, dtype=torch.float16))
        self.bias0 = nn.Parameter(torch.empty(hidden1_features, dtype=torch.float16))
        self.weight1 = nn.Parameter(torch.empty(hidden1_features, hidden2_features, dtype=torch.float16))
        self.bias1 = nn.Parameter(torch.empty(hidden2_features, dtype=torch.float16))
        self.weight2 = nn.Parameter(torch.empty(hidden2_features, out_features, dtype=torch.float16))
        for w in (self.weight0, self.weight1, self.weight2):
            nn.init.xavier_uniform_(w)
        nn.init.zeros_(self.bias0)
        nn.init.zeros_(self.bias1)

    def forward(self, x: torch.Tensor, data=None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("GPTQ W16A16 output network requires a GPU input. "
                               "No CPU fallback is provided to prevent silent bugs.")
        x = fused_linear_tanh_fp16_autograd(x.contiguous(), self.weight0, self.bias0)
        x = fused_linear_tanh_fp16_autograd(x, self.weight1, self.bias1)
        return linear_fp16_autograd(x, self.weight2)

    @classmethod
    def from_mlp(cls, mlp: nn.Module) -> "GPTQW16A16OutputNetwork":
        l0, l1, l2 = _linears(mlp)
        q = cls(l0.in_features, l0.out_features, l1.out_features, l2.out_features)
        with torch.no_grad():
            q.weight0.copy_(l0.weight.t().half())
            q.bias0.copy_(l0.bias.half()) if l0.bias is not None else q.bias0.zero_()
            q.weight1.copy_(l1.weight.t().half())
            q.bias1.copy_(l1.bias.half()) if l1.bias is not None else q.bias1.zero_()
            q.weight2.copy_(l2.weight.t().half())
        return q.to(l0.weight.device)


def apply_gptq_w16a16_to_model(model: nn.Module) -> nn.Module:
    """Swap every CFConv.filter_network and 2/3-linear output_network MLP for its W16A16 twin
    (reference gptq.py:374-443)."""
    filter_count = output_count = 0
    for _, module in model.named_modules():
        if hasattr(module, "filter_network") and hasattr(module.filter_network, "layers"):
            module.filter_network = GPTQW16A16FilterNetwork.from_mlp(module.filter_network)
            filter_count += 1
    for _, module in model.named_modules():
        if hasattr(module, "output_network") and hasattr(module.output_network, "layers"):
            n_lin = len(_linears(module.output_network))
            if n_lin == 3:
                module.output_network = GPTQW16A16OutputNetwork.from_mlp(module.output_network)
                output_count += 1
            elif n_lin == 2:
                module.output_network = GPTQW16A16FilterNetwork.from_mlp(module.output_network)
                output_count += 1
    if filter_count == 0:
        raise RuntimeError("No filter networks found to quantize. "
                           "Ensure the model contains CFConv layers with filter_network MLPs.")
    print(f"[GPTQ W16A16] Quantized {filter_count} filter networks, {output_count} output networks")
    return model


def validate_gptq_w16a16(model: nn.Module) -> bool:
    """reference gptq.py:446-484"""
    for name, module in model.named_modules():
        if hasattr(module, "filter_network") and not isinstance(module.filter_network,
                                                                GPTQW16A16FilterNetwork):
            raise RuntimeError(f"Filter network in {name} is not quantized! Got "
                               f"{type(module.filter_network).__name__} instead of "
                               "GPTQW16A16FilterNetwork. GPTQ quantization failed to apply.")
    return True
