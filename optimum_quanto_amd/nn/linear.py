"""``QLinear`` - the entry point of the hot path (optimum/quanto/nn/qlinear.py:26-50)."""
from typing import Optional

import torch

from ..tensor import Optimizer, QTensor, qtype
from .module import QModuleMixin, register_qmodule

__all__ = ["QLinear"]


@register_qmodule(torch.nn.Linear)
class QLinear(QModuleMixin, torch.nn.Linear):
    @classmethod
    def qcreate(cls, module, weights: qtype, activations: Optional[qtype] = None, optimizer: Optional[Optimizer] = None,
                device: Optional[torch.device] = None):
        return cls(module.in_features, module.out_features, module.bias is not None, dtype=module.weight.dtype,
                   device=device, weights=weights, activations=activations, optimizer=optimizer, quantize_input=True)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        # F.linear is intercepted by the weight's __torch_function__ and lands on quanto::qbytes_mm / quanto::qbits_mm.  For a plain
        # activation tensor the weight class's handler is called directly: the same code path minus torch's override dispatch
        # (C++ -> handle_torch_function -> Python, ~2 us of the ~15 us a decode-shaped call costs on the host)
        w = self.qweight
        if type(input) is torch.Tensor and isinstance(w, QTensor):
            return type(w).__torch_function__(torch.nn.functional.linear, (type(w),), (input, w, self.bias))
        return torch.nn.functional.linear(input, w, bias=self.bias)
