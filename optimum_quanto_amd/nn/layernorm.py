"""``QLayerNorm`` (optimum/quanto/nn/qlayernorm.py:26-53): never quantizes its weights; exists so that a model with
quantized activations gets its LayerNorm outputs quantized (``quantize_output`` hook of ``QModuleMixin``)."""
from typing import Optional

import torch

from ..tensor import Optimizer, qtype
from .module import QModuleMixin, register_qmodule

__all__ = ["QLayerNorm"]


@register_qmodule(torch.nn.LayerNorm)
class QLayerNorm(QModuleMixin, torch.nn.LayerNorm):
    @classmethod
    def qcreate(cls, module, weights: Optional[qtype] = None, activations: Optional[qtype] = None,
                optimizer: Optional[Optimizer] = None, device: Optional[torch.device] = None):
        if activations is None:
            return None  # nothing to do: the normalisation weights are never quantized, only the outputs are
        affine = module.elementwise_affine
        return cls(module.normalized_shape, eps=module.eps, elementwise_affine=affine, bias=module.bias is not None,
                   dtype=module.weight.dtype if affine else None, device=device, weights=None, activations=activations,
                   optimizer=None)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return torch.nn.functional.layer_norm(input, self.normalized_shape, self.weight, self.bias, self.eps)
