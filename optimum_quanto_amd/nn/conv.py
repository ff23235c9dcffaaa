"""``QConv2d`` (optimum/quanto/nn/qconv2d.py:26-55).

The forward is the reference's: ``_conv_forward(input, self.qweight, self.bias)``, i.e. ``F.conv2d`` with a quantized
weight.  In the reference that call falls back to "dequantize the weight, run the float convolution".  Here the weight
tensors intercept ``F.conv2d`` (tensor/weights.py, ``conv2d_as_gemm``): on a ROCm device a dense (``groups == 1``)
convolution is lowered to im2col + the same fused ``quanto::qbytes_mm`` / ``quanto::qbits_mm`` kernels that serve QLinear -
the [N, C, kh, kw] weight *is* the [N, C*kh*kw] GEMM operand, byte for byte, in both storage formats.  Everything else
(grouped convolutions, CPU tensors) keeps the reference behaviour.
"""
from typing import Optional

import torch

from ..tensor import Optimizer, qtype
from .module import QModuleMixin, register_qmodule

__all__ = ["QConv2d"]


# constructor arguments that describe the convolution geometry, copied verbatim from the float module
_GEOMETRY = ("in_channels", "out_channels", "kernel_size", "stride", "padding", "dilation", "groups", "padding_mode")


@register_qmodule(torch.nn.Conv2d)
class QConv2d(QModuleMixin, torch.nn.Conv2d):
    @classmethod
    def qcreate(cls, module, weights: qtype, activations: Optional[qtype] = None, optimizer: Optional[Optimizer] = None,
                device: Optional[torch.device] = None):
        geometry = {name: getattr(module, name) for name in _GEOMETRY}
        return cls(**geometry, bias=module.bias is not None, dtype=module.weight.dtype, device=device, weights=weights,
                   activations=activations, optimizer=optimizer)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        # F.conv2d is intercepted by the weight's __torch_function__ (im2col + fused GEMM on a ROCm device)
        return self._conv_forward(input, self.qweight, self.bias)
