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


@register_qmodule(torch.nn.Conv2d)
class QConv2d(QModuleMixin, torch.nn.Conv2d):
    @classmethod
    def qcreate(cls, module, weights: qtype, activations: Optional[qtype] = None, optimizer: Optional[Optimizer] = None,
                device: Optional[torch.device] = None):
        return cls(in_channels=module.in_channels, out_channels=module.out_channels, kernel_size=module.kernel_size,
                   stride=module.stride, padding=module.padding, dilation=module.dilation, groups=module.groups,
                   bias=module.bias is not None, padding_mode=module.padding_mode, dtype=module.weight.dtype, device=device,
                   weights=weights, activations=activations, optimizer=optimizer)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return self._conv_forward(input, self.qweight, self.bias)
