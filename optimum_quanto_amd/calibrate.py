"""Activation calibration: ``Calibration`` and ``absmax_scale`` (optimum/quanto/calibrate.py:28-189).

``with Calibration(): model(samples)`` records, for every quantized module with quantized activations, the per-tensor
absmax scale of its input and output (momentum-averaged over batches) into the ``input_scale`` / ``output_scale``
buffers that ``QModuleMixin.quantize_input`` / ``quantize_output`` use afterwards.  With ``streamline`` the mode also
watches which torch functions consume each module's quantized output: when none of them keeps it quantized, the
module's output quantization is removed (it would be dequantized immediately anyway).
"""
from typing import Optional

import torch
from torch.nn.modules.module import register_module_forward_hook, register_module_forward_pre_hook
from torch.overrides import TorchFunctionMode

from .nn import QModuleMixin
from .tensor import ActivationQBytesTensor, QTensor, axis_to_dim, dtype_info, qint8, qtype

__all__ = ["Calibration", "absmax_scale"]


def absmax_scale(base: torch.Tensor, qtype: qtype = qint8, axis: Optional[int] = None) -> torch.Tensor:
    """max(|base|) / qmax, per tensor (axis=None) or per slice along ``axis`` (calibrate.py:38-64)."""
    mag = torch.abs(base)
    peak = torch.max(mag) if axis is None else torch.amax(mag, dim=axis_to_dim(base, axis), keepdim=True)
    return peak / dtype_info(qtype.dtype).max


def _blend(current, observed, momentum):
    # a scale still at its initial value of one is simply replaced
    return observed if torch.all(current == 1) else momentum * current + observed * (1.0 - momentum)


class Calibration(TorchFunctionMode):
    """Torch-function mode that calibrates the activation scales of quantized modules.

    Args:
        momentum: weight of the running scale when a new batch is observed.
        streamline: drop the output quantization of modules whose outputs only feed functions that dequantize.
        debug: print the resulting configuration per parent module.
    """

    def __init__(self, *args, momentum: float = 0.9, streamline=True, debug=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.momentum, self.streamline, self.debug = momentum, streamline, debug
        if streamline:
            self.modules_qactivations = {}
            self.streamline_hooks = {}

    def __torch_function__(self, func, types, args=(), kwargs=None):
        output = func(*args, **(kwargs or {}))
        if self.streamline and QTensor in types:
            for arg in args:
                producer = getattr(arg, "src_module", None)
                if producer is None:
                    continue
                if isinstance(output, ActivationQBytesTensor):
                    self.modules_qactivations[producer] = True  # this consumer keeps activations quantized
                elif isinstance(output, torch.Tensor):
                    self.modules_qactivations.setdefault(producer, False)
        return output

    def __enter__(self):
        super().__enter__()
        self.pre_handle = register_module_forward_pre_hook(self.calibrate_input)
        self.post_handle = register_module_forward_hook(self.calibrate_output)

    def __exit__(self, exc_type, exc_val, exc_tb):
        super().__exit__(exc_type, exc_val, exc_tb)
        self.pre_handle.remove()
        self.post_handle.remove()
        if self.streamline:
            for handle in self.streamline_hooks.values():
                handle.remove()

    def calibrate_input(self, module: torch.nn.Module, input, momentum: float = 0.9):
        """Global forward-pre hook (runs before the module's own quantize_input hook)."""
        if not (isinstance(module, QModuleMixin) and module.activation_qtype is not None):
            return None
        input = input[0]
        if isinstance(input, ActivationQBytesTensor):
            module.input_scale = torch.max(input._scale)  # already quantized upstream: adopt its scale
        else:
            module.input_scale = _blend(module.input_scale, absmax_scale(input, module.activation_qtype), momentum)
        if self.streamline and module not in self.streamline_hooks:
            # registered after QModuleMixin.quantize_output, hence sees the quantized output
            self.streamline_hooks[module] = module.register_forward_hook(self.tag_outputs)
        return input

    def calibrate_output(self, module: torch.nn.Module, input, output):
        """Global forward hook (runs before the module's own quantize_output hook, on the float output)."""
        if isinstance(module, QModuleMixin) and module.activation_qtype is not None:
            module.output_scale = _blend(module.output_scale, absmax_scale(output, module.activation_qtype, axis=None), self.momentum)
            return output
        if self.streamline:
            for _, child in module.named_children():
                if isinstance(child, QModuleMixin) and child.activation_qtype is not None:
                    if not self.modules_qactivations.get(child, False):
                        child.disable_output_quantization()
        if self.debug:
            for name, child in module.named_children():
                if isinstance(child, QModuleMixin):
                    state = ("not quantized." if child.activation_qtype is None else
                             f"quantized to {child.activation_qtype} with scale {child.output_scale}.")
                    print(f"{name}({child.__class__.__name__}) activations are {state}")
        return None

    def tag_outputs(self, module: torch.nn.Module, input, output):
        """Mark an output with the module that produced it (streamline bookkeeping)."""
        output.src_module = module
