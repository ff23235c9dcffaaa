"""Quantized module mixin (API of optimum/quanto/nn/qmodule.py:38-308) for the QLinear path.

``QModuleMixin`` keeps the float weight until ``freeze()`` replaces it by a ``WeightQBytesTensor`` /
``WeightQBitsTensor`` parameter; ``qweight`` quantizes dynamically while unfrozen so gradients reach the float
weight.  Group size selection (128, stepping down by 32 until it divides in_features) and the default scale
optimizers are the reference's (qmodule.py:121-137) because they decide the layout the kernels read.
"""
from abc import ABC
from typing import Optional, Union

import torch

from ..tensor import (AbsmaxOptimizer, ActivationQBytesTensor, MaxOptimizer, Optimizer, QTensor, SymmetricOptimizer,
                      WeightQBitsTensor, WeightQBytesTensor, qint2, qint4, qtype, qtypes, quantize_activation, quantize_weight)

__all__ = ["QModuleMixin", "register_qmodule", "quantize_module"]

_QMODULE_TABLE = {}


def register_qmodule(module_cls):
    """Class decorator: declare the quantized counterpart of ``module_cls`` (qmodule.py:44-78)."""

    def wrapper(cls):
        _QMODULE_TABLE[module_cls] = cls
        return cls

    return wrapper


def quantize_module(module, weights=None, activations=None, optimizer: Optional[Optimizer] = None):
    """Return the quantized version of ``module`` or None when its class has no registered counterpart."""
    for cls, qcls in _QMODULE_TABLE.items():
        if isinstance(module, cls):
            return qcls.from_module(module, weights=weights, activations=activations, optimizer=optimizer)
    return None


def _as_qtype(q: Optional[Union[qtype, str]]):
    return q if (q is None or isinstance(q, qtype)) else qtypes[q]


def select_group_size(in_features: int) -> Optional[int]:
    """128 if possible, else the largest of 96/64/32 dividing ``in_features``; None -> per-channel (qmodule.py:121-129)."""
    group_size = 128
    if in_features <= group_size:
        return None
    while in_features % group_size != 0 and group_size > 32:
        group_size -= 32
    return group_size if in_features % group_size == 0 else None


class QModuleMixin(ABC):
    def __init__(self, *args, weights=None, activations=None, optimizer: Optional[Optimizer] = None,
                 quantize_input: Optional[bool] = False, device: Optional[torch.device] = None, **kwargs):
        mro = self.__class__.__mro__
        if torch.nn.Module not in mro:
            raise TypeError("Quantized modules must inherit from a torch.nn.Module class")
        if mro.index(__class__) > mro.index(torch.nn.Module):
            raise TypeError("QModuleMixin must be placed before any torch.nn.Module class in quantized module inheritance.")
        super().__init__(*args, device=device, **kwargs)
        weights, activations = _as_qtype(weights), _as_qtype(activations)
        self.weight_qtype = weights
        self.weight_group_size = None
        if weights in (qint2, qint4):
            out_features = self.weight.shape[0]
            self.weight_group_size = select_group_size(self.weight.numel() // out_features)
        self.activation_qtype = activations
        self._quantize_hooks = {}
        if activations is not None:
            # inputs are quantized with `input_scale` before forward, outputs with `output_scale` after it
            # (qmodule.py:131-134); both scales come from a Calibration pass
            if quantize_input:
                self._quantize_hooks["input"] = self.register_forward_pre_hook(self.quantize_input)
            self._quantize_hooks["output"] = self.register_forward_hook(self.quantize_output)
        if optimizer is None and weights is not None:
            optimizer = AbsmaxOptimizer() if weights.bits == 8 else MaxOptimizer()
        self.optimizer = optimizer
        scale_dtype = torch.float32 if self.weight is None else self.weight.dtype
        self.register_buffer("input_scale", torch.ones((), dtype=scale_dtype, device=device))
        self.register_buffer("output_scale", torch.ones((), dtype=scale_dtype, device=device))

    def disable_output_quantization(self):
        hook = self._quantize_hooks.pop("output", None)
        if hook is not None:
            hook.remove()

    # -- state dict: frozen weights are stored flattened (weight._data, weight._scale, ...) ------------
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self.weight_qtype is None or not self.frozen:
            destination[prefix + "weight"] = self.weight if (self.weight is None or keep_vars) else self.weight.detach()
        else:
            self.weight.save_to_state_dict(destination, prefix + "weight.", keep_vars)
        if self.bias is not None:
            destination[prefix + "bias"] = self.bias if keep_vars else self.bias.detach()
        for name in ("input_scale", "output_scale"):
            buf = getattr(self, name)
            destination[prefix + name] = buf if keep_vars else buf.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        weight_name = prefix + "weight"
        if self.weight_qtype is not None and weight_name not in state_dict:
            # a frozen checkpoint: rebuild the QTensor from its flattened inner tensors (qmodule.py:161-207)
            if self.weight_qtype.bits == 8:
                qw = WeightQBytesTensor.load_from_state_dict(
                    state_dict, weight_name + ".", qtype=self.weight_qtype, axis=0, size=self.weight.size(),
                    stride=self.weight.stride(), activation_qtype=self.activation_qtype, missing_keys=missing_keys)
            else:
                qw = WeightQBitsTensor.load_from_state_dict(
                    state_dict, weight_name + ".", qtype=self.weight_qtype, axis=0, group_size=self.weight_group_size,
                    size=self.weight.size(), stride=self.weight.stride(), missing_keys=missing_keys)
            if qw is not None:
                qw = qw.optimize()
                if local_metadata.get("assign_to_params_buffers", False):
                    self.weight = torch.nn.Parameter(qw)
                else:
                    self.weight = torch.nn.Parameter(qw.to(self.weight.device))
        super()._load_from_state_dict(state_dict, prefix, local_metadata, False, missing_keys, unexpected_keys, error_msgs)

    # -- construction ---------------------------------------------------------------------------------
    @classmethod
    def from_module(cls, module: torch.nn.Module, weights=None, activations=None, optimizer: Optional[Optimizer] = None):
        """Build on the meta device, then alias the float parameters of ``module`` (no copy) - qmodule.py:209-232."""
        qmodule = cls.qcreate(module, weights, activations, optimizer, device="meta")
        if qmodule is None:
            return None
        device = torch.device("cpu") if module.weight is None else module.weight.device
        qmodule = qmodule.to_empty(device=device)
        qmodule.input_scale = torch.ones_like(qmodule.input_scale)
        qmodule.output_scale = torch.ones_like(qmodule.output_scale)
        with torch.no_grad():
            qmodule.weight = module.weight
            if module.bias is not None:
                qmodule.bias = module.bias
        return qmodule.to(device)

    @classmethod
    def qcreate(cls, module, weights, activations=None, optimizer=None, device=None):
        raise NotImplementedError

    # -- quantized weight -----------------------------------------------------------------------------
    @property
    def qweight(self):
        """Frozen: the stored QTensor.  Unfrozen: quantize the float weight on the fly (qmodule.py:245-279)."""
        if self.weight_qtype is None:
            return None
        if isinstance(self.weight, QTensor):
            return self.weight
        if isinstance(self.optimizer, SymmetricOptimizer):
            scale, shift = self.optimizer(self.weight, qtype=self.weight_qtype, axis=0), None
        else:
            scale, shift = self.optimizer(self.weight, qtype=self.weight_qtype, axis=0, group_size=self.weight_group_size)
        return quantize_weight(self.weight, qtype=self.weight_qtype, axis=0, scale=scale, shift=shift,
                               group_size=self.weight_group_size, activation_qtype=self.activation_qtype)

    def qforward(self, input: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    # -- quantized activations (qmodule.py:281-299) -----------------------------------------------------
    def quantize_input(self, module: torch.nn.Module, input: torch.Tensor) -> torch.Tensor:
        input = input[0]
        if isinstance(input, ActivationQBytesTensor):
            if input.qtype != self.activation_qtype:
                raise ValueError("Models with heterogeneous quantized activations are not supported:"
                                 f" expected {self.activation_qtype.name} input but got {input.qtype.name} instead.")
            return input
        return quantize_activation(input, qtype=self.activation_qtype, scale=self.input_scale)

    def quantize_output(self, module: torch.nn.Module, input: torch.Tensor, output: torch.Tensor) -> torch.Tensor:
        return quantize_activation(output, qtype=self.activation_qtype, scale=self.output_scale)

    def freeze(self):
        qweight = self.qweight
        if qweight is not None:
            self.weight = torch.nn.Parameter(qweight)

    @property
    def frozen(self) -> bool:
        return isinstance(self.weight, QTensor)
