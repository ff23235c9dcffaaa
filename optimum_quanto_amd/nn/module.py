"""The quantized-module protocol of the QLinear path (public surface of optimum/quanto/nn/qmodule.py:38-308).

A quantized module keeps its float ``weight`` until ``freeze()`` swaps in a ``WeightQBytesTensor`` / ``WeightQBitsTensor``
parameter; before that ``qweight`` quantizes on the fly, so gradients still reach the float weight.  Two policies are taken
over from the reference unchanged because they decide the byte layout the kernels read: the group size of sub-byte weights
(qmodule.py:121-129) and the default scale search per bit width (qmodule.py:135-137).
"""
from abc import ABC
from typing import Dict, Optional, Type, Union

import torch

from ..tensor import (AbsmaxOptimizer, ActivationQBytesTensor, MaxOptimizer, Optimizer, QTensor, SymmetricOptimizer,
                      WeightQBitsTensor, WeightQBytesTensor, qint2, qint4, qtype, qtypes, quantize_activation, quantize_weight)

__all__ = ["QModuleMixin", "register_qmodule", "quantize_module"]

# float module class -> its quantized counterpart
_counterparts: Dict[Type[torch.nn.Module], type] = {}


def register_qmodule(module_cls):
    """``@register_qmodule(torch.nn.Linear)`` on a class declares it the quantized form of ``module_cls``."""

    def remember(qcls):
        _counterparts[module_cls] = qcls
        return qcls

    return remember


def quantize_module(module, weights=None, activations=None, optimizer: Optional[Optimizer] = None):
    """The quantized twin of ``module`` (sharing its parameters), or None for a class nobody registered."""
    qcls = next((q for base, q in _counterparts.items() if isinstance(module, base)), None)
    return None if qcls is None else qcls.from_module(module, weights=weights, activations=activations, optimizer=optimizer)


def _resolve(q: Optional[Union[qtype, str]]):
    return qtypes[q] if isinstance(q, str) else q


def select_group_size(in_features: int) -> Optional[int]:
    """Largest of 128 / 96 / 64 / 32 that divides ``in_features`` (strictly larger rows only); None means per-channel."""
    if in_features <= 128:
        return None
    return next((g for g in (128, 96, 64, 32) if in_features % g == 0), None)


def _flat_weight_prefix(prefix: str) -> str:
    return prefix + "weight."


class QModuleMixin(ABC):
    """Mix in FRONT of the torch.nn.Module class being quantized: ``class QLinear(QModuleMixin, torch.nn.Linear)``."""

    def __init__(self, *args, weights=None, activations=None, optimizer: Optional[Optimizer] = None,
                 quantize_input: Optional[bool] = False, device: Optional[torch.device] = None, **kwargs):
        self._require_module_base()
        super().__init__(*args, device=device, **kwargs)  # the torch.nn.Module constructor
        self.weight_qtype = _resolve(weights)
        self.activation_qtype = _resolve(activations)
        self.weight_group_size = self._pick_group_size()
        self.optimizer = optimizer if optimizer is not None else self._default_optimizer()
        self._quantize_hooks = {}
        if self.activation_qtype is not None:
            self._hook_activations(quantize_input)
        # calibrated by a Calibration pass; scalars in the weight's dtype
        dtype = self.weight.dtype if self.weight is not None else torch.float32
        for name in ("input_scale", "output_scale"):
            self.register_buffer(name, torch.ones((), dtype=dtype, device=device))

    # -- construction helpers ----------------------------------------------------------------------------------------
    def _require_module_base(self):
        order = type(self).__mro__
        if torch.nn.Module not in order:
            raise TypeError("Quantized modules must inherit from a torch.nn.Module class")
        if order.index(QModuleMixin) > order.index(torch.nn.Module):
            raise TypeError("QModuleMixin must be placed before any torch.nn.Module class in quantized module inheritance.")

    def _pick_group_size(self) -> Optional[int]:
        if self.weight_qtype not in (qint2, qint4):
            return None
        rows = self.weight.shape[0]
        return select_group_size(self.weight.numel() // rows)

    def _default_optimizer(self) -> Optional[Optimizer]:
        if self.weight_qtype is None:
            return None
        return AbsmaxOptimizer() if self.weight_qtype.bits == 8 else MaxOptimizer()

    def _hook_activations(self, also_inputs: bool):
        # outputs are always re-quantized with `output_scale`; inputs only on request (the first quantized module of a chain)
        if also_inputs:
            self._quantize_hooks["input"] = self.register_forward_pre_hook(self.quantize_input)
        self._quantize_hooks["output"] = self.register_forward_hook(self.quantize_output)

    def disable_output_quantization(self):
        handle = self._quantize_hooks.pop("output", None)
        if handle is not None:
            handle.remove()

    @classmethod
    def from_module(cls, module: torch.nn.Module, weights=None, activations=None, optimizer: Optional[Optimizer] = None):
        """Quantized twin of ``module``: created without storage, then pointed at ``module``'s own parameters (no copy)."""
        twin = cls.qcreate(module, weights, activations, optimizer, device="meta")
        if twin is None:
            return None
        where = module.weight.device if module.weight is not None else torch.device("cpu")
        twin = twin.to_empty(device=where)
        for name in ("input_scale", "output_scale"):  # to_empty left them uninitialised
            setattr(twin, name, torch.ones_like(getattr(twin, name)))
        with torch.no_grad():
            twin.weight = module.weight
            if module.bias is not None:
                twin.bias = module.bias
        return twin.to(where)

    @classmethod
    def qcreate(cls, module, weights, activations=None, optimizer=None, device=None):
        raise NotImplementedError

    def qforward(self, input: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    # -- weights ------------------------------------------------------------------------------------------------------------
    @property
    def frozen(self) -> bool:
        return isinstance(self.weight, QTensor)

    @property
    def qweight(self):
        """The weight as the kernels see it: the stored QTensor once frozen, a fresh quantization of the float weight before."""
        if self.weight_qtype is None:
            return None
        if self.frozen:
            return self.weight
        search = dict(qtype=self.weight_qtype, axis=0)
        if isinstance(self.optimizer, SymmetricOptimizer):
            scale, shift = self.optimizer(self.weight, **search), None
        else:
            scale, shift = self.optimizer(self.weight, group_size=self.weight_group_size, **search)
        return quantize_weight(self.weight, scale=scale, shift=shift, group_size=self.weight_group_size,
                               activation_qtype=self.activation_qtype, **search)

    def freeze(self):
        quantized = self.qweight
        if quantized is not None:
            self.weight = torch.nn.Parameter(quantized)

    # -- activations ----------------------------------------------------------------------------------------------------------
    def quantize_input(self, module: torch.nn.Module, input: torch.Tensor) -> torch.Tensor:
        (first,) = input[:1]
        if not isinstance(first, ActivationQBytesTensor):
            return quantize_activation(first, qtype=self.activation_qtype, scale=self.input_scale)
        if first.qtype != self.activation_qtype:
            raise ValueError("Models with heterogeneous quantized activations are not supported:"
                             f" expected {self.activation_qtype.name} input but got {first.qtype.name} instead.")
        return first

    def quantize_output(self, module: torch.nn.Module, input: torch.Tensor, output: torch.Tensor) -> torch.Tensor:
        return quantize_activation(output, qtype=self.activation_qtype, scale=self.output_scale)

    # -- serialisation: a frozen weight travels as its inner tensors (weight._data, weight._scale, ...) -----------------------
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        def put(key, tensor):
            destination[prefix + key] = tensor if (tensor is None or keep_vars) else tensor.detach()

        if self.frozen and self.weight_qtype is not None:
            self.weight.save_to_state_dict(destination, _flat_weight_prefix(prefix), keep_vars)
        else:
            put("weight", self.weight)
        if self.bias is not None:
            put("bias", self.bias)
        put("input_scale", self.input_scale)
        put("output_scale", self.output_scale)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        flattened = self.weight_qtype is not None and (prefix + "weight") not in state_dict
        if flattened:
            rebuilt = self._weight_from_flat(state_dict, _flat_weight_prefix(prefix), missing_keys)
            if rebuilt is not None:
                rebuilt = rebuilt.optimize()  # the device-specific subclass, if one applies
                if not local_metadata.get("assign_to_params_buffers", False):
                    rebuilt = rebuilt.to(self.weight.device)
                self.weight = torch.nn.Parameter(rebuilt)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, False, missing_keys, unexpected_keys, error_msgs)

    def _weight_from_flat(self, state_dict, flat_prefix, missing_keys):
        shape = dict(size=self.weight.size(), stride=self.weight.stride())
        if self.weight_qtype.bits == 8:
            return WeightQBytesTensor.load_from_state_dict(state_dict, flat_prefix, qtype=self.weight_qtype, axis=0,
                                                           activation_qtype=self.activation_qtype, missing_keys=missing_keys, **shape)
        return WeightQBitsTensor.load_from_state_dict(state_dict, flat_prefix, qtype=self.weight_qtype, axis=0,
                                                      group_size=self.weight_group_size, missing_keys=missing_keys, **shape)
