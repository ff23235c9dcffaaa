"""Plug-in mode: install the MI355X kernels into an already imported, unmodified ``optimum.quanto``.

``import optimum.quanto`` first, then ``import optimum_quanto_amd``: ``library/ops.py`` finds the ``quanto::`` operators
already defined by the reference, overrides their ``CUDA`` (= ROCm) implementations and adds the fused ops the reference
lacks.  ``install()`` below does the two things an op override cannot:

* registers ``quanto_hip`` in the REFERENCE's extension registry (``library/extensions/extension.py:58-86``), replacing the
  JIT-built one-kernel extension of ``library/extensions/hip/__init__.py:18-28`` when a ROCm device made the reference create
  it - ``get_extension("quanto_hip").lib`` is then the ctypes binding of ``libquanto_hip.so``
  (the reference's own ``tests/library/test_extensions.py:23-24`` keeps passing);
* routes ``F.linear`` on the reference's generic ``WeightQBitsTensor`` to ``quanto::qbits_mm`` for axis-0 2-D packed weights
  on a ROCm device - the branch ``WeightQBitsTensor.create()`` has for CUDA AWQ / TinyGemm layouts
  (``tensor/weights/qbits.py:97-138``, ``tensor/weights/awq/qbits.py:53-74``), except that the generic, serialisable layout
  is consumed as is, so no new subclass and no repacking.  Everything else (CPU tensors, per-axis -1 weights, activations
  that need gradients through the weight) keeps the reference path.

The reference's Python host code is used unchanged: QModuleMixin, quantize / freeze, QuantizedModelForCausalLM.
"""
import sys

import torch

__all__ = ["install", "installed", "fused_qbits_linear"]

_state = {"installed": False}


def installed() -> bool:
    return _state["installed"]


def _reference():
    return sys.modules.get("optimum.quanto")


def fused_qbits_linear(input, other, bias=None):
    """``F.linear(input, other, bias)`` for a reference ``WeightQBitsTensor`` through ``quanto::qbits_mm``.

    Same shape as ``AWQWeightQBitsLinearFunction.forward`` (tensor/weights/awq/qbits.py:53-74): dequantize a quantized
    activation, call the fused op on the inner tensors, inherit the straight-through backward of
    ``QuantizedLinearFunction`` (tensor/function.py:49-63)."""
    from optimum.quanto.tensor.function import QuantizedLinearFunction

    fn = _state.get("linear_function")
    if fn is None:
        class HipQBitsLinearFunction(QuantizedLinearFunction):
            @staticmethod
            def forward(ctx, input, other, bias=None):
                ctx.save_for_backward(input, other)
                if type(input) is not torch.Tensor:
                    input = input.dequantize()
                out_features, in_features = other.shape
                return torch.ops.quanto.qbits_mm(input, other._data._data, other._scale, other._shift, bias, other._data.bits,
                                                 other._group_size, out_features, in_features)

        fn = _state["linear_function"] = HipQBitsLinearFunction
    return fn.apply(input, other, bias)


def _routable(other) -> bool:
    """Generic class only (optimized subclasses keep their own kernels), axis 0, 2-D, packed data, ROCm device."""
    ref = _reference()
    WeightQBitsTensor = ref.tensor.weights.qbits.WeightQBitsTensor
    PackedTensor = ref.tensor.packed.PackedTensor
    return (type(other) is WeightQBitsTensor and other.axis == 0 and other.ndim == 2 and isinstance(other._data, PackedTensor)
            and other.device.type == "cuda" and torch.version.hip is not None)


def install() -> bool:
    """Idempotent.  Returns True when the reference is imported and the backend was installed into it.  An ``optimum.quanto`` that
    is only partially imported, or laid out differently from the version this was written against, is left untouched (a warning,
    not an ImportError of this package)."""
    if _reference() is None:
        return False
    if _state["installed"]:
        return True
    try:
        return _install()
    except (KeyError, AttributeError, ImportError) as e:
        import warnings

        warnings.warn(f"optimum_quanto_amd: optimum.quanto is imported but plug-in mode could not be installed into it ({type(e).__name__}: {e}); "
                      "the reference is left unmodified", RuntimeWarning)
        return False


def _install() -> bool:
    ref = _reference()
    from .hip import quanto_hip

    # look everything up first: nothing of the reference is modified unless all of it is where this code expects it
    registry = sys.modules["optimum.quanto.library.extensions.extension"]._extensions
    cls = ref.tensor.weights.qbits.WeightQBitsTensor
    original = cls.__dict__["__torch_function__"].__func__
    ref.tensor.packed.PackedTensor, ref.tensor.function.QuantizedLinearFunction  # noqa: B018  (used by the routed function)

    # 1. the reference's extension registry
    if torch.version.hip is not None:
        registry["quanto_hip"] = quanto_hip

    # 2. F.linear on the generic WeightQBitsTensor

    def __torch_function__(klass, func, types, args=(), kwargs=None):
        if func is torch.nn.functional.linear:
            kw = kwargs or {}
            other = args[1] if len(args) > 1 else kw.get("weight")
            if other is not None and _routable(other):
                input = args[0] if args else kw.get("input")
                bias = args[2] if len(args) > 2 else kw.get("bias")
                return fused_qbits_linear(input, other, bias)
        return original(klass, func, types, args, kwargs)

    cls.__torch_function__ = classmethod(__torch_function__)
    _state["original_torch_function"] = original
    _state["installed"] = True
    return True
