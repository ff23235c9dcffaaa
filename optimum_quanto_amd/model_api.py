"""Model-level API: ``quantize``, ``freeze``, ``requantize``, ``quantization_map`` (optimum/quanto/quantize.py:24-149)."""
from fnmatch import fnmatch
from typing import Any, Dict, List, Optional, Union

import torch

from .nn import QModuleMixin, quantize_module
from .tensor import Optimizer, QTensor, qtype

__all__ = ["quantize", "freeze", "requantize", "quantization_map"]


def _set_module_by_name(parent: torch.nn.Module, name: str, child: torch.nn.Module) -> None:
    *path, leaf = name.split(".")
    for part in path:
        parent = getattr(parent, part)
    setattr(parent, leaf, child)


def _quantize_submodule(model, name, module, weights=None, activations=None, optimizer=None):
    qmodule = quantize_module(module, weights=weights, activations=activations, optimizer=optimizer)
    if qmodule is None:
        return False
    _set_module_by_name(model, name, qmodule)
    qmodule.name = name
    for pname, param in module.named_parameters():
        # the quantized module aliases the parameters: release the originals
        setattr(module, pname, None)
        del param
    return True


def _as_patterns(p: Optional[Union[str, List[str]]]):
    return [p] if isinstance(p, str) else p


def quantize(model: torch.nn.Module, weights: Optional[Union[str, qtype]] = None,
             activations: Optional[Union[str, qtype]] = None, optimizer: Optional[Optimizer] = None,
             include: Optional[Union[str, List[str]]] = None, exclude: Optional[Union[str, List[str]]] = None):
    """Replace every eligible submodule in place by its quantized counterpart.

    ``include`` / ``exclude`` are Unix shell-style patterns on module names (quantize.py:55-98).  Weights stay
    float (dynamically quantized in forward) until ``freeze``.
    """
    include, exclude = _as_patterns(include), _as_patterns(exclude)
    for name, module in list(model.named_modules()):
        if include is not None and not any(fnmatch(name, pattern) for pattern in include):
            continue
        if exclude is not None and any(fnmatch(name, pattern) for pattern in exclude):
            continue
        _quantize_submodule(model, name, module, weights=weights, activations=activations, optimizer=optimizer)


def requantize(model: torch.nn.Module, state_dict: Dict[str, Any], quantization_map: Dict[str, Dict[str, str]],
               device: torch.device = None):
    """Rebuild a frozen model from a flattened state dict + ``quantization_map`` (quantize.py:101-140).

    One deliberate divergence from the reference: when the checkpoint's float dtype differs from the dtype ``model`` was built in
    (an fp32 checkpoint opened as a bf16 skeleton), the reference keeps the deserialized scale / shift as they are
    (nn/qmodule.py:161-207) and the module then mixes dtypes (its ``forward`` raises on fp32 scale x bf16 bias).  Here the rebuilt
    weight's scale and float shift are cast to the model's dtype - integers untouched - and a ``UserWarning`` names the modules:
    the dequantized weights are then the checkpoint's values re-rounded to the model dtype, not bit-identical to what was saved.
    Build the skeleton in the checkpoint's dtype to get the reference's bits."""
    if device is None:
        device = next(model.parameters()).device
        if device.type == "meta":
            device = torch.device("cpu")
    not_rebuilt = []
    for name, module in list(model.named_modules()):
        qconfig = quantization_map.get(name)
        if qconfig is None:
            continue
        weights = None if qconfig["weights"] == "none" else qconfig["weights"]
        activations = None if qconfig["activations"] == "none" else qconfig["activations"]
        if not _quantize_submodule(model, name, module, weights=weights, activations=activations):
            not_rebuilt.append(f"{name} ({type(module).__name__})")
    missing = sorted(set(quantization_map) - {n for n, _ in model.named_modules()})
    # Materialise what is still on the meta device, then load.  The float ``weight`` of a module whose quantized weight is in the
    # state dict is never materialised (the reference stages everything on the CPU, quantize.py:123-137; on the device that would
    # cost the full float model next to the quantized one): it stays on meta until ``load_state_dict(assign=True)`` replaces it.
    def serialized(prefix: str) -> bool:
        return any(k.startswith(prefix + "weight._") for k in state_dict)

    for name, m in model.named_modules():
        def move(t):
            if t.device.type == "meta":
                return torch.empty_like(t, device=device)
            return t.to(device)

        skip_weight = isinstance(m, QModuleMixin) and serialized(name + "." if name else "")
        for pname, p in list(m.named_parameters(recurse=False)):
            if skip_weight and pname == "weight" and p.device.type == "meta":
                continue
            setattr(m, pname, torch.nn.Parameter(move(p), requires_grad=p.requires_grad))
        for bname, b in list(m.named_buffers(recurse=False)):
            setattr(m, bname, move(b))
    # dtype of the model's own (non-quantized) parameters wins over the checkpoint's, as with a copying load_state_dict
    want = {k: v.dtype for k, v in list(model.named_parameters()) + list(model.named_buffers()) if v.device.type != "meta"}
    # ... including a quantized module's weight: its scale (and float shift) follow the dtype the module was built in, so that a
    # checkpoint saved in another float dtype does not leave bias / activations in one dtype and the weight's scale in another
    qdtype = {name: m.weight.dtype for name, m in model.named_modules() if isinstance(m, QModuleMixin) and m.weight is not None}
    loaded = model.load_state_dict(state_dict, strict=False, assign=True)
    # A checkpoint written by the reference may quantize module types this package has no counterpart for (the reference's QLayerNorm,
    # nn/qlayernorm.py): loading it silently would compute something else than what was saved - say so, loudly.
    if not_rebuilt or missing or loaded.unexpected_keys:
        import warnings

        parts = []
        if not_rebuilt:
            parts.append("no quantized counterpart for " + ", ".join(not_rebuilt[:4]) + (f" and {len(not_rebuilt) - 4} more" if len(not_rebuilt) > 4 else "")
                         + " (kept as float modules: their input / output scales are dropped)")
        if missing:
            parts.append("quantization_map names modules the model does not have: " + ", ".join(missing[:4]))
        if loaded.unexpected_keys:
            parts.append(f"{len(loaded.unexpected_keys)} state-dict entries were not used, e.g. " + ", ".join(list(loaded.unexpected_keys)[:4]))
        warnings.warn("requantize: the rebuilt model differs from the checkpoint - " + "; ".join(parts), UserWarning)
    for k, v in list(model.named_parameters()) + list(model.named_buffers()):
        if k in want and type(v.data) is torch.Tensor and v.is_floating_point() and v.dtype != want[k]:
            v.data = v.data.to(want[k])
    recast = []
    for name, m in model.named_modules():
        if name in qdtype and isinstance(m.weight, QTensor) and m.weight._scale.dtype != qdtype[name]:
            recast.append(f"{name} ({m.weight._scale.dtype} -> {qdtype[name]})")
            m.weight = torch.nn.Parameter(_cast_qweight(m.weight, qdtype[name]), requires_grad=False)
    if recast:
        import warnings

        warnings.warn("requantize: the checkpoint's scale dtype differs from the model's; scale / shift of " + ", ".join(recast[:4]) +
                      (f" and {len(recast) - 4} more" if len(recast) > 4 else "") + " were cast to the model dtype (see the docstring)", UserWarning)
    model.to(device)


def _cast_qweight(qw, dtype):
    """The same quantized weight with its float metadata (scale, float shift) in ``dtype``; integers untouched."""
    names, meta = qw.__tensor_flatten__()
    inner = {}
    for n in names:
        t = getattr(qw, n)
        inner[n] = t.to(dtype) if type(t) is torch.Tensor and t.is_floating_point() and n in ("_scale", "_shift") else t
    return type(qw).__tensor_unflatten__(inner, meta, None, None)


def freeze(model: torch.nn.Module):
    for m in model.modules():
        if isinstance(m, QModuleMixin):
            m.freeze()


def quantization_map(model: torch.nn.Module) -> Dict[str, Dict[str, str]]:
    """``{module name: {"weights": qtype name | "none", "activations": ...}}`` for every quantized module."""
    config = {}
    for name, m in model.named_modules():
        if isinstance(m, QModuleMixin):
            config[name] = {
                "weights": "none" if m.weight_qtype is None else m.weight_qtype.name,
                "activations": "none" if m.activation_qtype is None else m.activation_qtype.name,
            }
    return config
