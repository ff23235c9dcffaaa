"""``QuantizedModelForCausalLM``: the driver of BASELINE config 5 (a transformers causal LM with quantized Linears).

Mirrors the reference wrapper (optimum/quanto/models/transformers_models.py:35-183) for local directories: ``quantize``
(= quantize + freeze, lm_head usually excluded), attribute forwarding to the wrapped model (``generate`` ...),
``save_pretrained`` (safetensors of the flattened QTensors + ``quanto_qmap.json``) and ``from_pretrained`` (empty model on
the meta device -> ``requantize``).  Hub download/upload is out of scope (no network on the build or GPU boxes).
"""
import copy
import json
import os
from typing import Any, List, Optional, Union

import torch

from .checkpoint import load_state_dict_to_device, save_sharded_state_dict
from .model_api import freeze, quantization_map, quantize, requantize
from .nn import QLinear, QModuleMixin
from .tensor import Optimizer, qtype

__all__ = ["QuantizedTransformersModel", "QuantizedModelForCausalLM", "fuse_decode_projections"]

_QMAP_NAME = "quanto_qmap.json"


class QuantizedTransformersModel:
    auto_class = None

    def __init__(self, model):
        from transformers import PreTrainedModel

        if not isinstance(model, PreTrainedModel) or len(quantization_map(model)) == 0:
            raise ValueError("The source model must be a quantized transformers model.")
        self._wrapped = model

    def __getattr__(self, name: str) -> Any:
        return getattr(self.__dict__["_wrapped"], name)

    def forward(self, *args, **kwargs):
        return self._wrapped.forward(*args, **kwargs)

    __call__ = forward

    def __repr__(self):
        return repr(self._wrapped)

    @classmethod
    def quantize(cls, model, weights: Optional[Union[str, qtype]] = None, activations: Optional[Union[str, qtype]] = None,
                 optimizer: Optional[Optimizer] = None, include: Optional[Union[str, List[str]]] = None,
                 exclude: Optional[Union[str, List[str]]] = None):
        """Quantize in place and freeze; returns the wrapper (the result is inference-only, as in the reference)."""
        from transformers import PreTrainedModel

        if not isinstance(model, PreTrainedModel):
            raise ValueError("The source model must be a transformers model.")
        quantize(model, weights=weights, activations=activations, optimizer=optimizer, include=include, exclude=exclude)
        freeze(model)
        return cls(model)

    def save_pretrained(self, save_directory: Union[str, os.PathLike], max_shard_size: Optional[int] = None) -> None:

        model = self._wrapped
        os.makedirs(save_directory, exist_ok=True)
        if getattr(model.config, "tie_word_embeddings", True):
            if isinstance(model.get_input_embeddings(), QModuleMixin) or isinstance(model.get_output_embeddings(), QModuleMixin):
                model.config.tie_word_embeddings = False  # a quantized embedding / head is no longer tied
        # record the dtype the model computes in (transformers' own save_pretrained does; config.save_pretrained alone does not): from_pretrained
        # builds its skeleton in it, so the quantized scales come back as saved
        first = next((p for p in model.parameters() if type(p.data) is torch.Tensor and p.is_floating_point()), None)
        # (the stamp goes on a COPY: the live config keeps its attributes - a str where transformers expects a torch.dtype breaks model.to(config.dtype))
        cfg = copy.deepcopy(model.config)
        stamped = None
        if first is not None and getattr(cfg, "dtype", None) is None and getattr(cfg, "torch_dtype", None) is None:
            # transformers >= 4.56 serialises config.dtype; older ones only know torch_dtype (and would fail to JSON-dump a raw torch.dtype
            # stored under an attribute they do not know): set the one this version has
            stamped = "dtype" if "dtype" in getattr(type(cfg), "__dict__", {}) or hasattr(cfg, "dtype") else "torch_dtype"
            try:
                setattr(cfg, stamped, first.dtype)
            except Exception:  # a config class without either attribute: the loader falls back to from_config's default
                stamped = None
        try:
            cfg.save_pretrained(save_directory)
        except TypeError:  # the attribute is not one this transformers version serialises: store the name instead ("bfloat16")
            if stamped is None:
                raise
            setattr(cfg, stamped, str(first.dtype).split(".")[-1])
            cfg.save_pretrained(save_directory)
        state = {k: v.contiguous().cpu() for k, v in model.state_dict().items()}
        if getattr(model.config, "tie_word_embeddings", False) and model.get_output_embeddings() is not None:
            state.pop("lm_head.weight", None)  # shared storage: safetensors refuses aliases
        save_sharded_state_dict(state, str(save_directory), max_shard_size)
        with open(os.path.join(save_directory, _QMAP_NAME), "w", encoding="utf8") as f:
            json.dump(quantization_map(model), f, indent=4)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: Union[str, os.PathLike], device: Optional[torch.device] = None):
        if cls.auto_class is None:
            raise ValueError("use a specialized class such as QuantizedModelForCausalLM to reload a quantized model")
        from accelerate import init_empty_weights
        from transformers import AutoConfig

        path = str(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            raise ValueError(f"{path} is not a local directory (hub downloads are not supported by this backend)")
        qmap_path = os.path.join(path, _QMAP_NAME)
        if not os.path.exists(qmap_path):
            raise ValueError(f"No quantization map found in {path}: is this a quantized model ?")
        with open(qmap_path, "r", encoding="utf-8") as f:
            qmap = json.load(f)
        config = AutoConfig.from_pretrained(path)
        # the skeleton in the dtype the checkpoint was saved in (config.dtype): quantized scales are then taken as they are (requantize
        # only re-rounds them when the skeleton's dtype differs) and the reloaded model computes in the dtype it was quantized in
        dtype = getattr(config, "dtype", None) or getattr(config, "torch_dtype", None)
        with init_empty_weights():
            try:
                model = cls.auto_class().from_config(config, dtype=dtype) if isinstance(dtype, torch.dtype) else cls.auto_class().from_config(config)
            except TypeError:  # older transformers: torch_dtype=
                model = cls.auto_class().from_config(config, torch_dtype=dtype)
        # shards are read straight onto `device` (checkpoint.py): no CPU staging copy of the quantized weights
        requantize(model, state_dict=load_state_dict_to_device(path, device), quantization_map=qmap, device=device)
        if getattr(model.config, "tie_word_embeddings", True):
            model.tie_weights()
        model.eval()
        return cls(model)


class QuantizedModelForCausalLM(QuantizedTransformersModel):
    @staticmethod
    def auto_class():
        from transformers import AutoModelForCausalLM

        return AutoModelForCausalLM


# ------------------------------------------------------------------------------------------------------------------------
# decode-time launch fusion: Linears that read the same activation (q/k/v, gate/up) in ONE quanto::qbits_mm_multi launch
# ------------------------------------------------------------------------------------------------------------------------
class _SiblingGroup:
    """The frozen int4 (or int8 / fp8) QLinears of one parent module that are applied to the same input tensor.

    The model code is left alone (it still calls ``q_proj(h)``, ``k_proj(h)``, ``v_proj(h)`` one after the other): the first
    sibling called with a decode-shaped input runs ``quanto::qbits_mm_multi`` / ``qbytes_mm_multi`` for all of them and parks the other outputs,
    which the following calls pick up when they arrive with the very same tensor - same object, same storage address and same
    version counter (an in-place update of the activation between two siblings makes the parked outputs stale: they are dropped and
    the sibling recomputes).  Anything else - another input, a prefill-sized input, an unfrozen weight, gradients wanted for the
    input or for a bias - takes the module's normal forward."""

    def __init__(self, modules):
        self.modules = modules
        self._clear()

    def _clear(self):
        self.input = None      # strong reference: the storage cannot be recycled while outputs are parked
        self.stamp = None      # (data_ptr, _version) of the input when the outputs were computed
        self.outputs = {}

    def __getstate__(self):    # copy.deepcopy / pickle: the links travel (to the copied modules), parked tensors do not
        return {"modules": self.modules}

    def __setstate__(self, state):
        self.modules = state["modules"]
        self._clear()

    def wants_grad(self, x) -> bool:
        """The multi ops have no autograd formula: a call that wants a gradient (for the input or for a member's bias) must go
        through the members' own forward, or the gradient would be dropped without an error."""
        return torch.is_grad_enabled() and (x.requires_grad or any(m.bias is not None and m.bias.requires_grad for m in self.modules))

    def kind(self, x):
        """"qbits" (int4, group size 128), "qbytes" (int8 / fp8, per-channel scale) or None when the members cannot share a launch."""
        rows = x.numel() // x.shape[-1] if type(x) is torch.Tensor and x.dim() > 0 else 0
        if type(x) is not torch.Tensor or not x.is_cuda or not 1 <= rows <= 64:
            return None
        if self.wants_grad(x):
            return None
        from .tensor import WeightQBitsTensor, WeightQBytesTensor

        w0 = self.modules[0].weight
        if any(m.activation_qtype is not None or m.weight.dtype != x.dtype or m.weight.shape[1] != w0.shape[1] for m in self.modules):
            return None
        if all(isinstance(m.weight, WeightQBitsTensor) for m in self.modules):
            ok = all(w.qtype.bits == 4 and w._group_size == 128 and w.axis == 0 and w._shift.dtype == w0._shift.dtype
                     for w in (m.weight for m in self.modules))
            return "qbits" if ok else None
        if all(type(m.weight) is WeightQBytesTensor for m in self.modules):
            ok = all(w.axis == 0 and w._data.dtype == w0._data.dtype and w._scale.numel() == w.shape[0] and w._data.dim() == 2
                     for w in (m.weight for m in self.modules))
            return "qbytes" if ok else None
        return None

    @staticmethod
    def _stamp(x):
        """(storage address, version counter).  Inference tensors (created under ``torch.inference_mode()``) track no version
        counter - reading ``_version`` raises - and cannot be updated in place outside inference mode: the address alone identifies
        their contents for the few calls a parked output lives."""
        return (x.data_ptr(), None if x.is_inference() else x._version)

    def forward(self, index: int, x):
        if self.input is x and index in self.outputs and self.stamp == self._stamp(x):
            y = self.outputs.pop(index)
            if not self.outputs:
                self._clear()
            return y
        self._clear()
        kind = self.kind(x)
        if kind is None:
            return None
        ws = [m.weight for m in self.modules]
        biases = [m.bias for m in self.modules]
        if kind == "qbits":
            ys = torch.ops.quanto.qbits_mm_multi(x, [w._data._data for w in ws], [w._scale for w in ws], [w._shift for w in ws], biases,
                                                 4, 128, [w.shape[0] for w in ws], ws[0].shape[1])
        else:
            ys = torch.ops.quanto.qbytes_mm_multi(x, [w._data for w in ws], [w._scale for w in ws], biases)
        self.input, self.stamp = x, self._stamp(x)
        self.outputs = {i: y for i, y in enumerate(ys) if i != index}
        return ys[index]


class FusedDecodeQLinear(QLinear):
    """A ``QLinear`` linked to its siblings by ``fuse_decode_projections``.  The link is the module's class plus two plain attributes
    (``_sibling_group``, ``_sibling_index``), not a closure stored in ``module.forward``: ``copy.deepcopy`` / pickle produce modules
    that are linked to their OWN copied siblings, and wrappers that replace ``module.forward`` on the instance (accelerate hooks)
    compose with it."""

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        group = self.__dict__.get("_sibling_group")
        if group is not None:
            y = group.forward(self._sibling_index, input)
            if y is not None:
                return y
        return super().forward(input)


def fuse_decode_projections(model, groups=(("q_proj", "k_proj", "v_proj"), ("gate_proj", "up_proj"))):
    """Opt-in: make sibling QLinears (same parent, same input) share one kernel launch at decode time (up to 64 rows).

    Weights, state dict and module tree are untouched; the listed children become ``FusedDecodeQLinear`` (same parameters, same
    state-dict keys).  Returns the number of groups that were linked."""
    linked = 0
    for parent in model.modules():
        for names in groups:
            mods = [getattr(parent, n, None) for n in names]
            if not all(type(m) in (QLinear, FusedDecodeQLinear) for m in mods) or any(m.__dict__.get("_sibling_group") is not None for m in mods):
                continue
            group = _SiblingGroup(mods)
            for i, m in enumerate(mods):
                m.__class__ = FusedDecodeQLinear
                m._sibling_group, m._sibling_index = group, i
            linked += 1
    return linked
