"""``QuantizedModelForCausalLM``: the driver of BASELINE config 5 (a transformers causal LM with quantized Linears).

Mirrors the reference wrapper (optimum/quanto/models/transformers_models.py:35-183) for local directories: ``quantize``
(= quantize + freeze, lm_head usually excluded), attribute forwarding to the wrapped model (``generate`` ...),
``save_pretrained`` (safetensors of the flattened QTensors + ``quanto_qmap.json``) and ``from_pretrained`` (empty model on
the meta device -> ``requantize``).  Hub download/upload is out of scope (no network on the build or GPU boxes).
"""
import json
import os
from typing import Any, List, Optional, Union

import torch

from .checkpoint import load_state_dict_to_device, save_sharded_state_dict
from .model_api import freeze, quantization_map, quantize, requantize
from .nn import QModuleMixin
from .tensor import Optimizer, qtype

__all__ = ["QuantizedTransformersModel", "QuantizedModelForCausalLM"]

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
        model.config.save_pretrained(save_directory)
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
        with init_empty_weights():
            model = cls.auto_class().from_config(config)
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
