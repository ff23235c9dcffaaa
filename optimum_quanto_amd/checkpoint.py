"""Checkpoint fast path: safetensors shards -> device-resident generic layout, no CPU round trip.

The reference reloads a quantized model by moving every parameter to the CPU, materialising the quantized weights there and
only then moving the model to its device (quantize.py:124-140), reading shards through ``ShardedStateDict``
(models/shared_dict.py:22-53).  Here the flattened inner tensors (``weight._data._data``, ``weight._scale``, ``weight._shift``)
are read by ``safetensors.safe_open(..., device=<target>)`` straight into HBM, and ``requantize`` assembles the QTensors around
those buffers (``load_state_dict(assign=True)``): the on-disk format is the reference's, the kernels consume it as stored.
"""
import json
import os
from collections.abc import Mapping
from typing import Any, Dict, Optional

import torch

__all__ = ["ShardedStateDict", "load_state_dict_to_device", "save_sharded_state_dict", "WEIGHTS_NAME", "WEIGHTS_INDEX_NAME"]

WEIGHTS_NAME = "model.safetensors"
WEIGHTS_INDEX_NAME = "model.safetensors.index.json"


def _device_str(device: Optional[torch.device]) -> str:
    if device is None:
        return "cpu"
    device = torch.device(device)
    if device.type == "cuda":
        return f"cuda:{device.index if device.index is not None else torch.cuda.current_device()}"
    return str(device)


class ShardedStateDict(Mapping):
    """A state dict spread over several safetensors files, read lazily and directly onto ``device``."""

    def __init__(self, base_dir: str, tensor_index: Dict[str, str], device: Optional[torch.device] = None):
        self._base_dir = base_dir
        self._index = dict(tensor_index)
        self._device = _device_str(device)
        self._handles = {}

    def _handle(self, filename: str):
        from safetensors import safe_open

        if filename not in self._handles:
            self._handles[filename] = safe_open(os.path.join(self._base_dir, filename), framework="pytorch", device=self._device)
        return self._handles[filename]

    def __getitem__(self, key: Any) -> torch.Tensor:
        return self._handle(self._index[key]).get_tensor(key)

    def __iter__(self):
        yield from self._index

    def __len__(self):
        return len(self._index)

    def __contains__(self, key: object) -> bool:
        return key in self._index

    def keys(self):
        return self._index.keys()


def load_state_dict_to_device(directory: str, device: Optional[torch.device] = None) -> Mapping:
    """The state dict of a (sharded or single-file) safetensors checkpoint with every tensor placed on ``device``."""
    index_path = os.path.join(directory, WEIGHTS_INDEX_NAME)
    if os.path.exists(index_path):
        with open(index_path, "r", encoding="utf-8") as f:
            weight_map = json.load(f)["weight_map"]
        return ShardedStateDict(directory, weight_map, device)
    single = os.path.join(directory, WEIGHTS_NAME)
    if not os.path.exists(single):
        raise ValueError(f"No safetensor weights found in {directory}.")
    from safetensors.torch import load_file

    return load_file(single, device=_device_str(device))


def save_sharded_state_dict(state: Dict[str, torch.Tensor], directory: str, max_shard_bytes: Optional[int] = None) -> None:
    """One ``model.safetensors``, or ``model-0000i-of-0000n.safetensors`` shards + index when ``max_shard_bytes`` is exceeded."""
    from safetensors.torch import save_file

    sizes = {k: v.numel() * v.element_size() for k, v in state.items()}
    if max_shard_bytes is None or sum(sizes.values()) <= max_shard_bytes:
        save_file(state, os.path.join(directory, WEIGHTS_NAME), metadata={"format": "pt"})
        return
    shards, current, used = [], {}, 0
    for k, v in state.items():
        if current and used + sizes[k] > max_shard_bytes:
            shards.append(current)
            current, used = {}, 0
        current[k] = v
        used += sizes[k]
    shards.append(current)
    weight_map = {}
    for i, shard in enumerate(shards):
        name = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(shard, os.path.join(directory, name), metadata={"format": "pt"})
        weight_map.update({k: name for k in shard})
    with open(os.path.join(directory, WEIGHTS_INDEX_NAME), "w", encoding="utf-8") as f:
        json.dump({"metadata": {"total_size": sum(sizes.values())}, "weight_map": weight_map}, f, indent=2)
