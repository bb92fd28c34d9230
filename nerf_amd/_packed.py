"""Packed-weight cache shared by the two network modules."""
from typing import List

import torch

from . import ops


class PackedWeightsMixin:
    """Keeps one fragment-ordered weight blob per precision and re-packs it (a GPU kernel, a few
    microseconds) whenever a parameter was modified in place (optimizer step, load_state_dict)."""

    _net_id: int = -1

    def _linear_layers(self) -> List[torch.nn.Linear]:
        raise NotImplementedError

    def packed(self, precision: int) -> torch.Tensor:
        layers = self._linear_layers()
        key = tuple((l.weight.data_ptr(), l.weight._version, l.bias.data_ptr(), l.bias._version) for l in layers)
        cache = self.__dict__.setdefault("_packed_cache", {})
        hit = cache.get(precision)
        if hit is None or hit[0] != key:
            blob = ops.pack_weights(self._net_id, precision, [l.weight for l in layers], [l.bias for l in layers])
            cache[precision] = (key, blob)
            return blob
        return hit[1]


def require_no_grad(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(
            "nerf_amd: the HIP path is forward-only in this round (backward = SURVEY.md section 8f-1); "
            "wrap the call in torch.no_grad()")
