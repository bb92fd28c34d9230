"""Packed-weight cache shared by the two network modules."""
from typing import List

import torch

from . import ops


class PackedWeightsMixin:
    """Keeps one fragment-ordered weight blob per precision and re-packs it (a GPU kernel, a few microseconds) when the parameters
    may have changed.

    * eval mode: the blob is cached under (data_ptr, tensor._version) of every parameter, i.e. re-packed after an optimizer step or
      load_state_dict done by the Python process;
    * train mode: ALWAYS re-packed and never cached -- parameters also change without `_version` moving (a training step replayed
      from a hipGraph updates them on the device only), so a version key cannot be trusted while training;
    * every train()/eval() switch and `invalidate_packed()` drop the cache, so the first eval-mode render after (graph-replayed)
      training packs the current weights.  Call `invalidate_packed()` yourself when parameters are modified behind torch's back while
      the module stays in eval mode.
    """

    _net_id: int = -1

    def _linear_layers(self) -> List[torch.nn.Linear]:
        raise NotImplementedError

    def _pack_now(self, precision: int) -> torch.Tensor:
        layers = self._linear_layers()
        return ops.pack_weights(self._net_id, precision, [l.weight for l in layers], [l.bias for l in layers])

    def packed_backward(self, precision: int) -> torch.Tensor:
        """The transposed weights for the dgrad chain (nerf_amd_pack_weights_backward); packed when asked for -- the backward runs once
        per training step, after which the weights change anyway."""
        return ops.pack_weights_backward(self._net_id, precision, [l.weight for l in self._linear_layers()])

    def _packed_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def invalidate_packed(self) -> None:
        self.__dict__.setdefault("_packed_cache", {}).clear()

    def train(self, mode: bool = True):
        self.invalidate_packed()
        return super().train(mode)

    def packed(self, precision: int) -> torch.Tensor:
        cache = self.__dict__.setdefault("_packed_cache", {})
        if self.training:
            cache.clear()
            return self._pack_now(precision)
        key = self._packed_key()
        hit = cache.get(precision)
        if hit is None or hit[0] != key:
            blob = self._pack_now(precision)
            cache[precision] = (key, blob)
            return blob
        return hit[1]


def require_no_grad(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(
            "nerf_amd: the HIP path is forward-only in this round (backward = SURVEY.md section 8f-1); "
            "wrap the call in torch.no_grad()")
