"""Packed-weight cache shared by the two network modules."""
from typing import List

import torch

from . import ops


class PackedWeightsMixin:
    """Keeps one fragment-ordered weight blob per precision and re-packs it (a GPU kernel, a few microseconds) when the parameters
    may have changed.

    * eval mode: the blob is cached under (data_ptr, tensor._version) of every parameter + `ops.PARAM_GENERATION` (bumped by every
      nerf_amd_adam_step launch and every replayed TrainStep graph, which write parameters through raw pointers), i.e. re-packed after
      an optimizer step -- torch's or this package's -- or load_state_dict;
    * train mode: ALWAYS re-packed and never cached -- parameters also change without `_version` moving (a training step replayed
      from a hipGraph updates them on the device only), so a version key cannot be trusted while training;
    * every train()/eval() switch and `invalidate_packed()` drop the cache, so the first eval-mode render after (graph-replayed)
      training packs the current weights.  Call `invalidate_packed()` yourself when parameters are modified behind torch's back while
      the module stays in eval mode.
    """

    _net_id: int = -1

    def _linear_layers(self) -> List[torch.nn.Linear]:
        raise NotImplementedError

    def _layers(self) -> List[torch.nn.Linear]:
        """`_linear_layers()` resolved ONCE per module: the lookup walks nn.Module.__getattr__ and nn.Sequential.__getitem__ for every layer
        (~300 attribute look-ups per training step over both networks, 7 % of the eager 1 024-ray iteration's host time,
        scripts/gpu_host_profile.py).  The layer OBJECTS of a network never change after construction -- load_state_dict, .to() and the
        optimizers write into the same Parameters; a caller who swaps a layer object for another calls `_layers_changed()`."""
        c = self.__dict__.get("_layers_cache")
        if c is None:
            c = self.__dict__["_layers_cache"] = list(self._linear_layers())
        return c

    def _layer_params(self):
        """([weights], [biases]) of `_layers()`, resolved once like the layers themselves (each `l.weight` is an nn.Module.__getattr__ call)"""
        c = self.__dict__.get("_layer_params_cache")
        layers = self._layers()
        # (Parameters are written in place by load_state_dict / .to() / the optimizers; a caller that ASSIGNS new Parameter objects to the
        #  layers is caught by the identity check of the first and the last one -- anything finer calls `_layers_changed()`)
        if c is None or c[0][0] is not layers[0].weight or c[1][-1] is not layers[-1].bias:
            c = self.__dict__["_layer_params_cache"] = ([l.weight for l in layers], [l.bias for l in layers])
        return c

    def _layers_changed(self) -> None:
        self.__dict__.pop("_layers_cache", None)
        self.__dict__.pop("_layer_params_cache", None)
        self.invalidate_packed()

    # ---- narrower networks (--prop_net_width / --nerf_net_width < 256) ---------------------------------------------------------------
    # The kernels are compiled for 256-wide hidden layers.  A network with hidden_unit < 256 is evaluated EXACTLY by them with its
    # tensors zero-padded to the compiled shapes: the padded units have zero weights and zero bias, so they output relu(0) = 0 and feed
    # zeros forward, and an fp32 (or bf16 x bf16 -> fp32) accumulation is unchanged by added zero products; the same holds for the
    # backward (the padded rows / columns of every gradient are the discarded part).  Hidden features are always the LAST column segment
    # of a layer's input (cat(encoding, hidden)), so padding is at the end of each dimension.  It runs at the 256-wide speed: this is
    # the compatibility path of the width flags, not a narrow kernel.
    def _kernel_weight_shapes(self):
        """(out, in) of every tensor in _linear_layers() order as the kernels expect them; None = the module's own shapes"""
        return None

    # ---- other positional-encoding depths / cat_origin=False ---------------------------------------------------------------------------
    # The kernels evaluate the encoding [x | sin 2^0 x | cos 2^0 x | ... | sin 2^9 x | cos 2^9 x] (nerf_helper.py:38-48 behind the raw
    # position, mip_model.py:50-52) in registers.  A module built with FEWER octaves, or without the raw position (cat_origin=False), is
    # the same function as the compiled one with ZERO weights on the encoding columns it does not have -- every encoding value is finite,
    # so the added products are exact zeros.  Its weight columns are therefore not padded at the end but placed: `_column_segments()` lists,
    # per tensor, (kernel column, module column, count) runs; gradients come back through the same runs.
    def _column_segments(self):
        """per tensor of _linear_layers(): None (columns are padded at the end) or [(kernel column, module column, count), ...]"""
        return None

    @staticmethod
    def encoding_segment(levels: int, cat_origin: bool, kernel_col: int = 0, module_col: int = 0):
        """the run of one [raw 3 | 6 L] encoding block: L octaves are the first 6 L encoding columns of any deeper encoding"""
        return (kernel_col, module_col, 3 + 6 * levels) if cat_origin else (kernel_col + 3, module_col, 6 * levels)

    def kernel_params(self, shapes=None):
        """-> (weights, biases) in the kernels' shapes (the parameters themselves when nothing is padded); `shapes` overrides
        _kernel_weight_shapes() (the narrow-tile layout of a network)"""
        shapes = self._kernel_weight_shapes() if shapes is None else shapes
        ws, bs = self._layer_params()
        if shapes is None or all(tuple(w.shape) == tuple(s) for w, s in zip(ws, shapes)):
            return list(ws), list(bs)
        pw, pb = [], []
        segs = self._column_segments() or [None] * len(ws)
        with torch.no_grad():
            for w, b, s, seg in zip(ws, bs, shapes, segs):
                if tuple(w.shape) == tuple(s):
                    pw.append(w); pb.append(b)
                    continue
                W = torch.zeros(tuple(s), dtype=w.dtype, device=w.device)
                if seg is None:
                    W[: w.shape[0], : w.shape[1]] = w
                else:
                    for kc, mc, n in seg:
                        W[: w.shape[0], kc: kc + n] = w[:, mc: mc + n]
                B = torch.zeros((s[0],), dtype=b.dtype, device=b.device)
                B[: b.shape[0]] = b
                pw.append(W); pb.append(B)
        return pw, pb

    def unpad_grads(self, gW, gb):
        """gradients in the kernels' shapes -> the parameters' shapes"""
        layers = self._layers()
        segs = self._column_segments() or [None] * len(layers)

        def cols(g, l, seg):
            if tuple(g.shape) == tuple(l.weight.shape):
                return g
            if seg is None:
                return g[: l.weight.shape[0], : l.weight.shape[1]]
            return torch.cat([g[: l.weight.shape[0], kc: kc + n] for kc, _, n in seg], dim=1)
        return ([cols(g, l, seg) for g, l, seg in zip(gW, layers, segs)],
                [g[: l.bias.shape[0]] if tuple(g.shape) != tuple(l.bias.shape) else g for g, l in zip(gb, layers)])

    # ---- persistent gradient sinks ---------------------------------------------------------------------------------------------------
    # nerf_amd.parallel.FlatGradients keeps ONE flat fp32 buffer over the parameters of both networks and makes every `p.grad` a view of
    # it.  A module it is attached to has its weight-gradient kernels write straight into those views (the finalize kernels overwrite
    # every element) and reports "no gradient" to autograd: no per-tensor accumulate launches, no torch.cat before the all-reduce, and the
    # optimizer walks the same buffer.  Modules wrapped in DistributedDataParallel are NOT attached: there autograd must see the gradients
    # (DDP's hooks hang off the AccumulateGrad nodes).
    # ---- shapes LARGER than the compiled ones (hidden width > 256, > 10 octaves): nerf_amd/generic_path.py, layer by layer -----------------
    def _generic(self) -> bool:
        return False

    def grad_sinks(self):
        """-> (weight views, bias views, overwrite?) or None (not attached / zero-padded narrow network: the ordinary autograd path)"""
        owner = self.__dict__.get("_grad_owner")
        if owner is None or self._generic():
            return None
        layers = self._layers()
        shapes = self._kernel_weight_shapes()
        if shapes is not None and any(tuple(l.weight.shape) != tuple(sh) for l, sh in zip(layers, shapes)):
            return None
        return owner.sinks_for(self, layers)

    # ---- narrow-tile layouts -------------------------------------------------------------------------------------------------------
    # A module may have a second packed layout evaluated by a narrower kernel (ProposalNetwork with hidden_unit <= 128: NET_PROPOSAL_128,
    # a quarter of the 256-wide MACs).  It is a FORWARD layout: `packed(precision)` hands it to the eval / render entry points (the blob
    # carries its layout flag, which ops.* OR into the call's precision argument); `packed(precision, wide=True)` is the 256-wide blob the
    # training forward and the backward kernels take.
    def _narrow_layout(self) -> bool:
        return False

    def _pack_now(self, precision: int, narrow: bool = False) -> torch.Tensor:
        ws, bs = self.kernel_params()
        return ops.pack_weights(self._net_id, precision, ws, bs)

    def packed_backward(self, precision: int) -> torch.Tensor:
        """The transposed weights for the dgrad chain (nerf_amd_pack_weights_backward); packed when asked for -- the backward runs once
        per training step, after which the weights change anyway."""
        return ops.pack_weights_backward(self._net_id, precision, self.kernel_params()[0])

    def _packed_key(self):
        return (ops.PARAM_GENERATION[0],) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def invalidate_packed(self) -> None:
        self.__dict__.setdefault("_packed_cache", {}).clear()
        self.__dict__.setdefault("_rows_packed", {}).clear()          # the bf16-rows route's per-layer packing (generic_path._packed)

    def train(self, mode: bool = True):
        self.invalidate_packed()
        return super().train(mode)

    def packed(self, precision: int, wide: bool = False) -> torch.Tensor:
        narrow = (not wide) and self._narrow_layout()
        cache = self.__dict__.setdefault("_packed_cache", {})
        if self.training:
            cache.clear()
            return self._pack_now(precision, narrow)
        key = self._packed_key()
        hit = cache.get((precision, narrow))
        if hit is None or hit[0] != key:
            blob = self._pack_now(precision, narrow)
            cache[(precision, narrow)] = (key, blob)
            return blob
        return hit[1]


def require_no_grad(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(
            "nerf_amd: the HIP path is forward-only in this round (backward = SURVEY.md section 8f-1); "
            "wrap the call in torch.no_grad()")
