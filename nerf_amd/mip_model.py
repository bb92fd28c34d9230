"""Host-side mirror of the reference's ``nerf/mip_model.py``.  The module tree (and therefore every
``state_dict`` key and weight shape) is identical to the reference's, so checkpoints interchange;
``forward`` runs the fused HIP MLP kernel instead of eleven aten GEMMs."""
import torch
from torch import nn

from . import ops
from . import autograd_bridge as ab
from ._packed import PackedWeightsMixin, require_no_grad
from .nerf_base import NeRF
from .nerf_helper import makeMLP


class MipNeRF(PackedWeightsMixin, NeRF):
    _net_id = ops.NET_MIP
    _supports_grad_sinks = True          # the weight-gradient kernels can write into parallel.FlatGradients views

    def __init__(self, position_flevel, direction_flevel, hidden_unit=256, cat_origin=True) -> None:
        super().__init__(position_flevel, cat_origin)
        self.direction_flevel = direction_flevel
        self.hidden_unit = hidden_unit
        extra = 3 if cat_origin else 0
        in_dim = 6 * position_flevel + extra
        block1 = makeMLP(in_dim, hidden_unit)
        for _ in range(3):
            block1.extend(makeMLP(hidden_unit, hidden_unit))
        self.lin_block1 = nn.Sequential(*block1)                                   # before the skip connection
        self.lin_block2 = nn.Sequential(*makeMLP(hidden_unit + in_dim, hidden_unit), *makeMLP(hidden_unit, hidden_unit),
                                        *makeMLP(hidden_unit, 256))
        self.bottle_neck = nn.Sequential(*makeMLP(256, 256, None))
        self.opacity_head = nn.Sequential(*makeMLP(256, 1, None))
        self.rgb_layer = nn.Sequential(*makeMLP(280 + extra, 128), *makeMLP(128, 3, nn.Sigmoid()))
        self.apply(self.init_weight)

    def _linear_layers(self):
        return [self.lin_block1[0], self.lin_block1[2], self.lin_block1[4], self.lin_block1[6],
                self.lin_block2[0], self.lin_block2[2], self.lin_block2[4], self.bottle_neck[0], self.opacity_head[0],
                self.rgb_layer[0], self.rgb_layer[2]]

    def _check_config(self):
        # position_flevel < 10 and cat_origin=False run on the same kernels with zero weights on the encoding columns the module lacks
        # (_packed.py, `_column_segments`); the direction depth is 4 in the reference itself: rgb_layer.0 is built 280 (+3) wide
        # (mip_model.py:34), so its own forward only runs with direction_flevel == 4.
        # Shapes LARGER than the compiled ones (`--nerf_net_width` above 256, more than 10 octaves) run layer by layer on the generic MFMA GEMM
        # (nerf_amd/generic_path.py), forward and backward.
        if not (self.position_flevel >= 1 and self.direction_flevel == 4 and self.hidden_unit >= 1):
            raise NotImplementedError("nerf_amd: MipNeRF needs direction_flevel == 4 (the reference's own rgb_layer.0 is built for it, mip_model.py:34)")

    def _generic(self) -> bool:
        return self.hidden_unit > 256 or self.position_flevel > 10

    def _column_segments(self):
        if self.position_flevel == 10 and self.cat_origin:
            return None
        L, cat, W = self.position_flevel, self.cat_origin, self.hidden_unit
        enc = 6 * L + (3 if cat else 0)
        seg = self.encoding_segment
        return [[seg(L, cat)], None, None, None, [seg(L, cat), (63, enc, W)], None, None, None, None,
                None if cat else [(0, 0, 256), seg(4, False, 256, 256)], None]

    def _kernel_weight_shapes(self):
        return [(256, 63), (256, 256), (256, 256), (256, 256), (256, 319), (256, 256), (256, 256), (256, 256), (1, 256), (128, 283), (3, 128)]

    # hidden_unit <= 128 (`--nerf_net_width 128`): the narrow-tile kernel on its own packed layout (NERF_AMD_NET_MIP_128) for forward-only
    # calls with the point PE; widths below 128 are zero-padded to it.  lin_block2.4 / bottle_neck / heads are 256 wide at every hidden width.
    _NARROW_SHAPES = [(128, 63), (128, 128), (128, 128), (128, 128), (128, 191), (128, 128), (256, 128), (256, 256), (1, 256), (128, 283), (3, 128)]

    def _narrow_layout(self) -> bool:
        return self.hidden_unit <= 128

    def _pack_now(self, precision: int, narrow: bool = False) -> torch.Tensor:
        if not narrow:
            return super()._pack_now(precision)
        ws, bs = self.kernel_params(self._NARROW_SHAPES)
        blob = ops.pack_weights(ops.NET_MIP_128, precision, ws, bs)
        blob._nerf_amd_layout = ops.FINE_W128
        return blob

    def _params(self):
        ws, bs = self._layer_params()
        return ws + bs

    def _train_op(self, prec, run_forward, *tensors):
        """Training forward (activation dump) + the hand-written backward on it (mlp_backward.py) as one autograd node.
        `run_forward(training precision code)` -> (rgbo, dump)."""
        from . import mlp_backward
        held = {}
        n_in = len(tensors)
        tprec = ops.train_precision(prec)                                # (bf16 arithmetic with fp8 dumps: ops.set_train_dumps)

        def hip(*args):
            out, held["dump"] = run_forward(tprec)
            held["out"] = out.detach()                                   # (an alias without grad_fn: `out` itself would close a cycle out -> grad_fn -> held)
            return out

        def bwd(g, *args):
            if "dump" not in held:
                raise RuntimeError("nerf_amd: the activation dump of this forward was already consumed (backward twice over the same graph)")
            kw, kb = self.kernel_params()
            sinks = self.grad_sinks()                                    # persistent flat gradient buffer (parallel.FlatGradients)?
            direct = sinks is not None and sinks[2]
            gW, gb = mlp_backward.mip_backward(g.reshape(-1, 4), held.pop("out").reshape(-1, 4), None, held.pop("dump"), tprec,
                                               kw, kb, packed_bwd=ops.pack_weights_backward(ops.NET_MIP, prec, kw),
                                               out=(sinks[0], sinks[1]) if direct else None)
            if sinks is not None:
                if not direct:                                           # a second backward in the same step accumulates
                    torch._foreach_add_(list(sinks[0]) + list(sinks[1]), list(gW) + list(gb))
                return (*[None] * (n_in + len(gW) + len(gb)),)
            gW, gb = self.unpad_grads(gW, gb)
            return (*[None] * n_in, *gW, *gb)
        return ab.HipOp.apply(hip, bwd, 1, *tensors, *self._params())

    def forward(self, pts: torch.Tensor, contract: bool = False) -> torch.Tensor:
        """pts (N,S,6) = [position | raw direction] -> (N,S,4) = [sigmoid rgb | raw sigma]  (mip_model.py:41-60).
        `contract` (not in the reference; BASELINE configs[4]): Mip-NeRF 360 scene contraction of the positions before the encoding."""
        self._check_config()
        if self._generic():
            from . import generic_path
            return generic_path.mip_forward(self, pts, contract=contract)
        prec = ops.current_precision()
        params = self._params()
        if ab.needs_grad(pts, *params):
            if pts.requires_grad:                                          # the reference's loss never differentiates the fine positions (utils.py:35-36)
                ab.unsupported("MipNeRF.forward with sample positions that require a gradient")
            if pts.numel() == 0:                                           # an empty batch: nothing to launch, zero gradients for every parameter
                return ops.mip_forward(self.packed(prec), prec, pts, contract=contract) + sum(q.sum() for q in params) * 0.0
            # parameter gradients: the training forward dumps the hidden activations, the backward is hand-written kernels on them (mlp_backward.py)
            return self._train_op(prec, lambda tp: ops.mip_forward_train(self.packed(prec, wide=True), tp, pts.detach(), contract=contract), pts)
        return ops.mip_forward(self.packed(prec), prec, pts, contract=contract)

    def forward_rays(self, rays: torch.Tensor, z: torch.Tensor, n_samples: int, ipe_radius=None, ipe_dir_norm: torch.Tensor = None,
                     contract: bool = False) -> torch.Tensor:
        """Same as ``forward(NeRF.length2pts(rays, z[:, :n_samples]))`` without materialising the points -- and the entry of the two
        sample-fetch modes the reference has no caller for:
          * `ipe_radius` (BASELINE configs[2]): integrated PE (mip_methods.py:15-58) of the `n_samples` conical frusta between consecutive
            depths (z has >= n_samples + 1 columns); `ipe_dir_norm` = ops.dirs_norm(rays) (default: computed here);
          * `contract` (configs[4]): scene contraction of the sample positions (of the frustum means with IPE).
        Differentiable w.r.t. the parameters (training forward + HIP backward); rays and depths carry no gradient, like the reference's
        detached fine depths (utils.py:35-36)."""
        self._check_config()
        prec = ops.current_precision()
        if rays.requires_grad or z.requires_grad:
            raise NotImplementedError("nerf_amd: MipNeRF.forward_rays differentiates the parameters only (rays / depths must not require grad)")
        if self._generic():
            if ipe_radius is not None:
                # layer-by-layer route: the stand-alone integrated-PE encoder (nerf_amd_ipe_feature) feeds [frustum mean | feature] to the layers
                # (with `contract` the encoder contracts the frustum mean itself -- nerf_amd_ipe_feature_contracted, round 6; the layers then
                #  see [contracted mean | feature] and must NOT contract again)
                from . import generic_path
                feat, mu, _ = ops.ipe_feature(z[:, : n_samples + 1].contiguous(), rays, self.position_flevel, float(ipe_radius), ipe_dir_norm,
                                              contract=bool(contract))
                pts = torch.cat((mu, rays[:, None, 3:6].expand(-1, n_samples, -1)), dim=-1).contiguous()
                return generic_path.mip_forward(self, pts, encoded_x=feat)
            return self.forward(NeRF.length2pts(rays, z[:, :n_samples].contiguous()), contract=contract)
        rays, z = ops._dev(rays, "rays"), ops._dev(z, "z")
        if ipe_radius is not None and ipe_dir_norm is None:
            ipe_dir_norm = ops.dirs_norm(rays)
        s = ops.samples_rays(rays, n_samples, z=z, contract=contract, ipe_radius=ipe_radius, ipe_dir_norm=ipe_dir_norm)
        shape = (rays.shape[0], n_samples)
        if ab.needs_grad(*self._params()) and rays.shape[0] > 0:
            keep = (rays, z, ipe_dir_norm)                               # the descriptor holds raw pointers: keep the tensors alive
            return self._train_op(prec, lambda tp: (ops.mip_forward_train_samples(self.packed(prec, wide=True), tp, s, shape, rays.device), keep)[0])
        return ops.mip_forward_samples(self.packed(prec, wide=ipe_radius is not None), prec, s, shape, rays.device)
