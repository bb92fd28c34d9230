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
        if not (self.position_flevel == 10 and self.direction_flevel == 4 and 1 <= self.hidden_unit <= 256 and self.cat_origin):
            raise NotImplementedError("nerf_amd: the HIP fine-MLP kernel is instantiated for MipNeRF(10, 4, hidden_unit <= 256, cat_origin=True)")

    def _kernel_weight_shapes(self):
        return [(256, 63), (256, 256), (256, 256), (256, 256), (256, 319), (256, 256), (256, 256), (256, 256), (1, 256), (128, 283), (3, 128)]

    def forward(self, pts: torch.Tensor) -> torch.Tensor:
        """pts (N,S,6) = [position | raw direction] -> (N,S,4) = [sigmoid rgb | raw sigma]  (mip_model.py:41-60)."""
        self._check_config()
        prec = ops.current_precision()
        layers = self._linear_layers()
        params = [l.weight for l in layers] + [l.bias for l in layers]
        if ab.needs_grad(pts, *params):
            n = len(layers)
            expr = lambda p, *wb: ab.mip_expr(p, wb[:n], wb[n:])
            if pts.requires_grad or pts.numel() == 0:                      # (gradients w.r.t. positions: torch VJP of the expression)
                hip = lambda p, *wb: ops.mip_forward(self.packed(prec), prec, p)
                return ab.HipOp.apply(hip, expr, 0, pts, *params)
            # parameter gradients: the training forward dumps the hidden activations, the backward is hand-written kernels on them (mlp_backward.py)
            from . import mlp_backward
            held = {}

            def hip(p, *wb):
                out, held["dump"] = ops.mip_forward_train(self.packed(prec), prec, p)
                held["out"] = out
                return out

            def bwd(g, p, *wb):
                if "dump" not in held:
                    raise RuntimeError("nerf_amd: the activation dump of this forward was already consumed (backward twice over the same graph)")
                kw, kb = self.kernel_params()
                gW, gb = mlp_backward.mip_backward(g.reshape(-1, 4), held.pop("out").reshape(-1, 4), p.reshape(-1, 6), held.pop("dump"), prec,
                                                   kw, kb, packed_bwd=ops.pack_weights_backward(ops.NET_MIP, prec, kw))
                gW, gb = self.unpad_grads(gW, gb)
                return (None, *gW, *gb)
            return ab.HipOp.apply(hip, ab.with_hip_backward(expr, bwd), 0, pts, *params)
        return ops.mip_forward(self.packed(prec), prec, pts)

    def forward_rays(self, rays: torch.Tensor, z: torch.Tensor, n_samples: int) -> torch.Tensor:
        """Same as ``forward(NeRF.length2pts(rays, z[:, :n_samples]))`` without materialising the points."""
        self._check_config()
        prec = ops.current_precision()
        s = ops.samples_rays(rays, n_samples, z=z)
        return ops.mip_forward_samples(self.packed(prec), prec, s, (rays.shape[0], n_samples), rays.device)
