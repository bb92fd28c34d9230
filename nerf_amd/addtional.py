"""Host-side mirror of the reference's ``nerf/addtional.py`` (sic): ProposalNetwork, getBounds and the
scalar losses.  Same names, signatures and ``state_dict`` keys."""
import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from . import autograd_bridge as ab
from ._packed import PackedWeightsMixin, require_no_grad
from .nerf_helper import makeMLP


def getBounds(weights: torch.Tensor, inds: torch.Tensor):
    """Proposal weight mass covering each fine interval (addtional.py:14-18, index quirk included)."""
    if ab.needs_grad(weights):
        return ab.HipOp.apply(lambda w, i: ops.get_bounds(w, i), lambda g, w, i: (ops.get_bounds_backward(i, g, w.shape[-1]), None), 1, weights, inds)
    return ops.get_bounds(weights, inds)


class ProposalLoss(nn.Module):
    def forward(self, prop_bounds: torch.Tensor, nerf_weights: torch.Tensor) -> torch.Tensor:
        """sum relu(w - bound)^2 / (w + 1e-8)  (addtional.py:20-24)."""
        return torch.sum(F.relu(nerf_weights - prop_bounds) ** 2 / (nerf_weights + 1e-8))


class SoftL1Loss(nn.Module):
    def __init__(self, epsilon=0.001) -> None:
        super().__init__()
        self.eps = epsilon

    def forward(self, pred: torch.Tensor, target: torch.Tensor):
        """Despite the name: plain MSE (addtional.py:37-42)."""
        return torch.mean((pred - target) ** 2)


class LossPSNR(nn.Module):
    __LOG_10__ = 2.3025851249694824

    def forward(self, x):
        """-10 log10(mse)  (addtional.py:45-51)."""
        return -10. * torch.log(x) / LossPSNR.__LOG_10__


class ProposalNetwork(PackedWeightsMixin, nn.Module):
    _net_id = ops.NET_PROPOSAL
    _supports_grad_sinks = True          # the weight-gradient kernels can write into parallel.FlatGradients views

    @staticmethod
    def init_weight(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def __init__(self, position_flevel, hidden_unit=128, cat_origin=True) -> None:
        super().__init__()
        self.position_dims = position_flevel * 6
        self.position_flevel = position_flevel
        self.cat_origin = cat_origin
        self.hidden_unit = hidden_unit
        in_dim = self.position_dims + (3 if cat_origin else 0)
        self.layers = nn.Sequential(*makeMLP(in_dim, hidden_unit), *makeMLP(hidden_unit, hidden_unit),
                                    *makeMLP(hidden_unit, hidden_unit), *makeMLP(hidden_unit, hidden_unit),
                                    *makeMLP(hidden_unit, 1, None))
        self.apply(self.init_weight)

    def _linear_layers(self):
        return [self.layers[0], self.layers[2], self.layers[4], self.layers[6], self.layers[8]]

    def _check_config(self):
        # position_flevel < 10 / cat_origin=False: the same kernels with zero weights on the encoding columns the module lacks (_packed.py)
        # Shapes LARGER than the compiled ones (`--prop_net_width` above 256, more than 10 octaves): layer by layer on the generic MFMA GEMM
        # (nerf_amd/generic_path.py), forward and backward.
        if not (self.position_flevel >= 1 and self.hidden_unit >= 1):
            raise NotImplementedError("nerf_amd: ProposalNetwork needs position_flevel >= 1 and hidden_unit >= 1")

    def _generic(self) -> bool:
        return self.hidden_unit > 256 or self.position_flevel > 10

    def _column_segments(self):
        if self.position_flevel == 10 and self.cat_origin:
            return None
        return [[self.encoding_segment(self.position_flevel, self.cat_origin)], None, None, None, None]

    def _kernel_weight_shapes(self):
        return [(256, 63), (256, 256), (256, 256), (256, 256), (1, 256)]

    # hidden_unit <= 128 (the class default, addtional.py:61; `--prop_net_width 128`): the narrow-tile kernel on its own packed layout
    # (include/nerf_amd.h NERF_AMD_NET_PROPOSAL_128) for every forward-only call; widths below 128 are zero-padded to it (exact).
    _NARROW_SHAPES = [(128, 63), (128, 128), (128, 128), (128, 128), (1, 128)]

    def _narrow_layout(self) -> bool:
        return self.hidden_unit <= 128

    def _pack_now(self, precision: int, narrow: bool = False) -> torch.Tensor:
        if not narrow:
            return super()._pack_now(precision)
        ws, bs = self.kernel_params(self._NARROW_SHAPES)
        blob = ops.pack_weights(ops.NET_PROPOSAL_128, precision, ws, bs)
        blob._nerf_amd_layout = ops.PROP_W128                      # ops.* OR this into the precision argument of the calls that take the blob
        return blob

    def loadFromFile(self, load_path: str, use_amp=False, other_stuff=None):
        """addtional.py:73-86."""
        save = torch.load(load_path, map_location="cpu")
        own = self.state_dict()
        own.update({k: save["model"][k] for k in own.keys()})
        self.load_state_dict(own)
        if use_amp:
            from apex import amp
            amp.load_state_dict(save["amp"])
        print("NeRF Model loaded from '%s'" % (load_path))
        if other_stuff is not None:
            return [save[k] for k in other_stuff]

    def forward(self, pts: torch.Tensor, encoded_pt: torch.Tensor = None, contract: bool = False) -> torch.Tensor:
        """pts (N,C,3) -> density (N,C), no activation (addtional.py:88-96).  ``encoded_pt`` (a pre-computed
        encoding) is accepted for signature parity and ignored: the kernel encodes in-register.  ``contract`` (not in the reference;
        BASELINE configs[4]): Mip-NeRF 360 scene contraction of the positions before the encoding."""
        self._check_config()
        if self._generic():
            from . import generic_path
            return generic_path.proposal_forward(self, pts, contract=contract)
        prec = ops.current_precision()
        layers = self._layers()
        params = [l.weight for l in layers] + [l.bias for l in layers]
        if ab.needs_grad(pts, *params):
            if pts.numel() == 0:                                         # an empty batch: nothing to launch, zero gradients for every parameter
                return ops.proposal_forward(self.packed(prec), prec, pts.detach(), contract=contract) + sum(q.sum() for q in params) * 0.0
            # the training forward dumps the hidden activations; the backward is hand-written kernels on them:
            #   parameter gradients: fused dgrad chain + MFMA weight gradients (mlp_backward.py);
            #   RefNeRF.get_grad (train.py:165-168 with prop_normal: d density / d position): a dgrad-only chain down to the encoded
            #   position + the encoding's derivative (nerf_amd_density_grad).  The positions get a gradient only there.
            from . import mlp_backward
            held = {}
            # fp8 dumps (ops.set_train_dumps) unless the positions ask for a gradient: the density-gradient chain re-reads the bf16 activations
            tprec = prec if pts.requires_grad else ops.train_precision(prec)
            want_pos = bool(pts.requires_grad)

            def hip(p, *wb):
                out, held["dump"] = ops.proposal_forward_train(self.packed(prec, wide=True), tprec, p, contract=contract)
                return out

            def bwd(g, p, *wb):
                if "dump" not in held:
                    raise RuntimeError("nerf_amd: the activation dump of this forward was already consumed (backward twice over the same graph)")
                if "bwd_blob" not in held:
                    held["bwd_blob"] = self.packed_backward(prec)
                if ab._VJP.inputs_only:                                  # (contracted positions: through the contraction's Jacobian, round 5)
                    gx = ops.density_grad(ops.NET_PROPOSAL, held["bwd_blob"], prec, held["dump"], p.reshape(-1, 3), scale=g.reshape(-1), contract=contract)
                    return (gx.view(p.shape), *[None] * len(wb))
                gx = None
                if ab.POSITION_GRADS and want_pos:                       # d loss / d pts as well (autograd_bridge.POSITION_GRADS)
                    gx = ops.density_grad(ops.NET_PROPOSAL, held["bwd_blob"], prec, held["dump"], p.reshape(-1, 3), scale=g.reshape(-1), contract=contract).view(p.shape)
                sinks = self.grad_sinks()                                # persistent flat gradient buffer (parallel.FlatGradients)?
                direct = sinks is not None and sinks[2]
                kw = wb[:5]
                gW, gb = mlp_backward.proposal_backward(g.reshape(-1), p.reshape(-1, 3), held.pop("dump"), tprec, kw, packed_bwd=held["bwd_blob"],
                                                        out=(sinks[0], sinks[1]) if direct else None)
                if sinks is not None:
                    if not direct:                                       # a second backward in the same step accumulates
                        torch._foreach_add_(list(sinks[0]) + list(sinks[1]), list(gW) + list(gb))
                    return (gx, *[None] * len(wb))
                gW, gb = self.unpad_grads(gW, gb)
                return (gx, *gW, *gb)
            return ab.HipOp.apply(hip, bwd, 1, pts, *params)
        return ops.proposal_forward(self.packed(prec), prec, pts, contract=contract)

    @staticmethod
    def get_weights(density: torch.Tensor, zvals: torch.Tensor, ray_dirs: torch.Tensor = None) -> torch.Tensor:
        """relu(sigma) -> alpha -> exclusive transmittance product; z scaled by |d| when ray_dirs is
        given (addtional.py:100-107)."""
        if ab.needs_grad(density, zvals):
            if ray_dirs is not None:
                zvals = zvals * ray_dirs.norm(dim=-1, keepdim=True)
            if density.shape[-1] > ops.BWD_MAX_SAMPLES:
                ab.unsupported("a differentiable sigma -> weights row of %d samples (the backward kernel keeps a ray in registers: <= %d)" % (density.shape[-1], ops.BWD_MAX_SAMPLES))
            return ab.HipOp.apply(lambda s, z: ops.sigma_to_weights(s, z, None, ops.ACT_RELU),
                                  lambda g, s, z: (ops.sigma_to_weights_backward(s, z, None, ops.ACT_RELU, g), None), 1, density, zvals)
        return ops.sigma_to_weights(density, zvals, ray_dirs, ops.ACT_RELU)
