"""Host-side mirror of the reference's ``nerf/nerf_base.py``: same class, statics, defaults and return
arity; the arithmetic runs in the HIP kernels of libnerf_amd.so."""
from typing import Optional, Tuple

import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from . import autograd_bridge as ab
from ._packed import require_no_grad


def _act_code(density_act):
    """Map the reference's callable ``density_act`` onto a kernel activation code; unknown callables are
    applied on the device with torch first and the kernel then uses identity."""
    if density_act in (F.relu, torch.relu):
        return ops.ACT_RELU, None
    if density_act is F.softplus:
        return ops.ACT_SOFTPLUS, None
    return ops.ACT_IDENTITY, density_act


class NeRF(nn.Module):
    @staticmethod
    def init_weight(m):
        """`module.apply` hook (nerf_base.py:15-22): Linear -> trunc-normal(std 0.02) weights and zero bias, BatchNorm1d -> (1, 0).
        Only the trunc-normal draw consumes random numbers, so seeded initialisation matches the reference's."""
        with torch.no_grad():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    m.bias.zero_()
            elif isinstance(m, nn.BatchNorm1d):
                m.weight.fill_(1.0)
                m.bias.zero_()

    def __init__(self, position_flevel, cat_origin=True, density_act=F.relu) -> None:
        super().__init__()
        self.position_flevel = position_flevel
        self.cat_origin = cat_origin
        self.density_act = density_act

    def loadFromFile(self, load_path: str, use_amp=False, opt=None, other_stuff=None):
        """Checkpoint loader: strips DDP's ``module.`` prefix, restores the optimizer (nerf_base.py:30-50)."""
        save = torch.load(load_path, map_location="cpu")
        stripped = {(k[7:] if k.startswith("module") else k): v for k, v in save["model"].items()}
        own = self.state_dict()
        own.update({k: stripped[k] for k in own.keys()})
        self.load_state_dict(own)
        if opt is not None:
            opt.load_state_dict(save["optimizer"])
        if use_amp:
            from apex import amp
            amp.load_state_dict(save["amp"])
        print("NeRF Model loaded from '%s'" % (load_path))
        if other_stuff is not None:
            return [save[k] for k in other_stuff]

    @staticmethod
    def length2pts(rays: torch.Tensor, f_zvals: torch.Tensor) -> torch.Tensor:
        """(N,S,6) = [o + z d | d]  (nerf_base.py:53-56)."""
        require_no_grad(rays, f_zvals)
        return ops.length2pts(rays, f_zvals)

    @staticmethod
    def coarseFineMerge(rays: torch.Tensor, c_zvals: torch.Tensor, f_zvals: torch.Tensor, f_inds: Optional[torch.Tensor] = None):
        """cat(fine, coarse) -> sort -> drop last -> points (nerf_base.py:59-73).  On the device the sort, the index bookkeeping
        (arange / cat / gather) and the points are two HIP launches (nerf_amd_merge_depths_order: a stable merge of the two ascending
        depth sets, out-of-order rays sorted first; nerf_amd_length2pts).  CPU tensors -- host-side unit tests only -- take the torch
        expression of the same definition."""
        require_no_grad(rays, c_zvals, f_zvals)
        if rays.is_cuda:
            z, order, all_inds = ops.merge_depths_order(f_zvals, c_zvals, f_inds)
            samples = ops.length2pts(rays, z)
            if f_inds is not None:
                return samples, z, all_inds, order[..., :-1]
            return samples, z,
        z, order = torch.sort(torch.cat((f_zvals, c_zvals), dim=-1), dim=-1, stable=True)
        if f_inds is not None:
            c_inds = torch.arange(c_zvals.shape[-1], device=z.device).unsqueeze(0).expand(c_zvals.shape[0], -1)
            all_inds = torch.gather(torch.cat((f_inds, c_inds), dim=-1), -1, order)
        z = z[..., :-1].contiguous()
        samples = ops.length2pts(rays, z)
        if f_inds is not None:
            return samples, z, all_inds, order[..., :-1]
        return samples, z,

    @staticmethod
    def getNormedWeight(opacity: torch.Tensor, depth: torch.Tensor, density_act=F.relu) -> torch.Tensor:
        """alpha_i * prod_{j<i}(1 - alpha_j + 1e-10), delta_last = 1e10 (nerf_base.py:80-86)."""
        code, pre = _act_code(density_act)
        if pre is not None:
            opacity = pre(opacity)
        if ab.needs_grad(opacity, depth):
            if opacity.shape[-1] > ops.BWD_MAX_SAMPLES:
                ab.unsupported("a differentiable sigma -> weights row of %d samples (the backward kernel keeps a ray in registers: <= %d)" % (opacity.shape[-1], ops.BWD_MAX_SAMPLES))
            return ab.HipOp.apply(lambda s, z: ops.sigma_to_weights(s, z, None, code), lambda g, s, z: (ops.sigma_to_weights_backward(s, z, None, code, g), None),
                                  1, opacity, depth)
        return ops.sigma_to_weights(opacity, depth, None, code)

    @staticmethod
    def render(rgbo: torch.Tensor, depth: torch.Tensor, ray_dirs: torch.Tensor, mul_norm: bool = True,
               white_bkg: bool = False, density_act=F.relu, render_depth: Optional[Tuple[float, float]] = None,
               normal_info: Optional[Tuple] = None):
        """Alpha compositing (nerf_base.py:91-113) -> (rgb (N,3), weights (N,S), extras)."""
        code, pre = _act_code(density_act)
        if pre is not None:
            rgbo = torch.cat((rgbo[..., :3], pre(rgbo[..., 3:])), dim=-1)
        normal, cam_dir = (normal_info if normal_info is not None else (None, None))
        if ab.needs_grad(rgbo, depth):
            # training path (train.py:190,196): rgb AND weights are differentiable w.r.t. the network output (the Ref-NeRF step feeds
            # the weights into WeightedNormalLoss / BackFaceLoss un-detached, train.py:183-184); extras stay forward-only
            def hip(r, z, dd):
                rgb_, w_, _, _ = ops.composite(r, z, dd, mul_norm == True, bool(white_bkg), code, None)
                return rgb_, w_

            if rgbo.shape[1] > ops.BWD_MAX_SAMPLES:
                ab.unsupported("differentiable compositing of %d samples per ray (the backward kernel keeps a ray in registers: <= %d)" % (rgbo.shape[1], ops.BWD_MAX_SAMPLES))
            rgb, w = ab.HipOp.apply(hip, lambda g, r, z, dd: (ops.composite_backward(r, z, dd, mul_norm == True, bool(white_bkg), code, None, g[0], g[1], None),
                                                              None, None), 2, rgbo, depth, ray_dirs)
            extras = dict()
            if render_depth is not None or normal_info is not None:
                with torch.no_grad():
                    _, _, d, nimg = ops.composite(rgbo.detach(), depth, ray_dirs, mul_norm == True, bool(white_bkg), code, render_depth,
                                                  normal, cam_dir, want_weights=False)
                if render_depth is not None:
                    extras["depth_img"] = d
                if normal_info is not None:
                    extras["normal_img"] = nimg
            return rgb, w, extras
        rgb, w, d, nimg = ops.composite(rgbo, depth, ray_dirs, mul_norm == True, bool(white_bkg), code, render_depth, normal, cam_dir)
        extras = dict()
        if render_depth is not None:
            extras["depth_img"] = d
        if normal_info is not None:
            extras["normal_img"] = nimg
        return rgb, w, extras


class DecayLrScheduler:
    """Linear warm-up then exponential decay with a floor (nerf_base.py:115-134)."""

    def __init__(self, min_r, decay_r, step, lr, warmup_step=0):
        self.min_ratio, self.decay_rate, self.decay_step = min_r, decay_r, step
        self.warmup_step, self.lr = warmup_step, lr
        if warmup_step > 0:
            print("Warming up step: %d" % (warmup_step))

    def lr_at(self, train_cnt) -> float:
        """Learning rate of iteration `train_cnt` (the reference's expressions, term for term, so that the floats agree)."""
        if train_cnt < self.warmup_step:
            r = train_cnt / self.warmup_step
            return self.lr * (self.min_ratio * (1. - r) + r)
        decayed = self.decay_rate ** ((train_cnt - self.warmup_step) / self.decay_step)
        return self.lr * max(decayed, self.min_ratio)

    def update_opt_lr(self, train_cnt, opt: torch.optim.Optimizer = None):
        new_lr = self.lr_at(train_cnt)
        for group in (opt.param_groups if opt is not None else ()):
            group['lr'] = new_lr
        return opt, new_lr
