"""Host-side mirror of the reference's ``nerf/mip_methods.py``."""
import torch

from . import ops
from . import autograd_bridge as ab
from ._packed import require_no_grad


def maxBlurFilter(weights: torch.Tensor, alpha: float):
    """2-tap max then 2-tap blur plus ``alpha`` (mip_methods.py:61-66) -- HIP kernel."""
    if ab.needs_grad(weights):
        return ab.HipOp.apply(lambda w: ops.max_blur(w, alpha), lambda g, w: (ops.max_blur_backward(w, g),), 1, weights)
    return ops.max_blur(weights, alpha)


def coneParameters(zvals: torch.Tensor, r: float):
    """Conical-frustum Gaussian moments along the ray (mip_methods.py:15-23) -> (mu_t, sigma_t^2, sigma_r^2), each (N, S) -- HIP kernel."""
    require_no_grad(zvals)
    return ops.cone_parameters(zvals, r)


def ipe_feature(zvals: torch.Tensor, cam_rays: torch.Tensor, freq_lvs: int, r: float):
    """Integrated positional encoding with diagonal covariance (mip_methods.py:47-58, helpers :15-45) -- HIP kernel
    (nerf_amd_ipe_feature; the whole-tensor direction norm of :31 comes from nerf_amd_dirs_norm).  zvals (N, S+1), cam_rays (N, 6)
    -> (feature (N, S, 6 freq_lvs), mu (N, S, 3), mu_t (N, S)).  Forward-only, like every encoding input of the reference."""
    require_no_grad(zvals, cam_rays)
    return ops.ipe_feature(zvals, cam_rays, freq_lvs, r)
