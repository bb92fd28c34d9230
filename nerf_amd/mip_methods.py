"""Host-side mirror of the reference's ``nerf/mip_methods.py``."""
import torch

from . import ops
from . import autograd_bridge as ab


def maxBlurFilter(weights: torch.Tensor, alpha: float):
    """2-tap max then 2-tap blur plus ``alpha`` (mip_methods.py:61-66) -- HIP kernel."""
    if ab.needs_grad(weights):
        expr = ab.with_hip_backward(lambda w: ab.max_blur_expr(w, alpha), lambda g, w: (ops.max_blur_backward(w, g),))
        return ab.HipOp.apply(lambda w: ops.max_blur(w, alpha), expr, 0, weights)
    return ops.max_blur(weights, alpha)


def coneParameters(zvals: torch.Tensor, r: float):
    """Conical-frustum Gaussian moments along the ray (mip_methods.py:15-23; dead code in the reference)."""
    mid = (zvals[:, 1:] + zvals[:, :-1]) / 2
    hw2 = ((zvals[:, 1:] - zvals[:, :-1]) / 2) ** 2
    t = 3 * mid ** 2 + hw2
    mu_t = mid + 2 * mid * hw2 / t
    var_t = hw2 / 3 - 4 * (hw2 ** 2) * (12 * mid ** 2 - hw2) / 15 / (t ** 2)
    var_r = (r ** 2) * (0.25 * mid ** 2 + 5 / 12 * hw2 - 4 * hw2 ** 2 / (15 * t))
    return mu_t, var_t, var_r


def ipe_feature(zvals: torch.Tensor, cam_rays: torch.Tensor, freq_lvs: int, r: float):
    """Integrated positional encoding with diagonal covariance (mip_methods.py:47-58).  The reference
    never calls it; kept as device-agnostic torch expressions and pinned by golden G12."""
    mu_t, var_t, var_r = coneParameters(zvals, r)
    o, d = cam_rays[:, :3], cam_rays[:, 3:]
    mu = o[:, None, :] + mu_t[:, :, None] * d[:, None, :]
    dd = d * d
    perp = torch.ones(3, device=zvals.device)[None, :] - dd / d.norm()            # whole-tensor norm, as in the reference
    diag = var_t[:, :, None] * dd[:, None, :] + var_r[:, :, None] * perp[:, None, :]
    f2 = torch.tensor([2.0 ** i for i in range(freq_lvs)], device=zvals.device)
    f4 = torch.tensor([4.0 ** i for i in range(freq_lvs)], device=zvals.device)
    mu_r = f2[None, None, :, None] * mu[:, :, None, :]
    att = torch.exp(-0.5 * (f4[None, None, :, None] * diag[:, :, None, :]))
    n, s = mu.shape[0], mu.shape[1]
    feat = torch.cat((torch.sin(mu_r) * att, torch.cos(mu_r) * att), dim=-1).reshape(n, s, -1)
    return feat, mu, mu_t
