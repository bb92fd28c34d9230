"""The torch SPECIFICATION of every differentiable op of nerf_amd: the reference's expressions (nerf_helper.py:38-48, addtional.py:88-107,
mip_model.py:41-60, ref_model.py:68-106, nerf_base.py:80-86, mip_methods.py:61-66, addtional.py:14-18) written as plain torch ops, so that
tests (and scripts/gpu_*_check.py) can ask torch.autograd for the gradients the hand-written HIP backward kernels must reproduce.

Test infrastructure, like oracle/: round 3 kept these expressions -- and a library-GEMM VJP built on them -- inside the product
(nerf_amd/autograd_bridge.py) as an opt-in generic backward.  The product now has HIP backward kernels or raises; nothing under nerf_amd/
imports this file."""
import math

import torch
import torch.nn.functional as F

ACT_RELU, ACT_IDENTITY, ACT_SOFTPLUS = 0, 1, 2          # nerf_amd.ops.ACT_*


def _pe(x: torch.Tensor, L: int) -> torch.Tensor:
    """[sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] (nerf_helper.py:38-48) in five device ops instead of 4L + 1."""
    freq = torch.pow(2.0, torch.arange(L, dtype=x.dtype, device=x.device))
    a = x.unsqueeze(-2) * freq[:, None]                              # (..., L, 3)
    return torch.stack((torch.sin(a), torch.cos(a)), dim=-2).reshape(x.shape[:-1] + (6 * L,))



def lin(x, w, b):
    return F.linear(x, w, b)


def lin_relu(x, w, b):
    return F.relu(F.linear(x, w, b))


def contract_expr(pts):
    """Mip-NeRF 360 scene contraction (eq. 10) of the position columns of (..., 3) / (..., 6) samples -- the kernels' `contract` flag."""
    x = pts[..., :3]
    n = x.norm(dim=-1, keepdim=True).clamp(min=1.0)                   # inside the unit ball: factor (2 - 1) / 1 = 1
    xc = x * ((2.0 - 1.0 / n) / n)
    return torch.cat((xc, pts[..., 3:]), dim=-1) if pts.shape[-1] > 3 else xc


def proposal_expr(pts, w, b):
    """ProposalNetwork.forward as torch ops (addtional.py:88-96); w, b = lists in state_dict order."""
    h = torch.cat((pts, _pe(pts, 10)), dim=-1)
    for i in range(4):
        h = lin_relu(h, w[i], b[i])
    return lin(h, w[4], b[4]).squeeze(-1)


def mip_expr(pts, w, b):
    """MipNeRF.forward as torch ops (mip_model.py:41-60); tensors in the order of MipNeRF._linear_layers()."""
    x, d = pts[..., :3], pts[..., 3:6]
    d = d / d.norm(dim=-1, keepdim=True)
    ex = torch.cat((x, _pe(x, 10)), dim=-1)
    ed = torch.cat((d, _pe(d, 4)), dim=-1)
    h = ex
    for i in range(4):
        h = lin_relu(h, w[i], b[i])
    g = torch.cat((ex, h), dim=-1)
    for i in range(4, 7):
        g = lin_relu(g, w[i], b[i])
    bott = lin(g, w[7], b[7])
    sigma = lin(g, w[8], b[8])
    c = lin_relu(torch.cat((bott, ed), dim=-1), w[9], b[9])
    rgb = torch.sigmoid(lin(c, w[10], b[10]))
    return torch.cat((rgb, sigma), dim=-1)


def ref_expr(pos, d, noise, P, ide_fn, use_srgb: bool = False):
    """RefNeRF.forward as torch ops (ref_model.py:68-106); P = {state_dict key: tensor}; `noise` = the train-mode
    perturbation of the bottle-neck vector or None.  Returns cat(rgb, density, normal) (..., 7)."""
    plin = lambda name, t: F.linear(t, P[name + ".weight"], P[name + ".bias"])
    plin_relu = lambda name, t: F.relu(F.linear(t, P[name + ".weight"], P[name + ".bias"]))
    ex = torch.cat((pos, _pe(pos, 10)), dim=-1)
    h = ex
    for i in (0, 2, 4, 6):
        h = plin_relu("spa_block1.%d" % i, h)
    g = torch.cat((ex, h), dim=-1)
    for i in (0, 2, 4, 6):
        g = plin_relu("spa_block2.%d" % i, g)
    normal, diffuse, tint = plin("norm_col_tint_head", g).split((3, 3, 3), dim=-1)
    rough, density = plin("rho_tau_head", g).split((1, 1), dim=-1)
    rough = F.softplus(rough - 1.0)
    b = plin("bottle_neck", g)
    if noise is not None:
        b = b + noise
    normal = -normal / (normal.norm(dim=-1, keepdim=True) + 1e-7)
    refl = d - 2.0 * torch.sum(d * normal, dim=-1, keepdim=True) * normal
    allin = torch.cat((b, ide_fn(refl, rough), torch.sum(normal * d, dim=-1, keepdim=True)), dim=-1)
    r = allin
    for i in (0, 2, 4, 6):
        r = plin_relu("dir_block1.%d" % i, r)
    r = torch.cat((allin, r), dim=-1)
    for i in (0, 2, 4, 6):
        r = plin_relu("dir_block2.%d" % i, r)
    spec = torch.sigmoid(plin("spec_rgb_head.0", r)) * torch.sigmoid(tint)
    if use_srgb:                                                                   # ref_model.py:100-102
        from nerf_amd.nerf_helper import linear_to_srgb
        rgb = linear_to_srgb(spec + torch.sigmoid(diffuse - math.log(3.0)))
    else:
        rgb = spec + torch.sigmoid(diffuse)
    return torch.cat((rgb, density, normal), dim=-1)


def weights_expr(sigma, z, act_code: int):
    """sigma -> alpha -> exclusive transmittance product (nerf_base.py:80-86); z already scaled."""
    big = torch.full((z.shape[0], 1), 1e10, dtype=z.dtype, device=z.device)
    delta = torch.cat((z[:, 1:] - z[:, :-1], big), dim=-1)
    dens = F.relu(sigma) if act_code == ACT_RELU else (F.softplus(sigma) if act_code == ACT_SOFTPLUS else sigma)
    m = torch.exp(-dens * delta)
    ones = torch.ones((z.shape[0], 1), dtype=z.dtype, device=z.device)
    T = torch.cumprod(torch.cat((ones, m + 1e-10), dim=-1), dim=-1)[:, :-1]
    return (1.0 - m) * T


def max_blur_expr(w, alpha):
    mx = torch.maximum(w[..., :-1], w[..., 1:])
    return 0.5 * (torch.cat((w[..., :1], mx), dim=-1) + torch.cat((mx, w[..., -1:]), dim=-1)) + alpha


def bounds_expr(w, inds):
    sat = torch.cat((torch.zeros(w.shape[0], 1, device=w.device), torch.cumsum(w, dim=-1)), dim=-1)
    return torch.gather(sat, -1, inds[:, 1:] + 1) - torch.gather(sat, -1, inds[:, :-1])


