"""Host-side mirror of the reference's ``nerf/ref_func.py``: integrated directional encoding (IDE).

The fused RefNeRF kernel evaluates the IDE in registers from the coefficient table built here; the callable returned by
``generate_ide_fn`` keeps the reference's interface for stand-alone use and is a device-agnostic torch expression
(it is not on the render path)."""
import math

import numpy as np
import torch


def generalized_binomial_coeff(a, k):
    return np.prod(a - np.arange(k)) / math.factorial(k)


def assoc_legendre_coeff(l, m, k):
    """Coefficient of cos^k(theta) sin^m(theta) in P_l^m(cos theta)  (ref_func.py:14-30)."""
    return ((-1) ** m * 2 ** l * math.factorial(l) / math.factorial(k) / math.factorial(l - k - m) *
            generalized_binomial_coeff(0.5 * (l + k + m - 1.0), l))


def sph_harm_coeff(l, m, k):
    return (np.sqrt((2.0 * l + 1.0) * math.factorial(l - m) / (4.0 * np.pi * math.factorial(l + m))) * assoc_legendre_coeff(l, m, k))


def get_ml_array(deg_view):
    """All (m, l) pairs with l = 2^i, 0 <= m <= l  (ref_func.py:38-49)."""
    return np.array([(m, 2 ** i) for i in range(deg_view) for m in range(2 ** i + 1)]).T


def ide_table(deg_view: int) -> torch.Tensor:
    """(l_max+1, T) fp32 coefficient matrix of ref_func.py:60-74 (computed in float64, stored as float32)."""
    if deg_view > 5:
        raise ValueError('Only deg_view of at most 5 is numerically stable.')
    ml = get_ml_array(deg_view)
    mat = torch.zeros(2 ** (deg_view - 1) + 1, ml.shape[1])
    for i, (m, l) in enumerate(ml.T):
        for k in range(l - m + 1):
            mat[k, i] = sph_harm_coeff(l, m, k)
    return mat


def generate_ide_fn(deg_view):
    """Returns f(xyz (...,3), kappa_inv (...,1)) -> (..., 2T) = [real | imag]  (ref_func.py:51-110)."""
    ml = get_ml_array(deg_view)
    mat_cpu = ide_table(deg_view)

    on_device = {}                                            # the two constant tables, uploaded once per device (not per call)

    def integrated_dir_enc_fn(xyz, kappa_inv):
        if xyz.device not in on_device:
            on_device[xyz.device] = (mat_cpu.to(xyz.device), torch.from_numpy(ml).to(xyz.device))
        mat, ml_t = on_device[xyz.device]
        x, y, z = xyz[..., 0:1], xyz[..., 1:2], xyz[..., 2:3]
        vmz = torch.cat([z ** i for i in range(mat.shape[0])], dim=-1)
        vmxy = torch.cat([(x + 1j * y) ** m for m in ml_t[0, :]], dim=-1)       # (tensor exponents, like the reference)
        sph_harms = vmxy * (vmz @ mat)
        sigma = 0.5 * ml_t[1, :] * (ml_t[1, :] + 1)
        ide = sph_harms * torch.exp(-sigma * kappa_inv)
        return torch.cat([torch.real(ide), torch.imag(ide)], dim=-1)

    return integrated_dir_enc_fn
