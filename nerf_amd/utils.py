"""Host-side mirror of the reference's ``nerf/utils.py``: same names, argument meaning and RNG
protocol (every random draw comes from the CPU default generator in the reference's order and is
then moved to the device, utils.py:76,89,115), with the arithmetic in HIP kernels."""
import os
import shutil
from collections.abc import Iterable
from datetime import datetime

import numpy as np
import torch

from . import ops


def getSummaryWriter(epochs: int, del_dir: bool):
    """TensorBoard writer under ./logs/<timestamp>-epoch<n>/ (utils.py:15-20)."""
    from torch.utils.tensorboard import SummaryWriter
    logdir = './logs/'
    if os.path.exists(logdir) and del_dir:
        shutil.rmtree(logdir)
    stamp = "{0:%Y-%m-%d/%H-%M-%S}-epoch{1}/".format(datetime.now(), epochs)
    return SummaryWriter(log_dir=logdir + stamp)


def _focal_xy(focal):
    """(fx, fy): with a tuple focal the x coordinate is divided by focal[1] and y by focal[0] (utils.py:79-81)."""
    if isinstance(focal, Iterable):
        return float(focal[1]), float(focal[0])
    return float(focal), float(focal)


def inverseSample(weights: torch.Tensor, coarse_depth: torch.Tensor, sample_pnum: int, sort: bool = False, u: torch.Tensor = None):
    """Inverse-transform sampling of ``sample_pnum`` depths from the proposal histogram (utils.py:34-44).
    Returns z (and, when ``sort``, the ``below`` bin indices gathered by the sort permutation).
    ``u`` (N, sample_pnum) may be injected; by default it is drawn like the reference does (CPU generator)."""
    weights = weights.detach()
    if u is None:
        u = torch.rand(list(weights.shape[:-1]) + [sample_pnum])                  # utils.py:115
    u = u.to(weights.device)
    z, below = ops.inverse_sample(weights, coarse_depth, u, sort, want_below=sort)
    if sort:
        return z, below
    return z


def sample_pdf(bins, weights, N_samples, u: torch.Tensor = None):
    """utils.py:108-133 -> (samples, below, above)."""
    if u is None:
        u = torch.rand(list(weights.shape[:-1]) + [N_samples])
    return ops.sample_pdf(bins, weights, u.to(bins.device))


def randomFromOneImage(img: torch.Tensor, crop_xy: tuple):
    """Flattened pixel table and integer (col - W//2, H//2 - row) coordinates, optionally centre-cropped
    (utils.py:47-69).  Pure indexing -- stays a torch gather on the image's device."""
    if img.dim() > 3:
        img = img.squeeze(0)
    H, W = img.shape[1], img.shape[2]
    hw, hh = W // 2, H // 2
    x0, x1 = (int(hw * (1. - crop_xy[0])), int(hw + hw * crop_xy[0])) if crop_xy[0] < 9.9e-1 else (0, W)
    y0, y1 = (int(hh * (1. - crop_xy[1])), int(hh + hh * crop_xy[1])) if crop_xy[1] < 9.9e-1 else (0, H)
    rows, cols = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing='ij')
    coords = torch.stack((cols - hw, hh - rows), dim=-1).to(img.device).view(-1, 2)
    if crop_xy[0] < 9.9e-1 or crop_xy[1] < 9.9e-1:
        return img[:, rows, cols].view(3, -1).transpose(0, 1).contiguous(), coords
    return img.view(3, -1).transpose(0, 1).contiguous(), coords


def validSampler(rgbs: torch.Tensor, coords: torch.Tensor, cam_tf: torch.Tensor, ray_num: int, point_num: int, focal,
                 near: float, far: float, output_samples=True):
    """Random training rays + stratified coarse samples (utils.py:72-94).
    -> (pts (N,C,3), lengths (N,C), rgb (N,3), rays (N,6)) or (rgb, rays)."""
    dev = rgbs.device
    idx = torch.randint(0, coords.shape[0], (ray_num,)).to(dev)                  # CPU generator, like the reference
    rgb = rgbs[idx]
    fx, fy = _focal_xy(focal)
    rays = ops.pixel_rays(coords[idx], cam_tf, fx, fy)
    if not output_samples:
        return rgb, rays
    res = (far - near) / point_num
    base = torch.linspace(near, far - res, point_num).to(dev)
    u = torch.rand((ray_num, point_num)).to(dev)
    z, pts = ops.stratified_points(rays, base, u, res)
    return pts, z, rgb, rays


def fov2Focal(fov, img_size):
    """utils.py:96-105 (the square-image branch has no 1/2 -- reproduced, it is the caller's contract)."""
    if isinstance(fov, Iterable):
        if not isinstance(img_size, Iterable):
            raise ValueError("Error: If fov is iterable, img size should be iterable too, while we have typeof(img_size) =", type(img_size))
        return (0.5 * img_size[0] / np.tan(.5 * fov[1]), 0.5 * img_size[1] / np.tan(.5 * fov[0]))
    if img_size[0] == img_size[1]:
        img_size = img_size[0]
    f = img_size / np.tan(.5 * fov)
    return (f, f)


def pose_spherical(theta, phi, radius):
    """Orbit camera-to-world matrix (utils.py:136-159)."""
    ph, th = phi / 180. * np.pi, theta / 180. * np.pi
    t = torch.Tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]]).float()
    rp = torch.Tensor([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]]).float()
    rt = torch.Tensor([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]]).float()
    flip = torch.Tensor(np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]))
    return flip @ (rt @ (rp @ t))
