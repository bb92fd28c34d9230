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


_COORD_TABLES = {}


def randomFromOneImage(img: torch.Tensor, crop_xy: tuple):
    """Flattened pixel table and integer (col - W//2, H//2 - row) coordinates, optionally centre-cropped
    (utils.py:47-69).  Pure indexing -- stays a torch gather on the image's device.  The coordinate table depends only on
    (H, W, crop, device): it is built once and cached (the reference rebuilds a full H x W meshgrid on the CPU and copies it to the
    device every training iteration, train.py:153-157)."""
    if img.dim() > 3:
        img = img.squeeze(0)
    H, W = img.shape[1], img.shape[2]
    hw, hh = W // 2, H // 2
    x0, x1 = (int(hw * (1. - crop_xy[0])), int(hw + hw * crop_xy[0])) if crop_xy[0] < 9.9e-1 else (0, W)
    y0, y1 = (int(hh * (1. - crop_xy[1])), int(hh + hh * crop_xy[1])) if crop_xy[1] < 9.9e-1 else (0, H)
    key = (H, W, x0, x1, y0, y1, str(img.device))
    hit = _COORD_TABLES.get(key)
    if hit is None:
        rows, cols = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing='ij')
        coords = torch.stack((cols - hw, hh - rows), dim=-1).to(img.device).view(-1, 2)
        hit = (rows.to(img.device), cols.to(img.device), coords)
        if len(_COORD_TABLES) > 64:
            _COORD_TABLES.clear()
        _COORD_TABLES[key] = hit
    rows, cols, coords = hit
    if crop_xy[0] < 9.9e-1 or crop_xy[1] < 9.9e-1:
        return img[:, rows, cols].view(3, -1).transpose(0, 1).contiguous(), coords
    return img.view(3, -1).transpose(0, 1).contiguous(), coords


def validSampler(rgbs: torch.Tensor, coords: torch.Tensor, cam_tf: torch.Tensor, ray_num: int, point_num: int, focal,
                 near: float, far: float, output_samples=True, rng: str = "philox"):
    """Random training rays + stratified coarse samples (utils.py:72-94).
    -> (pts (N,C,3), lengths (N,C), rgb (N,3), rays (N,6)) or (rgb, rays).
    ``rng`` (an addition): "philox" (default) = ONE kernel launch, pixel indices and depth jitter drawn inside it from a seed taken
    off torch's CPU generator (reproducible under torch.manual_seed; nothing crosses PCIe); "reference" = the reference's own stream
    (torch.randint then torch.rand on the CPU default generator, copied to the device) for seeded bit-comparisons with it."""
    dev = rgbs.device
    fx, fy = _focal_xy(focal)
    if rng == "philox":
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        pts, z, rgb, rays = ops.sample_training_rays(rgbs, coords, cam_tf, fx, fy, near, far, ray_num, point_num if output_samples else 0, seed,
                                                     want_samples=bool(output_samples))
        return (pts, z, rgb, rays) if output_samples else (rgb, rays)
    idx = torch.randint(0, coords.shape[0], (ray_num,)).to(dev)                  # CPU generator, like the reference
    rgb = rgbs[idx]
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


def trans_t(t):
    """translation along the camera's z axis (utils.py:136-140)"""
    return torch.Tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, t], [0, 0, 0, 1]]).float()


def rot_phi(phi):
    """rotation about x by `phi` radians (utils.py:142-146)"""
    c, s_ = np.cos(phi), np.sin(phi)
    return torch.Tensor([[1, 0, 0, 0], [0, c, -s_, 0], [0, s_, c, 0], [0, 0, 0, 1]]).float()


def rot_theta(th):
    """rotation about y by `th` radians (utils.py:148-152)"""
    c, s_ = np.cos(th), np.sin(th)
    return torch.Tensor([[c, 0, -s_, 0], [0, 1, 0, 0], [s_, 0, c, 0], [0, 0, 0, 1]]).float()


def pose_spherical(theta, phi, radius):
    """Orbit camera-to-world matrix (utils.py:154-159): pitch, then yaw, then the COLMAP axis flip."""
    flip = torch.Tensor(np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]))
    return flip @ (rot_theta(theta / 180. * np.pi) @ (rot_phi(phi / 180. * np.pi) @ trans_t(radius)))
