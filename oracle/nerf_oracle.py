"""CPU restatement (torch, fp32) of the reference ray-march hot path.  TEST INFRASTRUCTURE ONLY.

This is the *checker* for the HIP path in ``nerf_amd``: it is imported by ``tests/``, by
``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline`` leg, never by the product.

Why torch and not numpy/C: the reference (Enigmatisms/NeRF) is itself pure torch, so restating it
on the same aten CPU kernels (``cumsum``/``cumprod`` accumulate in fp64 and round per element,
``searchsorted(right=True)``, ``sort``) is the closest thing to the reference that can travel to the
GPU box.  Every function below is pinned against the real reference, imported in the build
container, by the golden vectors in ``tests/golden/`` (generator: ``tests/golden/make_golden.py``).

Differences from the reference are deliberate and limited to the call convention:
  * every random draw is an explicit ``u`` argument (the reference draws from the CPU default
    generator inside the function: ``utils.py:89,115``, ``procedures.py:65``);
  * nothing calls ``.cuda()``; tensors stay on the device of their inputs;
  * networks are plain dicts of tensors keyed exactly like the reference ``state_dict``s.

All citations are ``file:line`` under ``/root/reference``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
RENDER_COARSE_PNUM = 64                     # procedures.py:22
POSSIBLE_PATCH_SIZE = (50, 40, 60, 30)      # procedures.py:21


# --------------------------------------------------------------------------------------------
# row 1: ray generation
# --------------------------------------------------------------------------------------------
def _focal_pair(focal) -> Tuple[float, float]:
    """(f_for_x, f_for_y).  Tuple focal: x is divided by focal[1], y by focal[0]
    (procedures.py:45-47, utils.py:79-81); scalar focal divides both (procedures.py:49)."""
    if isinstance(focal, (tuple, list)) or (hasattr(focal, "__len__") and len(focal) == 2):
        return float(focal[1]), float(focal[0])
    return float(focal), float(focal)


def pixel_camera_coords(H: int, W: int, focal, device=None) -> Tensor:
    """(H, W, 3) camera-space ray of every pixel: ((col - W/2 + .5)/fx, (H/2 - row + .5)/fy, -1).
    procedures.py:43-50."""
    fx, fy = _focal_pair(focal)
    col = torch.arange(W, dtype=torch.int64, device=device).view(1, W).expand(H, W)
    row = torch.arange(H, dtype=torch.int64, device=device).view(H, 1).expand(H, W)
    # the reference forms (col - W/2) in float (python float W/2) and then adds 0.5
    cx = (col - W / 2) + 0.5
    cy = (H / 2 - row) + 0.5
    cx = (cx / fx).to(torch.float32)
    cy = (cy / fy).to(torch.float32)
    return torch.stack((cx, cy, -torch.ones_like(cx)), dim=-1)


def ray_dirs_image(pose: Tensor, H: int, W: int, focal) -> Tensor:
    """Unnormalised world ray directions (H, W, 3) = R . c.  procedures.py:51."""
    cam = pixel_camera_coords(H, W, focal, pose.device)
    return torch.sum(cam.unsqueeze(-2) * pose[..., :-1], dim=-1)


def ray_dirs_pixels(coords: Tensor, pose: Tensor, focal) -> Tensor:
    """Training twin: integer (col - W//2, H//2 - row) coordinates from ``randomFromOneImage``
    -> world directions.  utils.py:78-85."""
    fx, fy = _focal_pair(focal)
    c = coords.to(torch.float32) + 0.5
    c = torch.stack((c[..., 0] / fx, c[..., 1] / fy), dim=-1)
    cam = torch.cat((c, -torch.ones_like(c[..., :1])), dim=-1)
    return torch.sum(cam.unsqueeze(-2) * pose[:, :-1], dim=-1)


def pixel_table(img: Tensor, crop_xy=(1.0, 1.0)) -> Tuple[Tensor, Tensor]:
    """``randomFromOneImage`` (utils.py:47-69): (H*W,3) pixels and (H*W,2) integer coords
    (col - W//2, H//2 - row), optionally centre-cropped."""
    if img.dim() > 3:
        img = img.squeeze(0)
    Himg, Wimg = img.shape[1], img.shape[2]
    hw, hh = Wimg // 2, Himg // 2
    x_lb, x_ub = (int(hw * (1.0 - crop_xy[0])), int(hw + hw * crop_xy[0])) if crop_xy[0] < 0.99 else (0, Wimg)
    y_lb, y_ub = (int(hh * (1.0 - crop_xy[1])), int(hh + hh * crop_xy[1])) if crop_xy[1] < 0.99 else (0, Himg)
    rows = torch.arange(y_lb, y_ub).view(-1, 1).expand(y_ub - y_lb, x_ub - x_lb)
    cols = torch.arange(x_lb, x_ub).view(1, -1).expand(y_ub - y_lb, x_ub - x_lb)
    coords = torch.stack((cols - hw, hh - rows), dim=-1).to(img.device).reshape(-1, 2)
    pix = img[:, rows, cols].reshape(3, -1).transpose(0, 1).contiguous()
    return pix, coords


def fov2focal(fov, img_size):
    """utils.py:96-105 -- note the square-image branch has no 1/2 (a caller-side quirk)."""
    if isinstance(fov, (tuple, list)):
        return (0.5 * img_size[0] / math.tan(0.5 * fov[1]), 0.5 * img_size[1] / math.tan(0.5 * fov[0]))
    if img_size[0] == img_size[1]:
        img_size = img_size[0]
    f = img_size / math.tan(0.5 * fov)
    return (f, f)


def pose_spherical(theta: float, phi: float, radius: float) -> Tensor:
    """utils.py:136-159: orbit camera-to-world (4,4)."""
    ph, th = phi / 180.0 * math.pi, theta / 180.0 * math.pi
    t = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], dtype=torch.float32)
    rp = torch.tensor([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0],
                       [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1]], dtype=torch.float32)
    rt = torch.tensor([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0],
                       [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1]], dtype=torch.float32)
    flip = torch.tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=torch.float32)
    return flip @ (rt @ (rp @ t))


# --------------------------------------------------------------------------------------------
# row 2: stratified sampling
# --------------------------------------------------------------------------------------------
def stratified_train(near: float, far: float, C: int, u: Tensor) -> Tensor:
    """z = linspace(near, far - d, C) + u*d, d = (far-near)/C.  utils.py:87-89."""
    res = (far - near) / C
    base = torch.linspace(near, far - res, C, device=u.device)
    return base + u * res


def stratified_render(near: float, far: float, sample_num: int, u: Tensor) -> Tensor:
    """render_image: 64 coarse planes over [near, far] *inclusive*, jitter scaled by the FINE
    count.  procedures.py:52,59,65.  u: (N, 64)."""
    res = (far - near) / sample_num
    base = torch.linspace(near, far, RENDER_COARSE_PNUM, device=u.device)
    return base + u * res


# --------------------------------------------------------------------------------------------
# row 3: positional encoding
# --------------------------------------------------------------------------------------------
def positional_encoding(x: Tensor, L: int) -> Tensor:
    """[sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] on the last dim (each block keeps the xyz order).
    nerf_helper.py:38-48."""
    parts = []
    for f in range(L):
        a = (2.0 ** f) * x
        parts.append(torch.sin(a))
        parts.append(torch.cos(a))
    return torch.cat(parts, dim=-1)


# --------------------------------------------------------------------------------------------
# rows 4 / 9: the two MLPs, functional, on state_dict-shaped dicts
# --------------------------------------------------------------------------------------------
def _bf16(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def _linear(x: Tensor, w: Tensor, b: Tensor, emulate_bf16: bool) -> Tensor:
    """nn.Linear.  ``emulate_bf16`` models the MFMA bf16 path of the HIP kernels: operands rounded
    to bf16 (RNE), products and sums in fp32, bias added in fp32."""
    if emulate_bf16:
        return F.linear(_bf16(x), _bf16(w)) + b
    return F.linear(x, w, b)


def proposal_forward(sd: Dict[str, Tensor], pts: Tensor, L: int = 10, emulate_bf16: bool = False, cat_origin: bool = True) -> Tensor:
    """ProposalNetwork.forward: [x, PE_L(x)] -> 4x(Linear+ReLU) -> Linear(.,1).  addtional.py:67-71,88-96.
    pts (N, C, 3) -> density (N, C) (no activation).  `cat_origin` = the constructor flag (addtional.py:61,93-94: the raw position in front)."""
    h = torch.cat((pts, positional_encoding(pts, L)), dim=-1) if cat_origin else positional_encoding(pts, L)
    for i in (0, 2, 4, 6):
        h = F.relu(_linear(h, sd[f"layers.{i}.weight"], sd[f"layers.{i}.bias"], emulate_bf16))
    return _linear(h, sd["layers.8.weight"], sd["layers.8.bias"], emulate_bf16).squeeze(-1)


def mip_forward(sd: Dict[str, Tensor], pts: Tensor, Lp: int = 10, Ld: int = 4, emulate_bf16: bool = False,
                encoded_x: Optional[Tensor] = None, cat_origin: bool = True) -> Tensor:
    """MipNeRF.forward (mip_model.py:41-60).  pts (N, S, 6) = [position | raw direction] -> (N, S, 4)
    = [sigmoid rgb | raw sigma].  Skip-cat order (enc, h) (:55); head-cat order (bottleneck, dir) (:59).
    ``encoded_x`` (not in the reference's forward): the 6 Lp encoding columns that follow the position, e.g. ipe_feature's output,
    used instead of positional_encoding(x).  `cat_origin` = the constructor flag (mip_model.py:50-52: raw position / direction in front)."""
    x = pts[..., :3]
    d = pts[..., 3:6]
    d = d / d.norm(dim=-1, keepdim=True)
    ex = positional_encoding(x, Lp) if encoded_x is None else encoded_x
    ed = positional_encoding(d, Ld)
    if cat_origin:
        ex, ed = torch.cat((x, ex), dim=-1), torch.cat((d, ed), dim=-1)
    h = ex
    for i in (0, 2, 4, 6):
        h = F.relu(_linear(h, sd[f"lin_block1.{i}.weight"], sd[f"lin_block1.{i}.bias"], emulate_bf16))
    g = torch.cat((ex, h), dim=-1)
    for i in (0, 2, 4):
        g = F.relu(_linear(g, sd[f"lin_block2.{i}.weight"], sd[f"lin_block2.{i}.bias"], emulate_bf16))
    sigma = _linear(g, sd["opacity_head.0.weight"], sd["opacity_head.0.bias"], emulate_bf16)
    b = _linear(g, sd["bottle_neck.0.weight"], sd["bottle_neck.0.bias"], emulate_bf16)
    c = F.relu(_linear(torch.cat((b, ed), dim=-1), sd["rgb_layer.0.weight"], sd["rgb_layer.0.bias"], emulate_bf16))
    rgb = torch.sigmoid(_linear(c, sd["rgb_layer.2.weight"], sd["rgb_layer.2.bias"], emulate_bf16))
    return torch.cat((rgb, sigma), dim=-1)


def init_linear_params(shapes: Sequence[Tuple[str, int, int]], seed: int, std: float = 0.02,
                       bias_std: float = 0.0) -> Dict[str, Tensor]:
    """Deterministic test weights: trunc-normal(std) like nerf_base.py:15-19 (bias_std>0 gives
    non-zero biases so tests exercise the bias path)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, out_f, in_f in shapes:
        w = torch.empty(out_f, in_f)
        torch.nn.init.trunc_normal_(w, std=std, a=-2 * std, b=2 * std, generator=g)
        sd[name + ".weight"] = w
        sd[name + ".bias"] = torch.randn(out_f, generator=g) * bias_std
    return sd


def proposal_shapes(L: int = 10, hidden: int = 256, cat_origin: bool = True):
    i = 6 * L + (3 if cat_origin else 0)
    return [("layers.0", hidden, i), ("layers.2", hidden, hidden), ("layers.4", hidden, hidden),
            ("layers.6", hidden, hidden), ("layers.8", 1, hidden)]


def mip_shapes(Lp: int = 10, Ld: int = 4, hidden: int = 256, cat_origin: bool = True):
    i = 6 * Lp + (3 if cat_origin else 0)
    return [("lin_block1.0", hidden, i), ("lin_block1.2", hidden, hidden), ("lin_block1.4", hidden, hidden),
            ("lin_block1.6", hidden, hidden), ("lin_block2.0", hidden, hidden + i), ("lin_block2.2", hidden, hidden),
            ("lin_block2.4", 256, hidden), ("bottle_neck.0", 256, 256), ("opacity_head.0", 1, 256),
            ("rgb_layer.0", 128, 256 + 6 * Ld + (3 if cat_origin else 0)), ("rgb_layer.2", 3, 128)]


# --------------------------------------------------------------------------------------------
# row 5: sigma -> weights
# --------------------------------------------------------------------------------------------
def sigma_to_weights(sigma: Tensor, z: Tensor, ray_dirs: Optional[Tensor] = None, density_act=F.relu) -> Tensor:
    """addtional.py:100-107 (ray_dirs given -> z scaled by |d|, act = relu) and nerf_base.py:80-86
    (caller pre-scales z).  delta_last = 1e10; alpha = 1 - exp(-act(sigma) delta);
    T = exclusive cumprod of (exp(..) + 1e-10)."""
    if ray_dirs is not None:
        z = z * ray_dirs.norm(dim=-1, keepdim=True)
    big = torch.full((z.shape[0], 1), 1e10, dtype=z.dtype, device=z.device)
    delta = torch.cat((z[:, 1:] - z[:, :-1], big), dim=-1)
    m = torch.exp(-density_act(sigma) * delta)
    alpha = 1.0 - m
    ones = torch.ones((z.shape[0], 1), dtype=z.dtype, device=z.device)
    T = torch.cumprod(torch.cat((ones, m + 1e-10), dim=-1), dim=-1)[:, :-1]
    return alpha * T


# --------------------------------------------------------------------------------------------
# row 6: max-blur filter
# --------------------------------------------------------------------------------------------
def max_blur(w: Tensor, alpha: float) -> Tensor:
    """mip_methods.py:61-66: mx_i = max(w_i, w_{i+1}); out = .5*([w_0, mx] + [mx, w_last]) + alpha."""
    mx = torch.maximum(w[..., :-1], w[..., 1:])
    front = torch.cat((w[..., :1], mx), dim=-1)
    rear = torch.cat((mx, w[..., -1:]), dim=-1)
    return 0.5 * (front + rear) + alpha


# --------------------------------------------------------------------------------------------
# in-kernel uniforms: Philox4x32-10 (Salmon et al. 2011), the counter layout of nerf_amd/csrc/device_common.h
# --------------------------------------------------------------------------------------------
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """numpy uint32 arrays (broadcastable) -> four uint32 arrays.  Ten rounds; multipliers 0xD2511F53 / 0xCD9E8D57, key increments
    0x9E3779B9 / 0xBB67AE85 (the paper's constants)."""
    import numpy as np
    u32, u64 = np.uint32, np.uint64
    c0, c1, c2, c3 = (np.asarray(c, dtype=u32) for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = u32(k0), u32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = u64(0xD2511F53) * c0.astype(u64)
            p1 = u64(0xCD9E8D57) * c2.astype(u64)
            hi0, lo0 = (p0 >> u64(32)).astype(u32), p0.astype(u32)
            hi1, lo1 = (p1 >> u64(32)).astype(u32), p1.astype(u32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = u32((int(k0) + 0x9E3779B9) & 0xFFFFFFFF), u32((int(k1) + 0xBB67AE85) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox_uniforms(seed: int, n_rays: int, ray_offset: int = 0, S: int = 64, K: int = 129):
    """The uniforms the HIP kernels draw when none are passed (include/nerf_amd.h, nerf_amd_samples.rng_seed).  One Philox block per
    (ray, slot j): block(ray, j) = Philox(key = seed, counter = (ray_lo, ray_hi, j, 'RS' = 0x5253)), ray = n + ray_offset (64 bit);
    u_strat (N, S) = word 0 of block j = s;  u_inv (N, K): k = 192 b + r, r < 192 -> word 1 + r // 64 of block j = 64 b + r % 64;
    value = (word >> 8) * 2^-24."""
    import numpy as np
    n = np.arange(n_rays, dtype=np.uint64) + np.uint64(ray_offset)
    nlo, nhi = (n & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None], (n >> np.uint64(32)).astype(np.uint32)[:, None]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    s = np.arange(S, dtype=np.uint32)[None, :]
    us = philox4x32_10(nlo, nhi, s, np.uint32(0x5253), k0, k1)[0] + np.zeros((n_rays, 1), np.uint32)
    k = np.arange(K, dtype=np.int64)[None, :]
    blk, r = k // 192, k % 192
    w = philox4x32_10(nlo, nhi, (64 * blk + r % 64).astype(np.uint32), np.uint32(0x5253), k0, k1)
    ui = np.choose(1 + r // 64 + np.zeros((n_rays, 1), np.int64), w)
    to_f = lambda x: torch.from_numpy(((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)))
    return to_f(us), to_f(ui)


def philox_normal(seed: int, n_samples: int, std: float, sample_offset: int = 0):
    """The bottle-neck perturbation the Ref-NeRF training forward draws in place (nerf_amd/csrc/device_common.h philox_normal8; the
    reference draws torch.normal(0, w, shape) on the device generator, ref_model.py:84-85 -- any N(0, w) stream serves).  (n_samples, 128)
    fp32: one Philox block per (sample m, q) of stream 'BN' = 0x424E, q = 2 (f >> 4) + ((f >> 2) & 1) for feature f; word w of the block
    -> 16-bit uniforms u0 = (lo + .5) / 65536, u1 = (hi + .5) / 65536 -> r = sqrt(-2 ln u0), (r cos 2 pi u1, r sin 2 pi u1) = elements
    2 w, 2 w + 1 of the block's eight deviates; feature f takes element (f & 3) + 4 ((f >> 3) & 1)."""
    import numpy as np
    m = np.arange(n_samples, dtype=np.uint64) + np.uint64(sample_offset)
    mlo, mhi = (m & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None], (m >> np.uint64(32)).astype(np.uint32)[:, None]
    q = np.arange(16, dtype=np.uint32)[None, :]
    w = philox4x32_10(mlo, mhi, q + np.zeros_like(mlo), np.uint32(0x424E), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)     # 4 x (n, 16)
    z = np.empty((n_samples, 16, 8), np.float32)
    for k in range(4):
        u0 = ((w[k] & np.uint32(0xFFFF)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -16)
        u1 = ((w[k] >> np.uint32(16)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -16)
        r = np.float32(std) * np.sqrt(np.float32(-2.0) * np.log(u0))
        z[:, :, 2 * k] = r * np.cos(2.0 * np.pi * u1.astype(np.float64)).astype(np.float32)
        z[:, :, 2 * k + 1] = r * np.sin(2.0 * np.pi * u1.astype(np.float64)).astype(np.float32)
    f = np.arange(128)
    return torch.from_numpy(z[:, 2 * (f >> 4) + ((f >> 2) & 1), (f & 3) + 4 * ((f >> 3) & 1)].astype(np.float32))


# --------------------------------------------------------------------------------------------
# row 7: inverse-transform sampling
# --------------------------------------------------------------------------------------------
def cascade_row_sum(x) -> "numpy.ndarray":
    """The ORDER in which torch's CPU kernel sums a contiguous fp32 row (ATen SumKernel.cpp: cascade_sum -> vectorized_inner_sum ->
    row_sum / multi_row_sum), restated in numpy for rows of < 512 elements: the row is read as 8-float vectors; vector v is added to
    accumulator v % 4 while whole groups of four vectors remain, to accumulator 0 afterwards; accumulators 1..3 are then added to 0;
    the scalar is 0 + the (< 8) tail elements in order + the 8 lanes of accumulator 0 in order.  Rows shorter than one vector take
    the scalar form of the same scheme (scalar_inner_sum): element j goes to accumulator j % 4 while whole groups of four remain, to
    accumulator 0 afterwards, then accumulators 1..3 are added to 0.  The HIP inverse-sampling kernels
    use this order for the pdf normaliser of utils.py:110-111 so that their CDF is bit-identical to the reference's; this function
    is what tests/test_oracle_golden.py checks against torch.sum itself (on the machine that runs the tests).  x: (N, n) float32."""
    import numpy as np
    x = np.asarray(x, dtype=np.float32)
    N, n = x.shape
    f32 = np.float32
    if n < 8:
        acc = [np.zeros(N, f32) for _ in range(4)]
        for i in range(n // 4):
            for k in range(4):
                acc[k] = (acc[k] + x[:, 4 * i + k]).astype(f32)
        for j in range(4 * (n // 4), n):
            acc[0] = (acc[0] + x[:, j]).astype(f32)
        for k in (1, 2, 3):
            acc[0] = (acc[0] + acc[k]).astype(f32)
        return acc[0]
    nv, n4 = n // 8, (n // 8) // 4
    vec = x[:, : nv * 8].reshape(N, nv, 8)
    acc = [np.zeros((N, 8), f32) for _ in range(4)]
    for i in range(n4):
        for k in range(4):
            acc[k] = (acc[k] + vec[:, 4 * i + k]).astype(f32)
    for v in range(4 * n4, nv):
        acc[0] = (acc[0] + vec[:, v]).astype(f32)
    for k in (1, 2, 3):
        acc[0] = (acc[0] + acc[k]).astype(f32)
    s = np.zeros(N, f32)
    for j in range(nv * 8, n):
        s = (s + x[:, j]).astype(f32)
    for c in range(8):
        s = (s + acc[0][:, c]).astype(f32)
    return s


def pdf_cdf(weights: Tensor) -> Tensor:
    """The CDF sample_pdf searches (utils.py:110-113): weights (N, B-1) -> (N, B) with the leading 0."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    return torch.cat((torch.zeros_like(cdf[..., :1]), cdf), -1)


def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """utils.py:108-133 with u explicit.  bins (N, B), weights (N, B-1), u (N, K) -> samples,
    below, above."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat((torch.zeros_like(cdf[..., :1]), cdf), -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_lo, bin_hi = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom
    return bin_lo + t * (bin_hi - bin_lo), below, above


def inverse_sample(weights: Tensor, coarse_z: Tensor, u: Tensor, sort: bool = True):
    """utils.py:34-44: bins = mid-points of z; pdf over weights[1:-1]; optional sort (+ gather of
    ``below`` by the sort permutation).  u: (N, sample_pnum)."""
    weights = weights.detach()
    mids = 0.5 * (coarse_z[..., 1:] + coarse_z[..., :-1])
    z, below, _ = sample_pdf(mids, weights[..., 1:-1], u)
    if sort:
        z, order = torch.sort(z, dim=-1)
        return z, torch.gather(below, -1, order)
    return z


# --------------------------------------------------------------------------------------------
# row 8: sample assembly
# --------------------------------------------------------------------------------------------
def length2pts(rays: Tensor, z: Tensor) -> Tensor:
    """nerf_base.py:53-56: (N, S, 6) = [o + z d | d]."""
    pts = rays[:, None, :3] + rays[:, None, 3:] * z[:, :, None]
    return torch.cat((pts, rays[:, None, 3:].expand(-1, z.shape[1], -1)), dim=-1)


def coarse_fine_merge(rays: Tensor, c_z: Tensor, f_z: Tensor, f_inds: Optional[Tensor] = None):
    """nerf_base.py:59-73: cat(fine, coarse) -> sort -> drop last -> points (2- or 4-tuple)."""
    z, order = torch.sort(torch.cat((f_z, c_z), dim=-1), dim=-1)
    if f_inds is not None:
        c_inds = torch.arange(c_z.shape[-1], device=z.device).unsqueeze(0).expand(c_z.shape[0], -1)
        inds = torch.gather(torch.cat((f_inds, c_inds), dim=-1), -1, order)
    z = z[..., :-1]
    samples = length2pts(rays, z)
    if f_inds is not None:
        return samples, z, inds, order[..., :-1]
    return samples, z


# --------------------------------------------------------------------------------------------
# row 10: alpha compositing
# --------------------------------------------------------------------------------------------
def composite(rgbo: Tensor, z: Tensor, ray_dirs: Tensor, mul_norm: bool = True, white_bkg: bool = False,
              density_act=F.relu, render_depth: Optional[Tuple[float, float]] = None, normal_info=None):
    """NeRF.render (nerf_base.py:91-113) -> (rgb (N,3), weights (N,S), extras)."""
    if mul_norm:
        z = z * ray_dirs.norm(dim=-1, keepdim=True)
    w = sigma_to_weights(rgbo[..., -1], z, None, density_act)
    rgb = torch.sum(w[:, :, None] * rgbo[..., :3], dim=-2)
    if white_bkg:
        rgb = rgb + (1.0 - torch.sum(w, -1)[..., None])
    extras = {}
    if render_depth is not None:
        near, far = render_depth
        extras["depth_img"] = (torch.sum(w * z, dim=-1) - near) / (far - near)
    if normal_info is not None:
        normal, cam_dir = normal_info
        extras["normal_img"] = (torch.sum(w * (normal @ cam_dir), dim=-1) + 1.0) * 0.5
    return rgb, w, extras


# --------------------------------------------------------------------------------------------
# row 11: distillation bound + losses
# --------------------------------------------------------------------------------------------
def get_bounds(prop_w: Tensor, below: Tensor) -> Tensor:
    """addtional.py:14-18 (index-offset quirk reproduced: ``below`` indexes the C-long prefix sum)."""
    starts, ends = below[:, :-1], below[:, 1:] + 1
    sat = torch.cat((torch.zeros(prop_w.shape[0], 1, device=prop_w.device), torch.cumsum(prop_w, dim=-1)), dim=-1)
    return torch.gather(sat, -1, ends) - torch.gather(sat, -1, starts)


def proposal_loss(bounds: Tensor, fine_w: Tensor) -> Tensor:
    """addtional.py:20-24."""
    return torch.sum(F.relu(fine_w - bounds) ** 2 / (fine_w + 1e-8))


def loss_psnr(mse: Tensor) -> Tensor:
    """addtional.py:45-51."""
    return -10.0 * torch.log(mse) / 2.3025851249694824


# --------------------------------------------------------------------------------------------
# row 12: integrated PE (dead code in the reference; pinned by golden only)
# --------------------------------------------------------------------------------------------
def cone_parameters(z: Tensor, r: float):
    """coneParameters (mip_methods.py:15-23): z (N, S+1) -> mu_t, sigma_t^2, sigma_r^2 (N, S)."""
    mid = (z[:, 1:] + z[:, :-1]) / 2
    hw2 = ((z[:, 1:] - z[:, :-1]) / 2) ** 2
    t1 = 3 * mid ** 2 + hw2
    mu_t = mid + 2 * mid * hw2 / t1
    var_t = hw2 / 3 - 4 * (hw2 ** 2) * (12 * mid ** 2 - hw2) / 15 / (t1 ** 2)
    var_r = (r ** 2) * (0.25 * mid ** 2 + 5 / 12 * hw2 - 4 * hw2 ** 2 / (15 * t1))
    return mu_t, var_t, var_r


def ipe_feature(z: Tensor, rays: Tensor, L: int, r: float, dir_norm: Optional[Tensor] = None, contracted: bool = False):
    """mip_methods.py:15-58.  z (N, S+1) -> (N, S, 6L) feature, mu (N,S,3), mu_t (N,S).
    Quirk kept: ``.norm()`` at :31 is over the whole (N,3) direction tensor; ``dir_norm`` (a test hook) substitutes the norm of a
    larger batch these rays were cut from.  ``contracted`` (not in the reference; the build's own definition of BASELINE configs[2] +
    configs[4] together): the frustum MEAN goes through contract() before the lift, the diagonal covariance stays in metric space."""
    mid = (z[:, 1:] + z[:, :-1]) / 2
    hw2 = ((z[:, 1:] - z[:, :-1]) / 2) ** 2
    t1 = 3 * mid ** 2 + hw2
    mu_t = mid + 2 * mid * hw2 / t1
    var_t = hw2 / 3 - 4 * (hw2 ** 2) * (12 * mid ** 2 - hw2) / 15 / (t1 ** 2)
    var_r = (r ** 2) * (0.25 * mid ** 2 + 5 / 12 * hw2 - 4 * hw2 ** 2 / (15 * t1))
    o, d = rays[:, :3], rays[:, 3:]
    mu = o[:, None, :] + mu_t[:, :, None] * d[:, None, :]
    if contracted:
        mu = contract(mu)
    dd = d * d
    perp = torch.ones(3, device=z.device, dtype=z.dtype)[None, :] - dd / (d.norm() if dir_norm is None else dir_norm)
    diag = var_t[:, :, None] * dd[:, None, :] + var_r[:, :, None] * perp[:, None, :]
    N, S, _ = mu.shape
    f2 = torch.tensor([2.0 ** i for i in range(L)], device=z.device, dtype=z.dtype)
    f4 = torch.tensor([4.0 ** i for i in range(L)], device=z.device, dtype=z.dtype)
    mu_r = (f2[None, None, :, None] * mu[:, :, None, :])                 # (N,S,L,3)
    var = (f4[None, None, :, None] * diag[:, :, None, :])
    att = torch.exp(-0.5 * var)
    feat = torch.cat((torch.sin(mu_r) * att, torch.cos(mu_r) * att), dim=-1).reshape(N, S, -1)
    return feat, mu, mu_t


# --------------------------------------------------------------------------------------------
# the per-batch pipeline = body of the render_image tile loop (procedures.py:64-85), non-ref
# --------------------------------------------------------------------------------------------
def contract(x: Tensor) -> Tensor:
    """Mip-NeRF 360 scene contraction (Barron et al. 2022, eq. 10): x if |x| <= 1 else (2 - 1/|x|) x/|x|.  NOT in the reference
    (BASELINE config 5 names it; parity unpinned): this is the build's own definition, which the HIP kernels are tested against."""
    n = x.norm(dim=-1, keepdim=True)
    k = torch.where(n > 1.0, (2.0 - 1.0 / n.clamp_min(1e-30)) / n.clamp_min(1e-30), torch.ones_like(n))
    return x * k


def render_rays(prop_sd, mip_sd, rays: Tensor, u_strat: Tensor, u_inv: Tensor, near: float, far: float,
                sample_num: int = 128, white_bkg: bool = False, emulate_bf16: bool = False,
                stages: Optional[dict] = None, contracted: bool = False, ipe_radius: Optional[float] = None,
                ipe_dir_norm: Optional[Tensor] = None):
    """rays (N,6), u_strat (N,64), u_inv (N,sample_num+1) -> rgb (N,3), weights (N,S), depth (N,).
    ``stages`` (optional dict) receives every intermediate for stage-by-stage parity tests.  ``contracted`` (not in the
    reference): every sample position goes through contract() before the networks see it; depths stay metric.
    ``ipe_radius`` (BASELINE config 3; the reference holds ipe_feature but no caller, so this wiring is the build's own definition
    -- parity of the LOOP unpinned, the function itself pinned by G12/G18): the fine network's input becomes [mu | ipe_feature] of
    the conical frusta between consecutive fine depths z_f[s], z_f[s+1] (all sample_num+1 sorted depths are used: sample_num frusta);
    the proposal pass and the compositing depths are unchanged."""
    z_c = stratified_render(near, far, sample_num, u_strat)
    pts_c = rays[:, None, :3] + z_c[..., None] * rays[:, None, 3:]
    if contracted:
        pts_c = contract(pts_c)
    density = proposal_forward(prop_sd, pts_c, emulate_bf16=emulate_bf16)       # no softplus here (:67-68)
    w_raw = sigma_to_weights(density, z_c, rays[:, 3:])
    w_prop = max_blur(w_raw, 0.01)
    z_f, below = inverse_sample(w_prop, z_c, u_inv, sort=True)
    z_all = z_f
    z_f = z_f[..., :-1]
    pts_f = length2pts(rays, z_f)
    enc = None
    if ipe_radius is not None:
        enc, mu, _ = ipe_feature(z_all, rays, 10, ipe_radius, ipe_dir_norm, contracted=contracted)   # (mu already contracted)
        pts_f = torch.cat((mu, pts_f[..., 3:]), dim=-1)
    elif contracted:
        pts_f = torch.cat((contract(pts_f[..., :3]), pts_f[..., 3:]), dim=-1)
    rgbo = mip_forward(mip_sd, pts_f, emulate_bf16=emulate_bf16, encoded_x=enc)
    rgb, w, extras = composite(rgbo, z_f, rays[:, 3:], white_bkg=white_bkg, render_depth=(near, far))
    if stages is not None:
        stages.update(z_coarse=z_c, density=density, w_raw=w_raw, w_prop=w_prop, z_fine=z_f, below=below,
                      rgbo=rgbo)
    return rgb, w, extras["depth_img"]


def patch_size(image_size):
    """procedures.py:24-31 (raises like the reference when no patch size divides W)."""
    for p in POSSIBLE_PATCH_SIZE:
        if image_size[1] % p == 0:
            return p, (image_size[0] // p, image_size[1] // p)
    raise UnboundLocalError("no patch size in (50,40,60,30) divides the image width")


def render_image(prop_sd, mip_sd, pose: Tensor, image_size, focal, near: float, far: float, sample_num: int = 128,
                 white_bkg: bool = False, render_depth: bool = False, generator: Optional[torch.Generator] = None,
                 max_tiles: Optional[int] = None):
    """Whole-image render with the reference's tile order AND RNG draw order (per tile: one
    (sz,sz,64) draw, then one (sz*sz, sample_num+1) draw; procedures.py:62-70, utils.py:115).
    With ``generator=None`` the CPU default generator is used, like the reference."""
    if not isinstance(image_size, (tuple, list)):
        image_size = (image_size, image_size)
    H, W = image_size
    dirs = ray_dirs_image(pose, H, W, focal)
    out = {"rgb": torch.zeros(3, H, W)}
    if render_depth:
        out["depth_img"] = torch.zeros(3, H, W)
    sz, (pr, pc) = patch_size(image_size)
    done = 0
    for k in range(pr):
        for j in range(pc):
            d = dirs[sz * k: sz * (k + 1), sz * j: sz * (j + 1)].reshape(-1, 3)
            rays = torch.cat((pose[:, -1].expand(sz * sz, -1), d), dim=-1)
            u1 = torch.rand((sz, sz, RENDER_COARSE_PNUM), generator=generator).view(-1, RENDER_COARSE_PNUM)
            u2 = torch.rand((sz * sz, sample_num + 1), generator=generator)
            rgb, _, depth = render_rays(prop_sd, mip_sd, rays, u1, u2, near, far, sample_num, white_bkg)
            out["rgb"][:, sz * k: sz * (k + 1), sz * j: sz * (j + 1)] = rgb.view(sz, sz, 3).permute(2, 0, 1)
            if render_depth:
                out["depth_img"][:, sz * k: sz * (k + 1), sz * j: sz * (j + 1)] = depth.view(sz, sz)
            done += 1
            if max_tiles is not None and done >= max_tiles:
                return out
    return out


# --------------------------------------------------------------------------------------------
# row 13: Ref-NeRF (ref_model.py:16-106) and the integrated directional encoding (ref_func.py:10-110)
# --------------------------------------------------------------------------------------------
def ide_tables(deg_view: int):
    """(ml_array (2,T) int64, mat (l_max+1, T) float32): spherical-harmonic coefficient table of ref_func.py:60-74,
    computed in float64 like the reference (np.math.factorial / np.prod) and stored as float32."""
    import numpy as np
    if deg_view > 5:
        raise ValueError("Only deg_view of at most 5 is numerically stable.")

    def gen_binom(a, k):
        return np.prod(a - np.arange(k)) / math.factorial(k)

    def legendre(l, m, k):
        return ((-1) ** m * 2 ** l * math.factorial(l) / math.factorial(k) / math.factorial(l - k - m) *
                gen_binom(0.5 * (l + k + m - 1.0), l))

    def sph(l, m, k):
        return np.sqrt((2.0 * l + 1.0) * math.factorial(l - m) / (4.0 * np.pi * math.factorial(l + m))) * legendre(l, m, k)

    ml = [(m, 2 ** i) for i in range(deg_view) for m in range(2 ** i + 1)]
    ml_array = np.array(ml).T
    l_max = 2 ** (deg_view - 1)
    mat = torch.zeros(l_max + 1, ml_array.shape[1])
    for i, (m, l) in enumerate(ml_array.T):
        for k in range(l - m + 1):
            mat[k, i] = sph(l, m, k)
    return torch.from_numpy(ml_array), mat


def ide_encode(xyz: Tensor, kappa_inv: Tensor, deg_view: int = 4) -> Tensor:
    """integrated_dir_enc_fn (ref_func.py:76-108): (..., 3), (..., 1) -> (..., 2T) = [real | imag]."""
    ml, mat = ide_tables(deg_view)
    ml, mat = ml.to(xyz.device), mat.to(xyz.device, xyz.dtype)          # (fp64 anchor runs: the reference's fp32 table, widened)
    x, y, z = xyz[..., 0:1], xyz[..., 1:2], xyz[..., 2:3]
    vmz = torch.cat([z ** i for i in range(mat.shape[0])], dim=-1)
    vmxy = torch.cat([(x + 1j * y) ** m for m in ml[0, :]], dim=-1)
    sph_harms = vmxy * (vmz @ mat)
    sigma = 0.5 * ml[1, :] * (ml[1, :] + 1)
    ide = sph_harms * torch.exp(-sigma * kappa_inv)
    return torch.cat([torch.real(ide), torch.imag(ide)], dim=-1)


def ref_shapes(Lp: int = 10, deg: int = 4, hidden: int = 256, bottle: int = 128, out_dim: int = 256, cat_origin: bool = True):
    i = 6 * Lp + (3 if cat_origin else 0)
    enc = ((1 << deg) - 1 + deg) << 1
    din = 1 + bottle + enc
    s = [("spa_block1.0", hidden, i), ("spa_block1.2", hidden, hidden), ("spa_block1.4", hidden, hidden), ("spa_block1.6", hidden, hidden),
         ("spa_block2.0", hidden, hidden + i), ("spa_block2.2", hidden, hidden), ("spa_block2.4", hidden, hidden), ("spa_block2.6", out_dim, hidden),
         ("rho_tau_head", 2, out_dim), ("norm_col_tint_head", 9, out_dim), ("bottle_neck", bottle, out_dim), ("spec_rgb_head.0", 3, out_dim),
         ("dir_block1.0", hidden, din), ("dir_block1.2", hidden, hidden), ("dir_block1.4", hidden, hidden), ("dir_block1.6", hidden, hidden),
         ("dir_block2.0", hidden, hidden + din), ("dir_block2.2", hidden, hidden), ("dir_block2.4", out_dim, hidden), ("dir_block2.6", out_dim, hidden)]
    return s


def linear_to_srgb(linear: Tensor) -> Tensor:
    """nerf_helper.py:50-56 (from multinerf): 323/25 x below the knee, (211 max(eps, x)^(5/12) - 11) / 200 above."""
    eps = torch.full((1,), torch.finfo(torch.float32).eps, dtype=linear.dtype, device=linear.device)
    srgb0 = 323 / 25 * linear
    srgb1 = (211 * torch.maximum(eps, linear) ** (5 / 12) - 11) / 200
    return torch.where(linear <= 0.0031308, srgb0, srgb1)


def ref_forward(sd: Dict[str, Tensor], pts: Tensor, ray_d: Optional[Tensor] = None, Lp: int = 10, deg: int = 4,
                emulate_bf16: bool = False, noise: Optional[Tensor] = None, use_srgb: bool = False, cat_origin: bool = True):
    """RefNeRF.forward (ref_model.py:68-106; `use_srgb`: lines 100-102 instead of 104-105).  pts (N,S,6) [or (N,S,3) + ray_d] ->
    ((N,S,4) = [rgb | raw density], normal (N,S,3)).  `noise` = the train-mode perturbation of the bottle-neck vector
    (ref_model.py:84-85: torch.normal(0, perturb_bottle_neck_w, shape)); None = eval mode."""
    lin = lambda name, t: _linear(t, sd[name + ".weight"], sd[name + ".bias"], emulate_bf16)
    x = pts[..., :3]
    ex = torch.cat((x, positional_encoding(x, Lp)), dim=-1) if cat_origin else positional_encoding(x, Lp)      # ref_model.py:70-74
    h = ex
    for i in (0, 2, 4, 6):
        h = F.relu(lin(f"spa_block1.{i}", h))
    g = torch.cat((ex, h), dim=-1)
    for i in (0, 2, 4, 6):
        g = F.relu(lin(f"spa_block2.{i}", g))
    nct = lin("norm_col_tint_head", g)
    normal, diffuse, tint = nct[..., 0:3], nct[..., 3:6], nct[..., 6:9]
    rt = lin("rho_tau_head", g)
    rough, density = rt[..., 0:1], rt[..., 1:2]
    rough = F.softplus(rough - 1.0)
    b = lin("bottle_neck", g)
    if noise is not None:
        b = b + noise
    normal = -normal / (normal.norm(dim=-1, keepdim=True) + 1e-7)
    d = pts[..., 3:] if ray_d is None else ray_d
    refl = d - 2.0 * torch.sum(d * normal, dim=-1, keepdim=True) * normal
    ide = ide_encode(refl, rough, deg)
    nv = torch.sum(normal * d, dim=-1, keepdim=True)
    allin = torch.cat((b, ide, nv), dim=-1)
    r = allin
    for i in (0, 2, 4, 6):
        r = F.relu(lin(f"dir_block1.{i}", r))
    r = torch.cat((allin, r), dim=-1)
    for i in (0, 2, 4, 6):
        r = F.relu(lin(f"dir_block2.{i}", r))
    spec = torch.sigmoid(lin("spec_rgb_head.0", r)) * torch.sigmoid(tint)
    if use_srgb:
        import math
        rgb = linear_to_srgb(spec + torch.sigmoid(diffuse - math.log(3.)))
    else:
        rgb = spec + torch.sigmoid(diffuse)
    return torch.cat((rgb, density), dim=-1), normal


def get_grad(func_val: Tensor, inputs: Tensor) -> Tensor:
    """RefNeRF.get_grad (ref_model.py:119-125): first-order d(func)/d(inputs), normalised, norm clamped at 1e-5."""
    grad, = torch.autograd.grad(func_val, inputs, torch.ones_like(func_val), retain_graph=True)
    n = grad.norm(dim=-1, keepdim=True)
    return grad / torch.maximum(torch.full_like(n, 1e-5), n)


def coarse_grad_select(fine_grads: Tensor, sort_inds: Tensor, c_pnum: int) -> Tensor:
    """ref_model.py:108-117: after cat(fine, coarse) + sort, pick the rows that came from the coarse samples."""
    n, total, _ = fine_grads.shape
    sel = torch.cat((torch.zeros(n, total - c_pnum, dtype=torch.bool), torch.ones(n, c_pnum, dtype=torch.bool)), dim=-1)
    return fine_grads[torch.gather(sel, -1, sort_inds)].reshape(n, c_pnum, -1)


def weighted_normal_loss(weight: Tensor, d_norm: Tensor, p_norm: Tensor) -> Tensor:
    return torch.sum(weight * (1.0 - torch.sum(d_norm * p_norm, dim=-1)))          # ref_model.py:127-135 (size_average False)


def back_face_loss(weight: Tensor, normal: Tensor, ray_d: Tensor) -> Tensor:
    return torch.mean(weight * F.relu(torch.sum(normal * ray_d, dim=-1)))           # ref_model.py:137-143


def ref_train_step(prop_sd, ref_sd, rays: Tensor, z_coarse: Tensor, u_inv: Tensor, noise: Tensor, rgb_tgt: Tensor, n_fine: int,
                   Lp: int = 10, deg: int = 4):
    """The Ref-NeRF branch of the training step with prop_normal on (train.py:164-199).  prop_sd / ref_sd hold leaf tensors
    (requires_grad) when parameter gradients are wanted.  Returns a dict of every intermediate the golden G17 pins."""
    C = z_coarse.shape[-1]
    pts = (rays[:, None, :3] + rays[:, None, 3:] * z_coarse[:, :, None]).detach().requires_grad_(True)
    dens = proposal_forward(prop_sd, pts, L=Lp)
    coarse_grad = -get_grad(dens, pts)
    pw = max_blur(sigma_to_weights(F.softplus(dens), z_coarse, rays[:, 3:]), 0.01)
    z_fine, below = inverse_sample(pw, z_coarse, u_inv, sort=True)
    samples, z_all, below_all, sort_ids = coarse_fine_merge(rays, z_coarse, z_fine, below)
    pos = samples[..., :3].detach().requires_grad_(True)
    d = samples[..., 3:]
    rgbo, normal = ref_forward(ref_sd, pos, d, Lp=Lp, deg=deg, noise=noise)
    density_grad = -get_grad(rgbo[..., -1], pos)
    rgbo_act = torch.cat((rgbo[..., :3], F.softplus(rgbo[..., 3:] + 0.5)), dim=-1)
    # train.py:182 passes mip_net.density_act positionally into `mul_norm`: a truthy object that is not `== True`, so the
    # depths are NOT scaled by |d| and the default ReLU is the density activation
    rendered, weights, _ = composite(rgbo_act, z_all, rays[:, 3:], mul_norm=False)
    n_loss = weighted_normal_loss(weights, density_grad, normal)
    bf = back_face_loss(weights, normal, d)
    cn_loss = weighted_normal_loss(pw, coarse_grad_select(density_grad, sort_ids, C).detach(), coarse_grad)
    img = torch.mean((rendered - rgb_tgt) ** 2)
    pl = proposal_loss(get_bounds(pw, below_all), weights.detach())
    loss = pl + img + 4e-4 * (n_loss + 0.1 * cn_loss) + 0.1 * bf
    return dict(z_fine=z_fine, z_merged=z_all, below_merged=below_all, sort_ids=sort_ids, rgbo_raw=rgbo, pred_normal=normal,
                density_grad=density_grad, coarse_grad=coarse_grad, weights=weights, rendered=rendered, normal_loss=n_loss, bf_loss=bf,
                coarse_normal_loss=cn_loss, img_loss=img, prop_loss=pl, loss=loss)


def render_rays_ref(prop_sd, ref_sd, rays: Tensor, u_strat: Tensor, u_inv: Tensor, near: float, far: float, sample_num: int = 128,
                    white_bkg: bool = False, cam_z: Optional[Tensor] = None, use_srgb: bool = False, contracted: bool = False,
                    Lp: int = 10, deg: int = 4):
    """Tile body of render_image for a RefNeRF (procedures.py:64-85, is_ref_model branch): coarse+fine merge, sigma ->
    softplus(sigma + 0.5), composite with relu (a no-op after softplus)."""
    z_c = stratified_render(near, far, sample_num, u_strat)
    pts_c = rays[:, None, :3] + z_c[..., None] * rays[:, None, 3:]
    if contracted:                                           # (not in the reference: the build's own definition, as in render_rays)
        pts_c = contract(pts_c)
    density = proposal_forward(prop_sd, pts_c)
    w_prop = max_blur(sigma_to_weights(density, z_c, rays[:, 3:]), 0.01)
    z_f, _ = inverse_sample(w_prop, z_c, u_inv, sort=True)
    samples, z_all = coarse_fine_merge(rays, z_c, z_f)
    if contracted:
        samples = torch.cat((contract(samples[..., :3]), samples[..., 3:]), dim=-1)
    rgbo, normal = ref_forward(ref_sd, samples, Lp=Lp, deg=deg, use_srgb=use_srgb)
    rgbo = torch.cat((rgbo[..., :3], F.softplus(rgbo[..., 3:] + 0.5)), dim=-1)
    rgb, w, extras = composite(rgbo, z_all, rays[:, 3:], white_bkg=white_bkg, render_depth=(near, far),
                               normal_info=(normal, cam_z) if cam_z is not None else None)
    return rgb, w, extras
