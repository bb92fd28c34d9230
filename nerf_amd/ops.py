"""Functional torch-tensor front end of the C-ABI (include/nerf_amd.h).  PyTorch is plumbing here:
it owns device memory and the HIP stream; every computation below is a hand-written HIP kernel in
libnerf_amd.so.  CPU tensors are rejected -- there is no fallback path."""
import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import lib, check, Samples, F32, BF16, BF16_F8, NET_PROPOSAL, NET_MIP, NET_REF, NET_PROPOSAL_128, NET_MIP_128, PROP_W128, FINE_W128, ACT_RELU, ACT_IDENTITY, ACT_SOFTPLUS

_PRECISION_OVERRIDE = None          # None -> follow torch autocast (on: bf16, off: fp32)


def set_precision(p: Optional[str]):
    """'fp32' (exact-fp32 MFMA, parity mode), 'bf16' (bf16 MFMA, fp32 accumulate) or None = follow
    ``torch.autocast`` like the reference's Linear layers do (train.py:202, procedures.py:150)."""
    global _PRECISION_OVERRIDE
    if p not in (None, "fp32", "bf16"):
        raise ValueError("precision must be None, 'fp32' or 'bf16'")
    _PRECISION_OVERRIDE = p


_TRAIN_DUMPS = __import__("os").environ.get("NERF_AMD_TRAIN_DUMPS", "bf16")      # (env: profiling scripts; the API is set_train_dumps)


def set_train_dumps(fmt: str):
    """Storage format of the training dumps (hidden activations written by the training forwards, deltas written by the dgrad chains,
    both read by the weight-gradient kernels) in 'bf16' precision mode: 'bf16' (default) or 'fp8' = OCP e4m3 with one power-of-two scale
    per sample and 16-feature group (NERF_AMD_BF16_F8): 288 instead of 512 bytes per sample and layer in each pass.  The arithmetic --
    forward, dgrad chain, MFMA weight gradients -- stays bf16 x bf16 -> fp32; only the weight gradients' operands are rounded."""
    global _TRAIN_DUMPS
    if fmt not in ("bf16", "fp8"):
        raise ValueError("train dumps: 'bf16' or 'fp8'")
    _TRAIN_DUMPS = fmt


def train_precision(precision: int) -> int:
    """precision code for the training entry points (forward_train / backward_chain / weight_grads) given the arithmetic precision"""
    return BF16_F8 if (precision == BF16 and _TRAIN_DUMPS == "fp8") else precision


def current_precision() -> int:
    if _PRECISION_OVERRIDE is not None:
        return BF16 if _PRECISION_OVERRIDE == "bf16" else F32
    return BF16 if torch.is_autocast_enabled() else F32


# torch.cuda.current_stream() builds a Stream object through three layers of Python (~10 us; thirteen calls per 1 024-ray training
# step = 9 % of the eager iteration's host time, scripts/gpu_host_profile.py); the raw handle comes straight from the C extension.
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return _RAW_STREAM(_CUR_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("nerf_amd: '%s' must live on the HIP device (got %s); there is no CPU path" % (name, t.device))
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _prop_prec(packed_prop: torch.Tensor, precision: int, packed_fine: Optional[torch.Tensor] = None) -> int:
    """precision argument of a call that takes packed blobs: + the blobs' layout flags (PROP_W128 for a NET_PROPOSAL_128 blob, FINE_W128
    for a NET_MIP_128 blob; a blob's flag rides on the tensor object: PackedWeightsMixin._pack_now)"""
    return int(precision) | int(getattr(packed_prop, "_nerf_amd_layout", 0)) | int(getattr(packed_fine, "_nerf_amd_layout", 0))


# ------------------------------------------------------------------------------------------------ persistent training buffers
# A 16 384-ray training step needs ~23 GB of dump / delta / workspace buffers.  They are OWNED here -- one grow-only buffer per
# (device, stream, purpose), sized by the largest step seen -- instead of being torch.empty()-allocated and freed every step (round 3:
# the driver's training-step line was 4 ms slower than the kernels' sum; bench.py's train_step now reports both).  Two kinds:
#   * scratch(tag, ...): contents die inside the call chain that asked for them (the delta dump between the dgrad chain and the
#     weight-gradient kernels, the products' partial sums): always the persistent buffer;
#   * leased(tag, ...): alive from a training forward to its backward (the activation dump).  The persistent buffer is handed out while
#     no earlier lease on it is alive (the returned view carries the lease; it ends when the view is freed, i.e. when the backward has
#     consumed the dump or the graph is dropped); a second forward before that gets a fresh allocation, so nothing is ever overwritten.
# While a hipGraph is being captured both fall back to torch.empty: the graph's private pool then owns the memory it replays into.
_PERSISTENT = __import__("os").environ.get("NERF_AMD_PERSISTENT_BUFFERS", "1") != "0"
_ARENA = {}
_ARENA_BUSY = {}                # key -> id of the lease that owns the persistent buffer (identity, not just "somebody": a lease that outlives
                                # release_buffers() must not free the key for a NEWER lease when it dies)
ARENA_STATS = {"persistent": 0, "fresh": 0, "grown": 0}


def set_persistent_buffers(on: bool) -> None:
    """Turn the persistent training buffers on / off (off: every call allocates through torch's caching allocator, as in round 3)."""
    global _PERSISTENT
    _PERSISTENT = bool(on)
    if not on:
        release_buffers()


def release_buffers() -> None:
    """Give the persistent training buffers back to torch's allocator (they are re-created on demand).  A dump still leased to a pending
    backward keeps its storage alive through its own view; its lease no longer owns any key afterwards (it cannot un-busy a newer lease)."""
    _ARENA.clear()
    _ARENA_BUSY.clear()


def _arena_key(tag, device):
    return (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream, tag)


def _arena_buffer(key, nbytes: int, device) -> torch.Tensor:
    buf = _ARENA.get(key)
    if buf is None or buf.numel() < nbytes:
        _ARENA.pop(key, None)
        buf = None                                             # (free the smaller one before allocating its replacement)
        buf = _ARENA[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        ARENA_STATS["grown"] += 1
    return buf


def scratch(tag, nbytes: int, device) -> torch.Tensor:
    nbytes = int(nbytes)
    device = torch.device(device)
    if nbytes == 0 or not _PERSISTENT or torch.cuda.is_current_stream_capturing():
        ARENA_STATS["fresh"] += 1
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    ARENA_STATS["persistent"] += 1
    return _arena_buffer(_arena_key(tag, device), nbytes, device)[:nbytes]


class _Lease:
    __slots__ = ("key",)

    def __init__(self, key):
        self.key = key
        _ARENA_BUSY[key] = id(self)

    def __del__(self):
        try:
            if _ARENA_BUSY.get(self.key) == id(self):          # only the owner frees the key
                del _ARENA_BUSY[self.key]
        except Exception:                                      # (interpreter shutdown: the module's globals may be gone already)
            pass


def leased(tag, nbytes: int, device) -> torch.Tensor:
    nbytes = int(nbytes)
    device = torch.device(device)
    if nbytes == 0 or not _PERSISTENT or torch.cuda.is_current_stream_capturing():
        ARENA_STATS["fresh"] += 1
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    key = _arena_key(tag, device)
    if key in _ARENA_BUSY:                                     # an earlier forward's dump is still waiting for its backward
        ARENA_STATS["fresh"] += 1
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    view = _arena_buffer(key, nbytes, device)[:nbytes]
    view._nerf_amd_lease = _Lease(key)                         # the lease lives exactly as long as this view object
    ARENA_STATS["persistent"] += 1
    return view


# ------------------------------------------------------------------------------------------------ weights
def _raw(t: torch.Tensor, name: str) -> torch.Tensor:
    """a parameter as the kernels read it: the tensor itself when it already is contiguous fp32 on the device (the usual case: no detach /
    conversion objects per tensor -- 44 of them per training step), else `_dev`'s converted copy"""
    if t.is_cuda and t.dtype is torch.float32 and t.is_contiguous():
        return t
    return _dev(t.detach(), name)


def pack_weights(net: int, precision: int, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor]) -> torch.Tensor:
    ws = [_raw(w, "weight") for w in weights]
    bs = [_raw(b, "bias") for b in biases]
    nbytes = lib.nerf_amd_packed_bytes(net, precision)
    packed = torch.empty(nbytes, dtype=torch.uint8, device=ws[0].device)
    n = len(ws)
    wp = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * n)(*[b.data_ptr() for b in bs])
    check(lib.nerf_amd_pack_weights(net, precision, wp, bp, n, _ptr(packed), _stream()), "nerf_amd_pack_weights")
    return packed


# ------------------------------------------------------------------------------------------------ MLPs
def _samples_pts(pts: torch.Tensor, stride: int, contract: bool = False) -> Samples:
    s = Samples()
    s.mode = 0
    s.M = pts.numel() // stride
    s.pts = pts.data_ptr()
    s.pts_stride = stride
    s.contract = int(bool(contract))
    return s


def samples_rays(rays: torch.Tensor, S: int, z: Optional[torch.Tensor] = None, z_base: Optional[torch.Tensor] = None,
                 u: Optional[torch.Tensor] = None, z_jitter: float = 0.0, contract: bool = False, ipe_radius: Optional[float] = None,
                 ipe_dir_norm: Optional[torch.Tensor] = None, seed: int = 0, rng_ray_offset: int = 0) -> Samples:
    """`contract`: Mip-NeRF 360 scene contraction of the sample positions before encoding (not in the reference; BASELINE config 5).
    `ipe_radius` (with explicit z of S+1 depths per ray): integrated PE of the frusta between consecutive depths (mip_methods.py:15-58);
    `ipe_dir_norm` = ops.dirs_norm(rays) -- the caller keeps that tensor alive until the kernel has run."""
    s = Samples()
    s.contract = int(bool(contract))
    if ipe_radius is not None:
        if z is None or z.shape[-1] < S + 1 or ipe_dir_norm is None:
            raise ValueError("nerf_amd: integrated PE needs explicit z with S+1 depths per ray and ipe_dir_norm")
        s.ipe, s.ipe_radius, s.ipe_dir_norm = 1, float(ipe_radius), ipe_dir_norm.data_ptr()
    s.mode = 1
    s.S = S
    s.M = rays.shape[0] * S
    s.rays = rays.data_ptr()
    if z is not None:
        s.z = z.data_ptr()
        s.z_stride = z.shape[-1]
    else:
        s.z_base = z_base.data_ptr()
        if u is not None:
            s.u = u.data_ptr()
        else:                                                # stratified uniforms drawn in the kernel (Philox4x32-10)
            s.rng_seed, s.rng_ray_offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(rng_ray_offset)
        s.z_jitter = z_jitter
        s.z_stride = S
    return s


def proposal_forward(packed: torch.Tensor, precision: int, pts: torch.Tensor, contract: bool = False) -> torch.Tensor:
    """pts (..., 3) -> density (...)   [addtional.py:88-96]"""
    pts = _dev(pts, "pts")
    out = torch.empty(pts.shape[:-1], dtype=torch.float32, device=pts.device)
    if out.numel() == 0:
        return out
    s = _samples_pts(pts, 3, contract)
    check(lib.nerf_amd_proposal_forward(_ptr(packed), _prop_prec(packed, precision), C.byref(s), _ptr(out), _stream()), "nerf_amd_proposal_forward")
    return out


def proposal_forward_samples(packed, precision, s: Samples, shape, device) -> torch.Tensor:
    out = torch.empty(shape, dtype=torch.float32, device=device)
    check(lib.nerf_amd_proposal_forward(_ptr(packed), _prop_prec(packed, precision), C.byref(s), _ptr(out), _stream()), "nerf_amd_proposal_forward")
    return out


def mip_forward(packed: torch.Tensor, precision: int, pts: torch.Tensor, contract: bool = False) -> torch.Tensor:
    """pts (..., 6) -> rgbo (..., 4)   [mip_model.py:41-60]"""
    pts = _dev(pts, "pts")
    out = torch.empty(pts.shape[:-1] + (4,), dtype=torch.float32, device=pts.device)
    if out.numel() == 0:
        return out
    s = _samples_pts(pts, 6, contract)
    check(lib.nerf_amd_mip_forward(_ptr(packed), _prop_prec(None, precision, packed), C.byref(s), _ptr(out), _stream()), "nerf_amd_mip_forward")
    return out


def mip_forward_samples(packed, precision, s: Samples, shape, device) -> torch.Tensor:
    out = torch.empty(tuple(shape) + (4,), dtype=torch.float32, device=device)
    check(lib.nerf_amd_mip_forward(_ptr(packed), _prop_prec(None, precision, packed), C.byref(s), _ptr(out), _stream()), "nerf_amd_mip_forward")
    return out


def mip_forward_composite(packed, precision, rays: torch.Tensor, z: torch.Tensor, n_samples: int, white_bkg: bool, near: float,
                          far: float, want_depth: bool = True, want_weights: bool = False):
    """Fine MLP + alpha compositing in one launch (rows 8-10); z (N, >= n_samples) row stride = z.shape[-1].  A NET_MIP_128 blob (the
    default `packed()` of a fine network of hidden width <= 128) has no fused-epilogue kernel: it takes the two launches of the render
    path (mip128_kernel + composite_kernel), which the fused epilogue is bit-compatible with."""
    N = rays.shape[0]
    dev = rays.device
    if int(getattr(packed, "_nerf_amd_layout", 0)) & FINE_W128:
        rgbo = mip_forward_samples(packed, precision, samples_rays(rays, n_samples, z=z), (N, n_samples), dev)
        rgb, w, depth, _ = composite(rgbo, z, rays, True, bool(white_bkg), ACT_RELU, (near, far), want_weights=want_weights)
        return rgb, (depth if want_depth else None), (w if want_weights else None)
    rgb = torch.empty((N, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((N,), dtype=torch.float32, device=dev) if want_depth else None
    w = torch.empty((N, n_samples), dtype=torch.float32, device=dev) if want_weights else None
    s = samples_rays(rays, n_samples, z=z)
    check(lib.nerf_amd_mip_forward_composite(_ptr(packed), _prop_prec(None, precision, packed), C.byref(s), int(white_bkg), float(near), float(far), _ptr(rgb),
                                             _ptr(depth), _ptr(w), _stream()), "nerf_amd_mip_forward_composite")
    return rgb, depth, w


def _ref_out(shape, device, want_normal):
    rgbo = torch.empty(tuple(shape) + (4,), dtype=torch.float32, device=device)
    normal = torch.empty(tuple(shape) + (3,), dtype=torch.float32, device=device) if want_normal else None
    return rgbo, normal


REF_SRGB = 1          # NERF_AMD_REF_SRGB: RefNeRF(use_srgb=True) output transform (ref_model.py:100-102)


def ref_forward(packed: torch.Tensor, precision: int, pts: torch.Tensor, want_normal: bool = True, noise: Optional[torch.Tensor] = None,
                flags: int = 0, contract: bool = False):
    """pts (..., 6) = [position | direction] -> (rgbo (..., 4), normal (..., 3))   [ref_model.py:68-106]; `noise` (..., 128) = the
    train-mode bottle-neck perturbation (ref_model.py:84-85), None = eval mode; `flags`: REF_SRGB = the module's use_srgb"""
    pts = _dev(pts, "pts")
    rgbo, normal = _ref_out(pts.shape[:-1], pts.device, want_normal)
    if rgbo.numel() == 0:
        return rgbo, normal
    s = _samples_pts(pts, 6, contract)
    if noise is None:
        check(lib.nerf_amd_ref_forward(_ptr(packed), precision, C.byref(s), int(flags), _ptr(rgbo), _ptr(normal), _stream()), "nerf_amd_ref_forward")
    else:
        noise = _dev(noise, "noise")
        if noise.numel() != s.M * 128:
            raise ValueError("nerf_amd: bottle-neck noise must be (..., 128) over the same samples")
        check(lib.nerf_amd_ref_forward_train(_ptr(packed), precision, C.byref(s), int(flags), _ptr(noise), _ptr(rgbo), _ptr(normal), _stream()),
              "nerf_amd_ref_forward_train")
    return rgbo, normal


def ref_forward_samples(packed, precision, s: Samples, shape, device, want_normal: bool = True, flags: int = 0):
    rgbo, normal = _ref_out(shape, device, want_normal)
    check(lib.nerf_amd_ref_forward(_ptr(packed), precision, C.byref(s), int(flags), _ptr(rgbo), _ptr(normal), _stream()), "nerf_amd_ref_forward")
    return rgbo, normal


# ------------------------------------------------------------------------------------------------ sampling ops
def positional_encoding(x: torch.Tensor, L: int) -> torch.Tensor:
    x = _dev(x, "x")
    out = torch.empty(x.shape[:-1] + (6 * L,), dtype=torch.float32, device=x.device)
    check(lib.nerf_amd_positional_encoding(_ptr(x), x.numel() // 3, L, _ptr(out), _stream()), "nerf_amd_positional_encoding")
    return out


def dirs_norm(rays: torch.Tensor) -> torch.Tensor:
    """Norm of the whole (N,3) direction tensor of a ray table (mip_methods.py:31's `.norm()` without a dim) -> (1,) on the device."""
    rays = _dev(rays, "rays")
    out = torch.empty((1,), dtype=torch.float32, device=rays.device)
    check(lib.nerf_amd_dirs_norm(_ptr(rays), rays.shape[0], _ptr(out), _stream()), "nerf_amd_dirs_norm")
    return out


def cone_parameters(z: torch.Tensor, r: float):
    """coneParameters (mip_methods.py:15-23): z (N,S+1) -> mu_t, sigma_t^2, sigma_r^2 (N,S)."""
    z = _dev(z, "zvals")
    N, S = z.shape[0], z.shape[1] - 1
    out = [torch.empty((N, S), dtype=torch.float32, device=z.device) for _ in range(3)]
    check(lib.nerf_amd_cone_parameters(_ptr(z), N, S, float(r), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _stream()), "nerf_amd_cone_parameters")
    return tuple(out)


def ipe_feature(z: torch.Tensor, rays: torch.Tensor, L: int, r: float, dir_norm: Optional[torch.Tensor] = None, contract: bool = False):
    """ipe_feature (mip_methods.py:47-58): z (N,S+1), rays (N,6) -> feat (N,S,6L), mu (N,S,3), mu_t (N,S).  `contract` (an addition): the
    frustum mean goes through the Mip-NeRF 360 contraction before the lift (nerf_amd_ipe_feature_contracted); `mu` is the contracted mean."""
    z, rays = _dev(z, "zvals"), _dev(rays, "cam_rays")
    N, S = z.shape[0], z.shape[1] - 1
    if dir_norm is None:
        dir_norm = dirs_norm(rays)
    feat = torch.empty((N, S, 6 * L), dtype=torch.float32, device=z.device)
    mu = torch.empty((N, S, 3), dtype=torch.float32, device=z.device)
    mu_t = torch.empty((N, S), dtype=torch.float32, device=z.device)
    fn, name = (lib.nerf_amd_ipe_feature_contracted, "nerf_amd_ipe_feature_contracted") if contract else (lib.nerf_amd_ipe_feature, "nerf_amd_ipe_feature")
    check(fn(_ptr(z), _ptr(rays), N, S, L, float(r), _ptr(dir_norm), _ptr(feat), _ptr(mu), _ptr(mu_t), _stream()), name)
    return feat, mu, mu_t


def generate_rays(pose: torch.Tensor, H: int, W: int, fx: float, fy: float, device, ray_offset: int = 0,
                  n: Optional[int] = None) -> torch.Tensor:
    n = H * W - ray_offset if n is None else n
    host = (C.c_float * 12)(*pose.detach().float().cpu().reshape(-1).tolist()[:12])
    rays = torch.empty((n, 6), dtype=torch.float32, device=device)
    check(lib.nerf_amd_generate_rays(host, H, W, float(fx), float(fy), ray_offset, n, _ptr(rays), _stream()), "nerf_amd_generate_rays")
    return rays


def length2pts(rays: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    rays, z = _dev(rays, "rays"), _dev(z, "z")
    N, S = z.shape
    out = torch.empty((N, S, 6), dtype=torch.float32, device=z.device)
    check(lib.nerf_amd_length2pts(_ptr(rays), _ptr(z), N, S, _ptr(out), _stream()), "nerf_amd_length2pts")
    return out


def sigma_to_weights(sigma: torch.Tensor, z: torch.Tensor, dirs: Optional[torch.Tensor], act: int = ACT_RELU) -> torch.Tensor:
    sigma, z = _dev(sigma, "sigma"), _dev(z, "z")
    dirs = None if dirs is None else _dev(dirs, "dirs")
    N, S = z.shape
    w = torch.empty((N, S), dtype=torch.float32, device=z.device)
    check(lib.nerf_amd_sigma_to_weights(_ptr(sigma), _ptr(z), _ptr(dirs), N, S, act, _ptr(w), _stream()), "nerf_amd_sigma_to_weights")
    return w


def max_blur(w: torch.Tensor, alpha: float) -> torch.Tensor:
    w = _dev(w, "weights")
    S = w.shape[-1]
    out = torch.empty_like(w)
    check(lib.nerf_amd_max_blur(_ptr(w), w.numel() // S, S, float(alpha), _ptr(out), _stream()), "nerf_amd_max_blur")
    return out


def inverse_sample(w: torch.Tensor, z: torch.Tensor, u: torch.Tensor, sort: bool, want_below: bool = True):
    w, z, u = _dev(w, "weights"), _dev(z, "z"), _dev(u, "u")
    N, Cn = z.shape
    K = u.shape[-1]
    z_out = torch.empty((N, K), dtype=torch.float32, device=z.device)
    below = torch.empty((N, K), dtype=torch.int64, device=z.device) if want_below else None
    check(lib.nerf_amd_inverse_sample(_ptr(w), _ptr(z), _ptr(u), N, Cn, K, int(sort), _ptr(z_out), _ptr(below), _stream()),
          "nerf_amd_inverse_sample")
    return z_out, below


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, u: torch.Tensor):
    bins, weights, u = _dev(bins, "bins"), _dev(weights, "weights"), _dev(u, "u")
    N, B = bins.shape
    K = u.shape[-1]
    out = torch.empty((N, K), dtype=torch.float32, device=bins.device)
    below = torch.empty((N, K), dtype=torch.int64, device=bins.device)
    above = torch.empty((N, K), dtype=torch.int64, device=bins.device)
    check(lib.nerf_amd_sample_pdf(_ptr(bins), _ptr(weights), _ptr(u), N, B, K, _ptr(out), _ptr(below), _ptr(above), _stream()),
          "nerf_amd_sample_pdf")
    return out, below, above


def pixel_rays(coords: torch.Tensor, pose: torch.Tensor, fx: float, fy: float) -> torch.Tensor:
    if not coords.is_cuda:
        raise RuntimeError("nerf_amd: 'coords' must live on the HIP device")
    coords = coords.to(torch.int64).contiguous()
    host = (C.c_float * 12)(*pose.detach().float().cpu().reshape(-1).tolist()[:12])
    rays = torch.empty((coords.shape[0], 6), dtype=torch.float32, device=coords.device)
    check(lib.nerf_amd_pixel_rays(host, float(fx), float(fy), _ptr(coords), coords.shape[0], _ptr(rays), _stream()), "nerf_amd_pixel_rays")
    return rays


def sample_training_rays(rgbs: torch.Tensor, coords: torch.Tensor, pose: torch.Tensor, fx: float, fy: float, near: float, far: float, n_rays: int,
                         n_points: int, seed: int, want_samples: bool = True):
    """validSampler (utils.py:72-94) in one launch, random numbers drawn in the kernel (Philox, `seed`).
    -> (pts (N,C,3) | None, lengths (N,C) | None, rgb (N,3), rays (N,6))"""
    rgbs = _dev(rgbs, "rgbs")
    if not coords.is_cuda:
        raise RuntimeError("nerf_amd: 'coords' must live on the HIP device")
    coords = coords.to(torch.int64).contiguous()
    dev = rgbs.device
    host = (C.c_float * 12)(*pose.detach().float().cpu().reshape(-1).tolist()[:12])
    pts = torch.empty((n_rays, n_points, 3), dtype=torch.float32, device=dev) if want_samples else None
    z = torch.empty((n_rays, n_points), dtype=torch.float32, device=dev) if want_samples else None
    rgb = torch.empty((n_rays, 3), dtype=torch.float32, device=dev)
    rays = torch.empty((n_rays, 6), dtype=torch.float32, device=dev)
    check(lib.nerf_amd_sample_training_rays(_ptr(rgbs), _ptr(coords), coords.shape[0], host, float(fx), float(fy), float(near), float(far), n_rays,
                                            n_points, int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(pts), _ptr(z), _ptr(rgb), _ptr(rays), _stream()),
          "nerf_amd_sample_training_rays")
    return pts, z, rgb, rays


def sample_training_rays_dev(rgbs: torch.Tensor, coords: torch.Tensor, pose_dev: torch.Tensor, fx: float, fy: float, near: float, far: float,
                             n_rays: int, n_points: int, seed_dev: torch.Tensor, want_samples: bool = True):
    """sample_training_rays with the pose (3,4) fp32 and the seed (1,) int64 read from device memory when the kernel RUNS: the launch is
    replayable from a captured hipGraph with new images / fresh random numbers (nerf_amd/training.py)."""
    rgbs = _dev(rgbs, "rgbs")
    pose_dev = _dev(pose_dev, "pose_dev")
    if not (coords.is_cuda and coords.dtype == torch.int64 and coords.is_contiguous()):
        raise RuntimeError("nerf_amd: 'coords' must be a contiguous int64 tensor on the HIP device")
    if not (seed_dev.is_cuda and seed_dev.dtype == torch.int64 and seed_dev.numel() == 1) or pose_dev.numel() < 12:
        raise RuntimeError("nerf_amd: seed_dev = one int64 on the device, pose_dev = (3,4) fp32 on the device")
    dev = rgbs.device
    pts = torch.empty((n_rays, n_points, 3), dtype=torch.float32, device=dev) if want_samples else None
    z = torch.empty((n_rays, n_points), dtype=torch.float32, device=dev) if want_samples else None
    rgb = torch.empty((n_rays, 3), dtype=torch.float32, device=dev)
    rays = torch.empty((n_rays, 6), dtype=torch.float32, device=dev)
    check(lib.nerf_amd_sample_training_rays_dev(_ptr(rgbs), _ptr(coords), coords.shape[0], _ptr(pose_dev), float(fx), float(fy), float(near), float(far),
                                                n_rays, n_points, _ptr(seed_dev), _ptr(pts), _ptr(z), _ptr(rgb), _ptr(rays), _stream()),
          "nerf_amd_sample_training_rays_dev")
    return pts, z, rgb, rays


def philox_uniforms(shape, seed: int = 0, seed_dev: Optional[torch.Tensor] = None, device=None) -> torch.Tensor:
    """u (N,K) in [0,1): the kernels' inverse-CDF Philox stream as a tensor (key = seed, or *seed_dev read at run time)."""
    N, K = int(shape[0]), int(shape[1])
    dev = seed_dev.device if seed_dev is not None else (device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    out = torch.empty((N, K), dtype=torch.float32, device=dev)
    check(lib.nerf_amd_philox_uniforms(_ptr(out), N, K, int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(seed_dev), _stream()), "nerf_amd_philox_uniforms")
    return out


def philox_stream(shape, seed: int, ray_offset: int = 0, strat: bool = False, device=None) -> torch.Tensor:
    """u (N,K) of GLOBAL rays ray_offset .. ray_offset+N-1: the render kernels' stratified-jitter (strat, K <= 64) or inverse-CDF Philox
    stream, bit-identical to their in-place draws for the same seed (nerf_amd_philox_stream)."""
    N, K = int(shape[0]), int(shape[1])
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    out = torch.empty((N, K), dtype=torch.float32, device=dev)
    check(lib.nerf_amd_philox_stream(_ptr(out), N, K, int(seed) & 0xFFFFFFFFFFFFFFFF, None, int(ray_offset), 1 if strat else 0, _stream()),
          "nerf_amd_philox_stream")
    return out


def advance_seed(seed_dev: torch.Tensor) -> None:
    """*seed_dev <- an unrelated key for the next step (in place, on the current stream)."""
    if not (seed_dev.is_cuda and seed_dev.dtype == torch.int64 and seed_dev.numel() == 1):
        raise RuntimeError("nerf_amd: seed_dev = one int64 on the device")
    check(lib.nerf_amd_advance_seed(_ptr(seed_dev), _stream()), "nerf_amd_advance_seed")


def stratified_points(rays: torch.Tensor, z_base: torch.Tensor, u: torch.Tensor, jitter: float, want_pts: bool = True):
    rays, z_base, u = _dev(rays, "rays"), _dev(z_base, "z_base"), _dev(u, "u")
    N, S = u.shape
    z = torch.empty((N, S), dtype=torch.float32, device=u.device)
    pts = torch.empty((N, S, 3), dtype=torch.float32, device=u.device) if want_pts else None
    check(lib.nerf_amd_stratified_points(_ptr(rays), _ptr(z_base), _ptr(u), float(jitter), N, S, _ptr(z), _ptr(pts), _stream()),
          "nerf_amd_stratified_points")
    return z, pts


def composite(rgbo: torch.Tensor, z: torch.Tensor, dirs: torch.Tensor, mul_norm: bool, white_bkg: bool, act: int,
              near_far=None, normal: Optional[torch.Tensor] = None, cam_dir: Optional[torch.Tensor] = None,
              want_weights: bool = True, sigma_shift: float = 0.0):
    """NeRF.render (nerf_base.py:91-113).  ``dirs`` is (N,3) ray directions or the (N,6) ray table (then the
    direction half is read in place with stride 6)."""
    rgbo, z = _dev(rgbo, "rgbo"), _dev(z, "depth")
    N, S = rgbo.shape[0], rgbo.shape[1]
    dev = rgbo.device
    if dirs.shape[-1] == 6:
        dirs = _dev(dirs, "rays")
        dirs_ptr, dirs_stride = C.c_void_p(dirs.data_ptr() + 12), 6
    else:
        dirs = _dev(dirs, "ray_dirs")
        dirs_ptr, dirs_stride = _ptr(dirs), 3
    rgb = torch.empty((N, 3), dtype=torch.float32, device=dev)
    w = torch.empty((N, S), dtype=torch.float32, device=dev) if want_weights else None
    depth = torch.empty((N,), dtype=torch.float32, device=dev) if near_far is not None else None
    nimg = None
    if normal is not None:
        normal, cam_dir = _dev(normal, "normal"), _dev(cam_dir.reshape(-1), "cam_dir")
        nimg = torch.empty((N,), dtype=torch.float32, device=dev)
    near, far = (near_far if near_far is not None else (0.0, 1.0))
    flags = (1 if mul_norm else 0) | (2 if white_bkg else 0)
    check(lib.nerf_amd_composite(_ptr(rgbo), _ptr(z), z.shape[-1], dirs_ptr, dirs_stride, N, S, flags, act, float(sigma_shift),
                                 float(near), float(far), _ptr(normal), _ptr(cam_dir), _ptr(rgb), _ptr(w), _ptr(depth), _ptr(nimg),
                                 _stream()), "nerf_amd_composite")
    return rgb, w, depth, nimg


def merge_depths_order(z_fine: torch.Tensor, z_coarse: torch.Tensor, f_inds: Optional[torch.Tensor] = None):
    """NeRF.coarseFineMerge's sort with its order (nerf_base.py:59-73, train.py:176) -> (z (N, K+C-1), order (N, K+C) int64,
    all_inds (N, K+C) int64 or None): one merge kernel instead of torch.sort + arange + cat + gather."""
    z_fine, z_coarse = _dev(z_fine, "z_fine"), _dev(z_coarse, "z_coarse")
    N, K = z_fine.shape
    Cn = z_coarse.shape[-1]
    dev = z_fine.device
    out = torch.empty((N, K + Cn - 1), dtype=torch.float32, device=dev)
    order = torch.empty((N, K + Cn), dtype=torch.int64, device=dev)
    all_inds = None
    if f_inds is not None:
        f_inds = _dev(f_inds, "f_inds", torch.int64)
        all_inds = torch.empty((N, K + Cn), dtype=torch.int64, device=dev)
    check(lib.nerf_amd_merge_depths_order(_ptr(z_fine), _ptr(z_coarse), _ptr(f_inds) if f_inds is not None else None, N, K, Cn, _ptr(out), _ptr(order),
                                          _ptr(all_inds) if all_inds is not None else None, _stream()), "nerf_amd_merge_depths_order")
    return out, order, all_inds


def coarse_grad_select(fine_grads: torch.Tensor, sort_inds: torch.Tensor, c_pnum: int) -> torch.Tensor:
    """RefNeRF.coarse_grad_select (ref_model.py:108-117) on the device: (N,T,D), (N,T) int64 -> (N,c_pnum,D)"""
    fine_grads, sort_inds = _dev(fine_grads, "fine_grads"), _dev(sort_inds, "sort_inds", torch.int64)
    N, T, D = fine_grads.shape
    out = torch.empty((N, int(c_pnum), D), dtype=torch.float32, device=fine_grads.device)
    check(lib.nerf_amd_coarse_grad_select(_ptr(fine_grads), _ptr(sort_inds), N, T, D, int(c_pnum), _ptr(out), _stream()), "nerf_amd_coarse_grad_select")
    return out


def mfma_stream(iters: int, workgroups: int, device, mode: int = 0) -> None:
    """launch the MFMA-only stream (bench.py's measured ceiling): iters * 64 MFMAs of 32 768 flop per wave, 4 waves per workgroup.
    mode 0 constant operands, 1 random operands, 2 random weights x post-ReLU-like activations, 3 = 2 with the A operands read from LDS"""
    sink = torch.zeros((256,), dtype=torch.float32, device=device)
    check(lib.nerf_amd_mfma_stream(int(iters), int(workgroups), int(mode), _ptr(sink), _stream()), "nerf_amd_mfma_stream")


DOT_LOSS_WORKSPACE_FLOATS = 512


def weighted_dot_loss(w: torch.Tensor, a: torch.Tensor, b: torch.Tensor, mode: int, scale: float) -> torch.Tensor:
    """scale * sum w f(<a, b>) (f = 1 - x: WeightedNormalLoss, mode 0; relu: BackFaceLoss, mode 1; ref_model.py:127-143) -> 0-dim tensor"""
    w, a, b = _dev(w, "weight"), _dev(a, "a"), _dev(b, "b")
    M = w.numel()
    if a.numel() != 3 * M or b.numel() != 3 * M:
        raise ValueError("nerf_amd: weighted_dot_loss needs weight (...,) and two (..., 3) tensors of the same leading shape")
    out = torch.empty((1,), dtype=torch.float32, device=w.device)
    ws = torch.empty((DOT_LOSS_WORKSPACE_FLOATS,), dtype=torch.float32, device=w.device)
    check(lib.nerf_amd_weighted_dot_loss(_ptr(w), _ptr(a), _ptr(b), M, int(mode), float(scale), _ptr(out), _ptr(ws), _stream()), "nerf_amd_weighted_dot_loss")
    return out.reshape(())


def weighted_dot_loss_backward(g: torch.Tensor, w: torch.Tensor, a: torch.Tensor, b: torch.Tensor, mode: int, scale: float, need=(True, True, True)):
    """gradients of weighted_dot_loss w.r.t. (w, a, b) for the upstream gradient g (a 0-dim device tensor); `need` picks which are produced"""
    g, w, a, b = _dev(g, "g").reshape(1), _dev(w, "weight"), _dev(a, "a"), _dev(b, "b")
    M = w.numel()
    d_w = torch.empty_like(w) if need[0] else None
    d_a = torch.empty_like(a) if need[1] else None
    d_b = torch.empty_like(b) if need[2] else None
    check(lib.nerf_amd_weighted_dot_loss_backward(_ptr(g), _ptr(w), _ptr(a), _ptr(b), M, int(mode), float(scale), _ptr(d_w), _ptr(d_a), _ptr(d_b), _stream()),
          "nerf_amd_weighted_dot_loss_backward")
    return d_w, d_a, d_b


def merge_depths(z_fine: torch.Tensor, z_coarse: torch.Tensor) -> torch.Tensor:
    """sort(cat(z_fine, z_coarse))[..., :-1] (the render path of coarseFineMerge): a merge when both sets are ascending, which they
    normally are; rays with an out-of-order input are sorted first."""
    z_fine, z_coarse = _dev(z_fine, "z_fine"), _dev(z_coarse, "z_coarse")
    N, K = z_fine.shape
    Cn = z_coarse.shape[-1]
    out = torch.empty((N, K + Cn - 1), dtype=torch.float32, device=z_fine.device)
    check(lib.nerf_amd_merge_depths(_ptr(z_fine), _ptr(z_coarse), N, K, Cn, _ptr(out), _stream()), "nerf_amd_merge_depths")
    return out


def get_bounds(w_prop: torch.Tensor, below: torch.Tensor) -> torch.Tensor:
    w_prop = _dev(w_prop, "weights")
    if not below.is_cuda:
        raise RuntimeError("nerf_amd: 'inds' must live on the HIP device")
    below = below.to(torch.int64).contiguous()
    N, Cn = w_prop.shape
    K = below.shape[-1]
    out = torch.empty((N, K - 1), dtype=torch.float32, device=w_prop.device)
    check(lib.nerf_amd_get_bounds(_ptr(w_prop), _ptr(below), N, Cn, K, _ptr(out), _stream()), "nerf_amd_get_bounds")
    return out


def resample(density, z, z_base, u_strat, z_jitter, rays, u_inv, K, softplus=False, alpha=0.01, want_below=False,
             want_w=False, want_zc=False, seed: int = 0, ray_offset: int = 0):
    """Fused rows 5-7 (procedures.py:68-70 / train.py:169-172).  rays (N,6).  u_strat / u_inv = None: drawn in the kernel from
    (seed, ray + ray_offset) (Philox4x32-10, include/nerf_amd.h)."""
    N, Cn = density.shape
    dev = density.device
    z_fine = torch.empty((N, K), dtype=torch.float32, device=dev)
    below = torch.empty((N, K), dtype=torch.int64, device=dev) if want_below else None
    w = torch.empty((N, Cn), dtype=torch.float32, device=dev) if want_w else None
    zc = torch.empty((N, Cn), dtype=torch.float32, device=dev) if want_zc else None
    dirs_ptr = C.c_void_p(rays.data_ptr() + 12)
    check(lib.nerf_amd_resample(_ptr(density), _ptr(z), _ptr(z_base), _ptr(u_strat), float(z_jitter), dirs_ptr, 6, _ptr(u_inv),
                                N, Cn, K, int(softplus), float(alpha), int(seed) & 0xFFFFFFFFFFFFFFFF, int(ray_offset), _ptr(z_fine),
                                _ptr(below), _ptr(w), _ptr(zc), _stream()),
          "nerf_amd_resample")
    return z_fine, below, w, zc


def render_rays(packed_prop, packed_mip, precision, rays, z_base, u_strat, u_inv, n_fine, near, far, white_bkg,
                want_depth=True, want_weights=False, workspace: Optional[torch.Tensor] = None, camera: Optional[Samples] = None,
                ray_offset: int = 0, n_rays: Optional[int] = None, contract: bool = False, ipe_radius: Optional[float] = None,
                seed: Optional[int] = None, rng_ray_offset: int = 0, ipe_dir_norm: Optional[torch.Tensor] = None):
    """The tile body of render_image (procedures.py:64-85) for all given rays in four launches.  `contract`: Mip-NeRF 360 scene
    contraction of every sample position (proposal and fine) before encoding.  `ipe_radius`: the fine pass encodes the conical frusta
    between consecutive fine depths with the integrated PE (mip_methods.py:15-58; explicit `rays` required).
    u_strat = u_inv = None with `seed`: every uniform is drawn inside the kernels (Philox4x32-10 keyed by `seed`, a pure function of
    (ray index + rng_ray_offset, sample)); pass `n_rays` (or rays) for the ray count."""
    in_kernel_rng = u_strat is None
    if in_kernel_rng:
        if u_inv is not None or seed is None:
            raise ValueError("nerf_amd: u_strat and u_inv are both tensors, or both None with a `seed`")
        dev = rays.device if rays is not None else z_base.device
    else:
        dev = u_strat.device
    if (contract or ipe_radius is not None or in_kernel_rng) and camera is None:
        camera = Samples()                                   # carries only the flags next to explicit rays
    if camera is not None:
        camera.contract = int(bool(contract))
        if ipe_radius is not None:
            if rays is None:
                raise ValueError("nerf_amd: integrated PE needs an explicit ray table")
            camera.ipe, camera.ipe_radius = 1, float(ipe_radius)
            # the direction norm of mip_methods.py:31 is over ALL rays of the reference's call: a caller holding only a shard of them
            # passes the whole list's norm (ops.dirs_norm of it); default = the norm of `rays`, computed by the entry point
            camera.ipe_dir_norm = _dev(ipe_dir_norm, "ipe_dir_norm").data_ptr() if ipe_dir_norm is not None else None
        if in_kernel_rng:
            camera.rng_seed, camera.rng_ray_offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(rng_ray_offset)
    if n_rays is None:
        N = u_strat.shape[0] if not in_kernel_rng else rays.shape[0]
    else:
        N = n_rays
    need = lib.nerf_amd_render_workspace_bytes(N, n_fine)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    rgb = torch.empty((N, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((N,), dtype=torch.float32, device=dev) if want_depth else None
    w = torch.empty((N, n_fine), dtype=torch.float32, device=dev) if want_weights else None
    check(lib.nerf_amd_render_rays(_ptr(packed_prop), _ptr(packed_mip), _prop_prec(packed_prop, precision, packed_mip), _ptr(rays),
                                   C.byref(camera) if camera is not None else None, ray_offset, _ptr(z_base), _ptr(u_strat),
                                   _ptr(u_inv), N, n_fine, float(near), float(far), int(white_bkg), _ptr(rgb), _ptr(depth),
                                   _ptr(w), _ptr(workspace), _stream()), "nerf_amd_render_rays")
    return rgb, depth, w, workspace


def render_rays_ref(packed_prop, packed_ref, precision, rays, z_base, u_strat, u_inv, n_fine, near, far, white_bkg,
                    want_depth=True, cam_dir: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
                    camera: Optional[Samples] = None, ray_offset: int = 0, n_rays: Optional[int] = None, flags: int = 0,
                    seed: Optional[int] = None, rng_ray_offset: int = 0, contract: bool = False):
    """The tile body of render_image for a Ref-NeRF fine network (procedures.py:64-85, is_ref_model branch) in six launches.
    `cam_dir` (3,) = render_pose[:, -2] asks for the normal image (procedures.py:79-81).  u_strat = u_inv = None with `seed`: in-kernel
    uniforms as in render_rays."""
    in_kernel_rng = u_strat is None
    if in_kernel_rng:
        if u_inv is not None or seed is None:
            raise ValueError("nerf_amd: u_strat and u_inv are both tensors, or both None with a `seed`")
        dev = rays.device if rays is not None else z_base.device
        if camera is None:
            camera = Samples()
        camera.rng_seed, camera.rng_ray_offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(rng_ray_offset)
    else:
        dev = u_strat.device
    if contract:                                             # Mip-NeRF 360 scene contraction of every sample position (the build's own definition)
        if camera is None:
            camera = Samples()
        camera.contract = 1
    N = n_rays if n_rays is not None else (rays.shape[0] if in_kernel_rng else u_strat.shape[0])
    need = lib.nerf_amd_render_ref_workspace_bytes(N, n_fine)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    rgb = torch.empty((N, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((N,), dtype=torch.float32, device=dev) if want_depth else None
    cam_dir = _dev(cam_dir, "cam_dir") if cam_dir is not None else None
    normal_img = torch.empty((N,), dtype=torch.float32, device=dev) if cam_dir is not None else None
    check(lib.nerf_amd_render_rays_ref(_ptr(packed_prop), _ptr(packed_ref), _prop_prec(packed_prop, precision), int(flags), _ptr(rays),
                                       C.byref(camera) if camera is not None else None, ray_offset, _ptr(z_base), _ptr(u_strat),
                                       _ptr(u_inv), N, n_fine, float(near), float(far), int(white_bkg), _ptr(cam_dir), _ptr(rgb),
                                       _ptr(depth), _ptr(normal_img), _ptr(workspace), _stream()), "nerf_amd_render_rays_ref")
    return rgb, depth, normal_img, workspace


# ------------------------------------------------------------------------------------------------ backward
BWD_MAX_SAMPLES = 1024         # the backward kernels keep a ray in 4 / 8 / 16 register chunks of 64 samples (picked by the row length)


def sigma_to_weights_backward(sigma: torch.Tensor, z: torch.Tensor, dirs: Optional[torch.Tensor], act: int, d_weights: torch.Tensor) -> torch.Tensor:
    sigma, z, d_weights = _dev(sigma, "density"), _dev(z, "zvals"), _dev(d_weights, "d_weights")
    dirs = _dev(dirs, "ray_dirs") if dirs is not None else None
    N, S = sigma.shape
    out = torch.empty_like(sigma)
    check(lib.nerf_amd_sigma_to_weights_backward(_ptr(sigma), _ptr(z), _ptr(dirs), N, S, act, _ptr(d_weights), _ptr(out), _stream()),
          "nerf_amd_sigma_to_weights_backward")
    return out


def composite_backward(rgbo: torch.Tensor, z: torch.Tensor, dirs: torch.Tensor, mul_norm: bool, white_bkg: bool, act: int, near_far,
                       d_rgb: Optional[torch.Tensor], d_weights: Optional[torch.Tensor], d_depth: Optional[torch.Tensor],
                       sigma_shift: float = 0.0) -> torch.Tensor:
    rgbo, z = _dev(rgbo, "rgbo"), _dev(z, "depth")
    N, S = rgbo.shape[0], rgbo.shape[1]
    if dirs.shape[-1] == 6:
        dirs = _dev(dirs, "rays")
        dirs_ptr, dirs_stride = C.c_void_p(dirs.data_ptr() + 12), 6
    else:
        dirs = _dev(dirs, "ray_dirs")
        dirs_ptr, dirs_stride = _ptr(dirs), 3
    d_rgb = _dev(d_rgb, "d_rgb") if d_rgb is not None else torch.zeros((N, 3), dtype=torch.float32, device=rgbo.device)
    d_weights = _dev(d_weights, "d_weights") if d_weights is not None else None
    d_depth = _dev(d_depth, "d_depth") if d_depth is not None else None
    near, far = (near_far if near_far is not None else (0.0, 1.0))
    out = torch.empty_like(rgbo)
    flags = (1 if mul_norm else 0) | (2 if white_bkg else 0)
    check(lib.nerf_amd_composite_backward(_ptr(rgbo), _ptr(z), z.shape[-1], dirs_ptr, dirs_stride, N, S, flags, act, float(sigma_shift),
                                          float(near), float(far), _ptr(d_rgb), _ptr(d_weights), _ptr(d_depth), _ptr(out), _stream()),
          "nerf_amd_composite_backward")
    return out


def max_blur_backward(weights: torch.Tensor, d_out: torch.Tensor) -> torch.Tensor:
    weights, d_out = _dev(weights, "weights"), _dev(d_out, "d_out")
    S = weights.shape[-1]
    out = torch.empty_like(weights)
    check(lib.nerf_amd_max_blur_backward(_ptr(weights), _ptr(d_out), weights.numel() // S, S, _ptr(out), _stream()), "nerf_amd_max_blur_backward")
    return out


def get_bounds_backward(below: torch.Tensor, d_bounds: torch.Tensor, n_coarse: int) -> torch.Tensor:
    d_bounds = _dev(d_bounds, "d_bounds")
    below = below.to(torch.int64).contiguous()
    N, K = below.shape
    out = torch.empty((N, n_coarse), dtype=torch.float32, device=d_bounds.device)
    check(lib.nerf_amd_get_bounds_backward(_ptr(below), _ptr(d_bounds), N, n_coarse, K, _ptr(out), _stream()), "nerf_amd_get_bounds_backward")
    return out


# ------------------------------------------------------------------------------------------------ training forward
NET_PROPOSAL, NET_MIP = 0, 1


def _train_forward_samples(net: int, packed: torch.Tensor, precision: int, s: Samples, out_shape, device):
    out = torch.empty(out_shape, dtype=torch.float32, device=device)
    dump = leased(("dump", net), lib.nerf_amd_train_dump_bytes(net, precision, s.M), device)
    if s.M:
        fn = lib.nerf_amd_proposal_forward_train if net == NET_PROPOSAL else lib.nerf_amd_mip_forward_train
        check(fn(_ptr(packed), precision, C.byref(s), _ptr(out), _ptr(dump), _stream()), "nerf_amd_*_forward_train")
    return out, dump


def proposal_forward_train(packed: torch.Tensor, precision: int, pts: torch.Tensor, contract: bool = False):
    """Same result as proposal_forward plus the activation dump the backward needs."""
    pts = _dev(pts, "pts")
    return _train_forward_samples(NET_PROPOSAL, packed, precision, _samples_pts(pts, 3, contract), pts.shape[:-1], pts.device)


def mip_forward_train(packed: torch.Tensor, precision: int, pts: torch.Tensor, contract: bool = False):
    pts = _dev(pts, "pts")
    return _train_forward_samples(NET_MIP, packed, precision, _samples_pts(pts, 6, contract), pts.shape[:-1] + (4,), pts.device)


def mip_forward_train_samples(packed: torch.Tensor, precision: int, s: Samples, shape, device):
    """Training forward on a samples descriptor (rays + depths: positions formed in the kernel; integrated PE and scene contraction are
    flags of the descriptor, samples_rays) -> (rgbo (*shape, 4), dump)."""
    return _train_forward_samples(NET_MIP, packed, precision, s, tuple(shape) + (4,), device)


def proposal_forward_train_samples(packed: torch.Tensor, precision: int, s: Samples, shape, device):
    return _train_forward_samples(NET_PROPOSAL, packed, precision, s, tuple(shape), device)


def train_dump_rows(dump: torch.Tensor, net: int, precision: int, M: int, layer: int, n_features: int) -> torch.Tensor:
    """Layer `layer` of a training dump as a row-major (M, n_features) matrix (bf16 / fp32 like the kernels' activations)."""
    out = torch.empty((M, n_features), dtype=torch.bfloat16 if precision == BF16 else torch.float32, device=dump.device)
    check(lib.nerf_amd_train_dump_to_rows(_ptr(dump), net, precision, M, layer, n_features, _ptr(out), _stream()), "nerf_amd_train_dump_to_rows")
    return out


def train_dump_rows_mask_(dump: torch.Tensor, net: int, precision: int, layer: int, delta: torch.Tensor):
    """train_dump_rows + relu_mask_bias_ of one layer in a single pass: -> (act rows (M, F), delta masked IN PLACE, fp32 column sums)."""
    assert delta.is_contiguous() and delta.dim() == 2 and delta.dtype == (torch.bfloat16 if precision == BF16 else torch.float32)
    M, F = delta.shape
    act = torch.empty_like(delta)
    part = torch.empty((lib.nerf_amd_train_dump_rows_mask_partials(), F), dtype=torch.float32, device=delta.device)
    check(lib.nerf_amd_train_dump_rows_mask(_ptr(dump), net, precision, M, layer, F, _ptr(act), _ptr(delta), _ptr(part), _stream()),
          "nerf_amd_train_dump_rows_mask")
    return act, delta, part.sum(0)


def encode_rows(x: torch.Tensor, L: int, precision: int, normalize: bool = False) -> torch.Tensor:
    """[x | PE_L(x)] zero-padded to a multiple of 8 columns, as bf16 / fp32 rows: the wgrad operand of the first and skip layers.
    x: (M, >=3) float32 with unit inner stride (a column slice of the (M,6) sample matrix is fine)."""
    if not x.is_cuda:
        raise RuntimeError("nerf_amd: 'x' must live on the HIP device (got %s); there is no CPU path" % x.device)
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1 or x.shape[1] < 3:
        x = x.reshape(-1, x.shape[-1]).float().contiguous()
    M = x.shape[0]
    out = torch.empty((M, (3 + 6 * L + 7) // 8 * 8), dtype=torch.bfloat16 if precision == BF16 else torch.float32, device=x.device)
    check(lib.nerf_amd_encode_rows(_ptr(x), max(x.stride(0), 3), M, L, int(normalize), precision, _ptr(out), _stream()), "nerf_amd_encode_rows")
    return out


def relu_mask_(delta: torch.Tensor, act: torch.Tensor, precision: int) -> torch.Tensor:
    """In place: delta = where(act > 0, delta, 0); both contiguous, same shape and dtype (bf16 for BF16, fp32 for F32)."""
    assert delta.is_contiguous() and act.is_contiguous() and delta.shape == act.shape and delta.dtype == act.dtype
    check(lib.nerf_amd_relu_mask(_ptr(delta), _ptr(act), precision, delta.numel(), _stream()), "nerf_amd_relu_mask")
    return delta


def relu_mask_bias_(delta: torch.Tensor, act: torch.Tensor, precision: int):
    """In place delta = where(act > 0, delta, 0) over a (rows, cols) matrix AND its column sums (fp32) = the bias gradient."""
    assert delta.is_contiguous() and act.is_contiguous() and delta.shape == act.shape and delta.dtype == act.dtype and delta.dim() == 2
    n_part = lib.nerf_amd_relu_mask_bias_partials(precision, delta.shape[0], delta.shape[1])
    part = torch.empty((n_part, delta.shape[1]), dtype=torch.float32, device=delta.device)
    check(lib.nerf_amd_relu_mask_bias(_ptr(delta), _ptr(act), precision, delta.shape[0], delta.shape[1], _ptr(part), _stream()), "nerf_amd_relu_mask_bias")
    return delta, part.sum(0)



# ------------------------------------------------------------------------------------------------ MLP backward on the matrix cores
def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def pack_weights_backward(net: int, precision: int, weights: Sequence[torch.Tensor]) -> torch.Tensor:
    """The TRANSPOSED weights of a network in the dgrad-chain kernels' fragment order (same tensor order as pack_weights)."""
    ws = [_raw(w, "weight") for w in weights]
    blob = torch.empty(lib.nerf_amd_packed_backward_bytes(net, precision), dtype=torch.uint8, device=ws[0].device)
    check(lib.nerf_amd_pack_weights_backward(net, precision, _ptr_array(ws), len(ws), _ptr(blob), _stream()), "nerf_amd_pack_weights_backward")
    return blob


def proposal_backward_chain(packed_bwd: torch.Tensor, precision: int, g_density: torch.Tensor, dump: torch.Tensor) -> torch.Tensor:
    """dgrad chain of the proposal network: g_density (M,) + the training forward's activation dump -> delta dump."""
    g = _dev(g_density.reshape(-1), "g_density")
    delta = leased(("delta", NET_PROPOSAL), dump.numel(), dump.device)      # (held by the returned tensor until the products consumed it)
    check(lib.nerf_amd_proposal_backward_chain(_ptr(packed_bwd), precision, _ptr(g), g.numel(), _ptr(dump), _ptr(delta), _stream()),
          "nerf_amd_proposal_backward_chain")
    return delta


def mip_backward_chain(packed_bwd: torch.Tensor, precision: int, g_rgbo: torch.Tensor, rgbo: torch.Tensor, dump: torch.Tensor) -> torch.Tensor:
    g, o = _dev(g_rgbo.reshape(-1, 4), "g_rgbo"), _dev(rgbo.reshape(-1, 4), "rgbo")
    delta = leased(("delta", NET_MIP), dump.numel(), dump.device)
    check(lib.nerf_amd_mip_backward_chain(_ptr(packed_bwd), precision, _ptr(g), _ptr(o), g.shape[0], _ptr(dump), _ptr(delta), _stream()),
          "nerf_amd_mip_backward_chain")
    return delta


def _grad_buffers(shapes, device):
    """One flat fp32 buffer, one view per tensor (the kernels overwrite every element)."""
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    out, off = [], 0
    for s, n in zip(shapes, sizes):
        out.append(flat[off: off + n].view(s))
        off += n
    return out


def _check_sinks(sinks, shapes, what):
    for t, sh in zip(sinks, shapes):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == tuple(sh)):
            raise RuntimeError("nerf_amd: %s gradient sink must be a contiguous fp32 device tensor of shape %s" % (what, tuple(sh)))


PROP_W_SHAPES = [(256, 63), (256, 256), (256, 256), (256, 256), (1, 256)]


def proposal_weight_grads(precision: int, M: int, dump: torch.Tensor, delta: torch.Tensor, out=None):
    """-> ([dW of layers.{0,2,4,6,8}], [db ...]) in the reference's (out, in) layout.  `out` = (weight sinks, bias sinks): existing
    tensors (e.g. views of one persistent flat gradient buffer, nerf_amd.parallel.FlatGradients) the kernels write instead of fresh ones."""
    dev = dump.device
    if out is not None:
        gw, gb = list(out[0]), list(out[1])
        _check_sinks(gw, PROP_W_SHAPES, "proposal weight"); _check_sinks(gb, [(s[0],) for s in PROP_W_SHAPES], "proposal bias")
    else:
        gw = _grad_buffers(PROP_W_SHAPES, dev)
        gb = _grad_buffers([(s[0],) for s in PROP_W_SHAPES], dev)
    ws = scratch(("wgrad", NET_PROPOSAL), lib.nerf_amd_weight_grads_workspace_bytes(NET_PROPOSAL, precision, M), dev)
    check(lib.nerf_amd_proposal_weight_grads(precision, M, _ptr(dump), _ptr(delta), _ptr_array(gw), _ptr_array(gb), _ptr(ws), _stream()),
          "nerf_amd_proposal_weight_grads")
    return gw, gb


def mip_weight_grads(precision: int, M: int, dump: torch.Tensor, delta: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
                     out=None):
    """-> ([dW], [db]) in MipNeRF._linear_layers() order; `weights` / `biases` are needed to un-fold bottle_neck.0 / rgb_layer.0.
    `out` = (weight sinks, bias sinks) as in proposal_weight_grads."""
    dev = dump.device
    w = [_dev(t.detach(), "weight") for t in weights]
    b = [_dev(t.detach(), "bias") for t in biases]
    if out is not None:
        gw, gb = list(out[0]), list(out[1])
        _check_sinks(gw, [t.shape for t in w], "MipNeRF weight"); _check_sinks(gb, [t.shape for t in b], "MipNeRF bias")
    else:
        gw = _grad_buffers([tuple(t.shape) for t in w], dev)
        gb = _grad_buffers([tuple(t.shape) for t in b], dev)
    ws = scratch(("wgrad", NET_MIP), lib.nerf_amd_weight_grads_workspace_bytes(NET_MIP, precision, M), dev)
    check(lib.nerf_amd_mip_weight_grads(precision, M, _ptr(dump), _ptr(delta), _ptr_array(w), _ptr_array(b), _ptr_array(gw), _ptr_array(gb),
                                        _ptr(ws), _stream()), "nerf_amd_mip_weight_grads")
    return gw, gb


# Parameters updated through raw pointers (nerf_amd_adam_step, a replayed hipGraph) never move torch's `_version`: this counter is part
# of every packed-weight cache key (PackedWeightsMixin.packed, generic_path._packed) and is bumped by whatever writes parameters behind
# torch's back, so that an eval-mode module never renders with weights packed before the update.
PARAM_GENERATION = [0]


def parameters_changed() -> None:
    PARAM_GENERATION[0] += 1


def adam_step(params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], exp_avg: Sequence[torch.Tensor], exp_avg_sq: Sequence[torch.Tensor],
              step: torch.Tensor, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, grad_scale: float = 1.0,
              lr_dev: Optional[torch.Tensor] = None) -> None:
    """torch.optim.Adam's update (no weight decay / amsgrad) over all tensors in one launch; `step` = device float, incremented here.
    `lr_dev` (one float64 on the device) overrides `lr` when the kernel runs: a captured graph then follows a learning-rate schedule."""
    if lr_dev is not None and not (lr_dev.is_cuda and lr_dev.dtype == torch.float64 and lr_dev.numel() == 1):
        raise RuntimeError("nerf_amd.adam_step: lr_dev = one float64 on the HIP device")
    n = len(params)
    for t in list(params) + list(grads) + list(exp_avg) + list(exp_avg_sq):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("nerf_amd.adam_step: tensors must be contiguous fp32 on the HIP device")
    numel = (C.c_int64 * n)(*[p.numel() for p in params])
    check(lib.nerf_amd_adam_step(_ptr_array(params), _ptr_array(grads), _ptr_array(exp_avg), _ptr_array(exp_avg_sq), numel, n, _ptr(step),
                                 float(lr), _ptr(lr_dev), float(beta1), float(beta2), float(eps), float(grad_scale), _stream()), "nerf_amd_adam_step")
    parameters_changed()


# ------------------------------------------------------------------------------------------------ Ref-NeRF training / density gradients
def ref_forward_train(packed: torch.Tensor, precision: int, pts: torch.Tensor, noise: Optional[torch.Tensor], flags: int = 0,
                      noise_std: float = 0.0, noise_seed: int = 0, noise_seed_dev: Optional[torch.Tensor] = None, contract: bool = False):
    """RefNeRF.forward in training (ref_model.py:68-106) + the activation dump and the pre-activation head values of the backward.
    pts (..., 6) -> (rgbo (..., 4), normal (..., 3), dump, aux (M, 16)).  The bottle-neck perturbation (ref_model.py:84-85): `noise`
    (..., 128) given as a tensor, or -- noise None and noise_std > 0 -- drawn inside the kernel from Philox keyed by `noise_seed` /
    the device scalar `noise_seed_dev` and the sample index (= ops.philox_normal of the same key)."""
    pts = _dev(pts, "pts")
    rgbo, normal = _ref_out(pts.shape[:-1], pts.device, True)
    s = _samples_pts(pts, 6, contract)
    dump = leased(("dump", NET_REF), lib.nerf_amd_train_dump_bytes(NET_REF, precision, s.M), pts.device)
    aux = torch.empty((s.M, 16), dtype=torch.float32, device=pts.device)
    if s.M:
        if noise is None and noise_std > 0.0:
            check(lib.nerf_amd_ref_forward_train_dump_rng(_ptr(packed), precision, C.byref(s), int(flags), int(noise_seed) & 0xFFFFFFFFFFFFFFFF,
                                                          _ptr(noise_seed_dev), float(noise_std), _ptr(rgbo), _ptr(normal), _ptr(dump), _ptr(aux), _stream()),
                  "nerf_amd_ref_forward_train_dump_rng")
        else:
            noise = _dev(noise, "noise") if noise is not None else None
            check(lib.nerf_amd_ref_forward_train_dump(_ptr(packed), precision, C.byref(s), int(flags), _ptr(noise), _ptr(rgbo), _ptr(normal), _ptr(dump), _ptr(aux),
                                                      _stream()), "nerf_amd_ref_forward_train_dump")
    return rgbo, normal, dump, aux


def philox_normal(n_samples: int, std: float, seed: int = 0, seed_dev: Optional[torch.Tensor] = None, sample_offset: int = 0, device=None) -> torch.Tensor:
    """(n_samples, 128) ~ N(0, std): the bottle-neck perturbation ref_forward_train(noise=None, noise_std=std, ...) draws in place for the
    same key, as a tensor (nerf_amd_philox_normal)."""
    dev = seed_dev.device if seed_dev is not None else (device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    out = torch.empty((int(n_samples), 128), dtype=torch.float32, device=dev)
    check(lib.nerf_amd_philox_normal(_ptr(out), int(n_samples), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(seed_dev), float(std), int(sample_offset), _stream()),
          "nerf_amd_philox_normal")
    return out


def density_grad(net: int, packed_bwd: torch.Tensor, precision: int, dump: torch.Tensor, x: torch.Tensor, scale: Optional[torch.Tensor] = None,
                 contract: bool = False) -> torch.Tensor:
    """d density / d position of every sample (RefNeRF.get_grad before its normalisation), times `scale` (M,) -- proposal network or
    Ref-NeRF's spatial network.  x (M, >= 3) contiguous rows; -> (M, 3)."""
    x = _dev(x, "positions")
    M = x.shape[0]
    out = torch.empty((M, 3), dtype=torch.float32, device=x.device)
    if M == 0:
        return out
    ws = scratch(("density_grad", net), lib.nerf_amd_density_grad_workspace_bytes(net, precision, M), x.device)        # (`contract`: NERF_AMD_CONTRACTED in `net`)
    sc_stride = 0
    if scale is not None:
        if scale.dtype != torch.float32 or not scale.is_cuda or scale.dim() != 1:
            scale = scale.reshape(-1).float().contiguous()
        sc_stride = scale.stride(0)
    check(lib.nerf_amd_density_grad(net | (0x100 if contract else 0), _ptr(packed_bwd), precision, M, _ptr(dump), _ptr(x), x.shape[1], _ptr(scale), sc_stride, _ptr(out), _ptr(ws),
                                    _stream()), "nerf_amd_density_grad")
    return out


REF_GRAD_SHAPES = ([(256, 63)] + [(256, 256)] * 3 + [(256, 319)] + [(256, 256)] * 3 + [(128, 256), (9, 256), (2, 256), (256, 167)] + [(256, 256)] * 3 +
                   [(256, 423)] + [(256, 256)] * 3 + [(3, 256)])


def ref_backward(packed_bwd: torch.Tensor, precision: int, dump: torch.Tensor, aux: torch.Tensor, dirs: torch.Tensor, g_out: torch.Tensor,
                 ide_table: torch.Tensor, flags: int = 0):
    """Every parameter gradient of RefNeRF.forward.  g_out (M,7) = d loss / d [rgb | raw density | predicted normal]; dirs (M,3).
    -> ([dW]*20, [db]*20): 0..7 spatial, 8 bottle_neck, 9 norm_col_tint_head, 10 rho_tau_head, 11..18 directional, 19 spec_rgb_head.0"""
    g_out, dirs, aux = _dev(g_out, "g_out"), _dev(dirs, "dirs"), _dev(aux, "aux")
    M = g_out.shape[0]
    dev = g_out.device
    gw = _grad_buffers(REF_GRAD_SHAPES, dev)
    gb = _grad_buffers([(s[0],) for s in REF_GRAD_SHAPES], dev)
    ws = scratch(("ref_backward",), lib.nerf_amd_ref_backward_workspace_bytes(precision, M), dev)
    check(lib.nerf_amd_ref_backward(_ptr(packed_bwd), precision, int(flags), M, _ptr(dump), _ptr(aux), _ptr(dirs), dirs.shape[1], _ptr(g_out), g_out.shape[1],
                                    _ptr(_dev(ide_table, "ide_table")), _ptr_array(gw), _ptr_array(gb), _ptr(ws), _stream()), "nerf_amd_ref_backward")
    return gw, gb


# ------------------------------------------------------------------------------------------------ generic-shape layer product (ABI 119)
def _view2d(t: torch.Tensor, name: str):
    """-> (tensor, stride 0, stride 1) of a 2-D fp32 device view whose strides the GEMM can walk (one of them 1, or a 1-wide dimension)"""
    if not t.is_cuda:
        raise RuntimeError("nerf_amd: '%s' must live on the HIP device (got %s); there is no CPU path" % (name, t.device))
    if t.dim() != 2 or t.dtype != torch.float32:
        raise RuntimeError("nerf_amd: '%s' must be a 2-D float32 tensor" % name)
    s0, s1 = t.stride()
    if t.shape[1] == 1:
        s1 = 1
    elif t.shape[0] == 1 and s1 != 1:
        s0 = 1
    if s0 != 1 and s1 != 1:
        t = t.contiguous()
        s0, s1 = t.stride()
    return t, int(s0), int(s1)


_GEMM_RELAYOUT_B = __import__("os").environ.get("NERF_AMD_GEMM_RELAYOUT_B", "1") != "0"        # (A/B switch of the re-layout below)


def gemm(precision: int, a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, act: int = 0,
         mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i, j] = act(sum_p a[i, p] b[p, j] + bias[j]) * [mask[i, j] > 0]  (nerf_amd_gemm: one hand-written MFMA GEMM with explicit strides;
    act 0 none / 1 ReLU / 2 sigmoid).  a (M, P), b (P, N): 2-D fp32 device VIEWS (a transposed view is a stride pair, nothing is copied);
    out / mask: (M, N) views with unit column stride.  The layer products of networks larger than the fused kernels' compiled shapes."""
    a, a_si, a_sp = _view2d(a, "a")
    b, b_sp, b_sj = _view2d(b, "b")
    M, P = a.shape
    if b.shape[0] != P:
        raise RuntimeError("nerf_amd.gemm: inner dimensions differ (%s x %s)" % (tuple(a.shape), tuple(b.shape)))
    N = b.shape[1]
    # The kernel stages an operand fastest when its CONTRACTION index is the unit-stride one (16-byte loads, 8-byte LDS stores; the other
    # order goes through 4-byte LDS stores).  The input-gradient form dx = dy W has b = W row-major, i.e. the slow order -- but W is small
    # (a layer's weights) while M is the sample count: re-lay it out once per call (a copy of <= 16 MiB, no arithmetic).
    if _GEMM_RELAYOUT_B and b_sj == 1 and b_sp != 1 and P > 1 and N > 1 and M >= 4096 and P * N <= (1 << 22):
        b = b.t().contiguous().t()
        b_sp, b_sj = 1, int(b.stride(1))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    if tuple(out.shape) != (M, N) or out.dtype != torch.float32 or not out.is_cuda or (N > 1 and out.stride(1) != 1):
        raise RuntimeError("nerf_amd.gemm: out must be a (%d, %d) float32 device view with unit column stride" % (M, N))
    ldm = 0
    if mask is not None:
        if tuple(mask.shape) != (M, N) or mask.dtype != torch.float32 or (N > 1 and mask.stride(1) != 1):
            raise RuntimeError("nerf_amd.gemm: mask must be a (%d, %d) float32 view with unit column stride" % (M, N))
        ldm = max(int(mask.stride(0)), N)
    if bias is not None:
        bias = _dev(bias.reshape(-1), "bias")
    prec = F32 if (precision & 0xff) == F32 else BF16
    wsb = lib.nerf_amd_gemm_workspace_bytes(M, N, P)
    ws = scratch(("gemm", 0), wsb, a.device) if wsb else None
    check(lib.nerf_amd_gemm(prec, M, N, P, _ptr(a), a_si, a_sp, _ptr(b), b_sp, b_sj, _ptr(out), max(int(out.stride(0)), N), _ptr(bias), int(act),
                            _ptr(mask), ldm, _ptr(ws), _stream()), "nerf_amd_gemm")
    return out


def sigmoid_backward(g: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """g * y * (1 - y) over (M, cols) fp32 views with unit column stride (the adjoint of y = sigmoid(.))"""
    M, cols = y.shape
    out = torch.empty((M, cols), dtype=torch.float32, device=y.device)
    check(lib.nerf_amd_sigmoid_backward(_ptr(g), int(g.stride(0)), _ptr(y), int(y.stride(0)), M, cols, _ptr(out), cols, _stream()), "nerf_amd_sigmoid_backward")
    return out


# ------------------------------------------------------------------------------------------------ generic-shape Ref-NeRF stages (ABI 120)
def _rows(t: torch.Tensor, name: str, min_cols: int) -> torch.Tensor:
    """a 2-D fp32 device view with unit column stride (row stride = t.stride(0)); anything else is made contiguous"""
    if not t.is_cuda:
        raise RuntimeError("nerf_amd: '%s' must live on the HIP device (got %s); there is no CPU path" % (name, t.device))
    if t.dim() != 2 or t.dtype != torch.float32 or (t.shape[1] > 1 and t.stride(1) != 1) or t.stride(0) < t.shape[1]:
        t = t.reshape(-1, t.shape[-1]).float().contiguous()
    if t.shape[1] < min_cols:
        raise RuntimeError("nerf_amd: '%s' needs at least %d columns (got %d)" % (name, min_cols, t.shape[1]))
    return t


def ref_dir_inputs(heads: torch.Tensor, dirs: torch.Tensor, ide_level: int, table: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """ref_model.py:80-92 per sample: heads (M,11) [normal | diffuse | tint | roughness | density] raw, dirs (M,3) -> writes
    [IDE real T | IDE imag T | n.d] into the (M, 2T+1) view `out`, returns the predicted normal (M,3)."""
    heads, dirs, out_v = _rows(heads, "heads", 11), _rows(dirs, "dirs", 3), out
    M = heads.shape[0]
    T = (1 << ide_level) - 1 + ide_level
    if out_v.dim() != 2 or out_v.dtype != torch.float32 or tuple(out_v.shape) != (M, 2 * T + 1) or out_v.stride(1) != 1 or not out_v.is_cuda:
        raise RuntimeError("nerf_amd.ref_dir_inputs: out must be a (%d, %d) float32 device view with unit column stride" % (M, 2 * T + 1))
    normal = torch.empty((M, 3), dtype=torch.float32, device=heads.device)
    check(lib.nerf_amd_ref_dir_inputs(_ptr(heads), int(heads.stride(0)), _ptr(dirs), int(dirs.stride(0)), M, int(ide_level), _ptr(_dev(table, "ide_table")),
                                      _ptr(out_v), int(out_v.stride(0)), _ptr(normal), _stream()), "nerf_amd_ref_dir_inputs")
    return normal


def ref_dir_inputs_backward(heads: torch.Tensor, dirs: torch.Tensor, ide_level: int, table: torch.Tensor, d_out: torch.Tensor, g_normal: torch.Tensor,
                            d_heads: torch.Tensor) -> None:
    """adjoint of ref_dir_inputs: writes columns 0-2 (normal) and 9 (roughness) of the (M, >= 11) view `d_heads`"""
    heads, dirs, d_out, g_normal = _rows(heads, "heads", 11), _rows(dirs, "dirs", 3), _rows(d_out, "d_out", 3), _rows(g_normal, "g_normal", 3)
    M = heads.shape[0]
    check(lib.nerf_amd_ref_dir_inputs_backward(_ptr(heads), int(heads.stride(0)), _ptr(dirs), int(dirs.stride(0)), M, int(ide_level), _ptr(_dev(table, "ide_table")),
                                               _ptr(d_out), int(d_out.stride(0)), _ptr(g_normal), int(g_normal.stride(0)), _ptr(d_heads), int(d_heads.stride(0)),
                                               _stream()), "nerf_amd_ref_dir_inputs_backward")


def ref_combine(heads: torch.Tensor, spec: torch.Tensor, flags: int) -> torch.Tensor:
    """ref_model.py:98-105: spec (M,3) = sigmoid(spec_rgb_head) -> rgbo (M,4) = [rgb | raw density]"""
    heads, spec = _rows(heads, "heads", 11), _rows(spec, "spec", 3)
    M = heads.shape[0]
    rgbo = torch.empty((M, 4), dtype=torch.float32, device=heads.device)
    check(lib.nerf_amd_ref_combine(_ptr(heads), int(heads.stride(0)), _ptr(spec), int(spec.stride(0)), M, int(flags), _ptr(rgbo), _stream()), "nerf_amd_ref_combine")
    return rgbo


def ref_combine_backward(g_rgbo: torch.Tensor, heads: torch.Tensor, spec: torch.Tensor, flags: int, d_heads: torch.Tensor) -> torch.Tensor:
    """adjoint of ref_combine: -> d_spec (M,3) w.r.t. spec_rgb_head's pre-activation; writes columns 3-8 and 10 of `d_heads`"""
    g, heads, spec = _rows(g_rgbo, "g_rgbo", 4), _rows(heads, "heads", 11), _rows(spec, "spec", 3)
    M = heads.shape[0]
    d_spec = torch.empty((M, 3), dtype=torch.float32, device=heads.device)
    check(lib.nerf_amd_ref_combine_backward(_ptr(g), int(g.stride(0)), _ptr(heads), int(heads.stride(0)), _ptr(spec), int(spec.stride(0)), M, int(flags),
                                            _ptr(d_spec), 3, _ptr(d_heads), int(d_heads.stride(0)), _stream()), "nerf_amd_ref_combine_backward")
    return d_spec


def positional_encoding_backward(d_enc: torch.Tensor, x: torch.Tensor, L: int, cat_origin: bool) -> torch.Tensor:
    """d_x (M,3) from the gradient w.r.t. [x | sin 2^f x | cos 2^f x] rows (nerf_helper.py:38-48 layout, raw position in front when cat_origin)"""
    d_enc, x = _rows(d_enc, "d_enc", 6 * L + (3 if cat_origin else 0)), _rows(x, "x", 3)
    M = x.shape[0]
    out = torch.empty((M, 3), dtype=torch.float32, device=x.device)
    check(lib.nerf_amd_positional_encoding_backward(_ptr(d_enc), int(d_enc.stride(0)), _ptr(x), int(x.stride(0)), M, int(L), int(bool(cat_origin)), _ptr(out),
                                                    _stream()), "nerf_amd_positional_encoding_backward")
    return out


def contract_positions(x: torch.Tensor, grad: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Mip-NeRF 360 scene contraction of (M,3) positions (grad None), or the pull-back of `grad` (M,3) = a gradient w.r.t. contract(x)
    through the contraction's Jacobian (nerf_amd_contract_positions: the layer-by-layer route's stage; the fused kernels contract in
    their sample fetch)."""
    x = _rows(x, "x", 3)
    M = x.shape[0]
    out = torch.empty((M, 3), dtype=torch.float32, device=x.device)
    g = _rows(grad, "grad", 3) if grad is not None else None
    check(lib.nerf_amd_contract_positions(_ptr(x), int(x.stride(0)), M, _ptr(g), int(g.stride(0)) if g is not None else 0, _ptr(out), _stream()),
          "nerf_amd_contract_positions")
    return out


def add_rows_(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """dst += src over (M, cols) fp32 device views with unit column stride"""
    if tuple(dst.shape) != tuple(src.shape) or dst.dim() != 2 or dst.dtype != torch.float32 or src.dtype != torch.float32 or not dst.is_cuda:
        raise RuntimeError("nerf_amd.add_rows_: (M, cols) float32 device views of one shape")
    M, cols = dst.shape
    if cols > 1 and (dst.stride(1) != 1 or src.stride(1) != 1):
        raise RuntimeError("nerf_amd.add_rows_: unit column stride")
    check(lib.nerf_amd_add_rows(_ptr(dst), int(dst.stride(0)), _ptr(src), int(src.stride(0)), M, cols, _stream()), "nerf_amd_add_rows")
    return dst


# ---------------------------------------------------------------------------------------------------------------- layer products on bf16 rows
def _pad(n: int, m: int) -> int:
    return (int(n) + m - 1) // m * m


def rows_to_bf16(src: torch.Tensor, dst: torch.Tensor, col0: int = 0, fill: Optional[int] = None, rows: Optional[int] = None) -> torch.Tensor:
    """dst[:rows, col0 : col0 + fill] = bf16(src) (RNE), zeros beyond src's rows / columns (nerf_amd_rows_to_bf16): fp32 rows -- an encoding,
    an element-wise stage's output, an nn.Linear.weight -- into a column range of 2-D bf16 rows, with the zero padding nerf_amd_rows_gemm's
    last chunk (and the packed weights' tile rows) rely on.  `fill` defaults to src's width rounded up to 8 (clipped to dst)."""
    src, _, s1 = _view2d(src, "src")
    if s1 != 1:
        src = src.contiguous()
    if dst.dim() != 2 or dst.dtype != torch.bfloat16 or not dst.is_cuda or (dst.shape[1] > 1 and dst.stride(1) != 1):
        raise RuntimeError("nerf_amd.rows_to_bf16: dst must be 2-D bfloat16 device rows with unit column stride")
    R, cols = src.shape
    rows = R if rows is None else int(rows)
    fill = min(_pad(cols, 8), dst.shape[1] - col0) if fill is None else int(fill)
    if rows > dst.shape[0] or R > rows or col0 < 0 or col0 + fill > dst.shape[1] or fill < cols:
        raise RuntimeError("nerf_amd.rows_to_bf16: the column range [%d, %d) x %d rows does not fit dst %s" % (col0, col0 + fill, rows, tuple(dst.shape)))
    check(lib.nerf_amd_rows_to_bf16(_ptr(src), R, int(src.stride(0)), rows, cols, fill, dst.data_ptr() + 2 * col0, int(dst.stride(0)), _stream()),
          "nerf_amd_rows_to_bf16")
    return dst


class PackedLinear:
    """An nn.Linear's parameters in nerf_amd_rows_gemm's layout: weight (N, K) as zero-padded bf16 rows (n_pad % 256 == 0, ldw % 64 == 0), bias
    as n_pad floats.  `columns` = [(first column, count), ...] re-orders the input features (the concatenations of the reference put the
    encoding FIRST, mip_model.py:55 / ref_model.py:76,95; the bf16 rows keep the hidden features first so that every product writes at an
    aligned column 0)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], columns=None):
        w = weight.detach().float()
        if columns is not None:
            w = torch.cat([w[:, c0:c0 + n] for c0, n in columns], dim=1)
        self.N, self.K = int(w.shape[0]), int(w.shape[1])
        self.n_pad, self.ldw = _pad(self.N, 256), _pad(self.K, 64)
        self.weight = torch.empty((self.n_pad, self.ldw), dtype=torch.bfloat16, device=w.device)
        rows_to_bf16(w.contiguous(), self.weight, 0, self.ldw, self.n_pad)
        self.bias = torch.zeros((self.n_pad,), dtype=torch.float32, device=w.device)
        if bias is not None:
            self.bias[:self.N] = bias.detach().float().reshape(-1)


def rows_gemm(x: torch.Tensor, layer: PackedLinear, act: int = 0, out: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """out[m, n] = act(sum_k x[m, k] W[n, k] + bias[n]) (nerf_amd_rows_gemm): x (M, K) bf16 rows -- a view whose row stride is a multiple of 8
    elements, 16-byte aligned, with finite elements up to the next multiple of 8 columns; out: (M, N) bf16 rows for the next layer (N % 4
    == 0) or fp32 rows (heads / element-wise stages), given as a view or allocated."""
    if x.dim() != 2 or x.dtype != torch.bfloat16 or not x.is_cuda or (x.shape[1] > 1 and x.stride(1) != 1):
        raise RuntimeError("nerf_amd.rows_gemm: x must be 2-D bfloat16 device rows with unit column stride")
    M, K = x.shape
    if K != layer.K:
        raise RuntimeError("nerf_amd.rows_gemm: x has %d columns, the packed layer %d" % (K, layer.K))
    if out is None:
        out = torch.empty((M, layer.N if out_dtype == torch.float32 else _pad(layer.N, 8)), dtype=out_dtype, device=x.device)[:, :layer.N]
    if tuple(out.shape) != (M, layer.N) or out.dtype not in (torch.bfloat16, torch.float32) or not out.is_cuda or (layer.N > 1 and out.stride(1) != 1):
        raise RuntimeError("nerf_amd.rows_gemm: out must be a (%d, %d) bfloat16 / float32 device view with unit column stride" % (M, layer.N))
    check(lib.nerf_amd_rows_gemm(M, layer.N, K, x.data_ptr(), int(x.stride(0)) if M > 1 else _pad(K, 8), layer.weight.data_ptr(), layer.ldw, layer.n_pad,
                                 layer.bias.data_ptr(), int(act), out.data_ptr(), max(int(out.stride(0)), layer.N) if M > 1 else _pad(layer.N, 8),
                                 1 if out.dtype == torch.bfloat16 else 0, _stream()), "nerf_amd_rows_gemm")
    return out
