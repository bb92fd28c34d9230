"""The reference's training iteration (train.py:151-218, MipNeRF branch of ``run()``) as ONE device-resident step.

The reference draws the pixel indices, the depth jitter and the inverse-CDF uniforms on the CPU generator and copies them to the device
every iteration (utils.py:76,89,115), reads the camera pose on the host and rewrites the optimizer's learning rate from Python.  At its
batch size (512-1024 rays) the iteration is launch- and copy-bound, not compute-bound.  Here the training image's pixel table, the camera
pose, the random-number seed, the Adam step count and the learning rate all live in DEVICE memory and every random number is drawn in
kernels (Philox4x32-10), so the iteration never synchronises with the host -- and can therefore be captured once in a hipGraph and
replayed: 0.9 ms instead of 1.7 ms per 512-ray step on one MI355X (bench.py ``train_step.rays_512_hipgraph``).

    step = TrainStep(prop_net, mip_net, optimizer, image_hw=(800, 800), focal=f, near=2., far=6., ray_num=512)
    step.capture()                                   # optional: hipGraph replay from now on
    for img, pose in loader:                         # img (3,H,W), pose (3,4): device tensors (nerf_amd.dataset keeps the scene in HBM)
        lr_sch.update_opt_lr(cnt, optimizer)         # DecayLrScheduler as in train.py:218 -- picked up through the device-side lr
        loss, img_loss = step(img, pose)             # device scalars; read them (``.item()``) only when logging

Semantics are those of train.py:164-199: proposal forward -> softplus -> get_weights -> maxBlurFilter -> inverseSample(sort) ->
MipNeRF forward -> render -> getBounds -> ProposalLoss + MSE -> backward -> Adam; with a RefNeRF as the fine network the
``is_ref_model`` branch (train.py:176-187: coarse/fine merge, density-gradient normals, normal and back-face losses, and with
``prop_normal`` the proposal network's normals, train.py:165-168).  Random streams: the in-kernel
sampler of ``validSampler(rng="philox")`` and ``ops.philox_uniforms`` for inverseSample's ``u``, both keyed by one device-resident
seed that ``nerf_amd_advance_seed`` replaces at the end of every step.
"""
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import ops
from .addtional import ProposalLoss, ProposalNetwork, getBounds
from .mip_methods import maxBlurFilter
from .nerf_base import NeRF
from .optim import Adam
from .utils import _focal_xy, inverseSample, randomFromOneImage


class TrainStep:
    def __init__(self, prop_net, mip_net, optimizer: Adam, image_hw: Tuple[int, int], focal, near: float, far: float, ray_num: int = 512,
                 coarse_pnum: int = 64, fine_pnum: int = 128, crop_xy=(1.0, 1.0), seed: Optional[int] = None, white_bkg: bool = False,
                 prop_normal: bool = False, grad_hook=None, ipe_radius: Optional[float] = None, contract: bool = False, flat_grads=None,
                 grad_clip: float = -0.01):
        """``grad_hook``: called between ``loss.backward()`` and ``optimizer.step()`` -- the place of ddp_train.py's gradient all-reduce
        (``lambda: parallel.allreduce_gradients([mip_net, prop_net])``).  An iteration with a hook runs eagerly (``capture`` refuses).
        ``ipe_radius`` (BASELINE configs[2]): the fine network encodes the conical frusta between consecutive fine depths with the
        integrated PE (mip_methods.py:15-58) instead of the point PE; ``contract`` (configs[4]): Mip-NeRF 360 scene contraction of every
        sample position (proposal and fine).  Neither has a caller in the reference -- the wiring is the build's own (oracle.render_rays
        states it), parity unpinned.
        ``flat_grads`` (default: built here; ``False`` = ordinary per-tensor autograd gradients; or pass
        ``nerf_amd.parallel.FlatGradients([mip_net, prop_net], optimizer, group=...)``): data-parallel training the native way --
        the weight-gradient kernels write into ONE persistent flat buffer, and between backward and the optimizer step ONE all_reduce
        (RCCL) averages it over the ranks.  Unlike a ``grad_hook`` this is part of the captured iteration: ``capture()`` records the
        collective into the hipGraph (backend nccl), so the replayed iteration keeps its launch-free pace on N GPUs.
        ``grad_clip`` (train.py:119-121,217 `--grad_clip`, negative = off like the reference's default): global-norm clipping between
        backward and the optimizer step, evaluated on the device (no host read: capturable)."""
        if not isinstance(optimizer, Adam) or not optimizer.lr_on_device:
            raise ValueError("nerf_amd.training.TrainStep needs nerf_amd.optim.Adam(..., lr_on_device=True): the step must not read host state")
        self.prop_net, self.mip_net, self.opt = prop_net, mip_net, optimizer
        self.near, self.far, self.ray_num, self.coarse_pnum, self.fine_pnum = float(near), float(far), int(ray_num), int(coarse_pnum), int(fine_pnum)
        self.fx, self.fy = _focal_xy(focal)
        self.white_bkg = bool(white_bkg)
        from .ref_model import RefNeRF
        self.is_ref = isinstance(mip_net, RefNeRF)
        self.ipe_radius, self.contract = (None if ipe_radius is None else float(ipe_radius)), bool(contract)
        if self.is_ref and self.ipe_radius is not None:
            raise NotImplementedError("nerf_amd.training.TrainStep: the integrated PE is wired for the MipNeRF branch (the Ref-NeRF kernel encodes points)")
        self.prop_normal = bool(prop_normal) and self.is_ref                              # (train.py: prop_normal only acts with a Ref-NeRF)
        dev = next(mip_net.parameters()).device
        H, W = image_hw
        self.image = torch.zeros((3, H, W), dtype=torch.float32, device=dev)             # static inputs of the (captured) step
        self.pose = torch.zeros((3, 4), dtype=torch.float32, device=dev)
        self.crop_xy = tuple(crop_xy)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())                            # torch.manual_seed governs the whole run
        self.seed = torch.full((1,), seed, dtype=torch.int64, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.img_loss = torch.zeros((), dtype=torch.float32, device=dev)
        if self.is_ref:                                      # the bottle-neck perturbation keyed by this step's device-resident seed
            mip_net.__dict__["noise_seed_dev"] = self.seed   # (RefNeRF.forward, noise_rng "philox": a replayed graph draws fresh noise)
        self.prop_loss_fn = ProposalLoss()
        self.grad_hook = grad_hook
        # an explicitly passed FlatGradients is a request for data-parallel averaging; the default one only holds the gradients (ranks of a
        # model-averaging run, model_average.py, train independently: no implicit collective)
        self._reduce = flat_grads is not None and flat_grads is not False
        if flat_grads is None:
            from .parallel import FlatGradients
            owner = mip_net.__dict__.get("_grad_owner")                                   # a second TrainStep over the same networks (centre crop /
            if owner is not None and prop_net.__dict__.get("_grad_owner") is owner and owner.covers([mip_net, prop_net]):
                flat_grads = owner                                                        # full image) shares the buffer the kernels write into
            else:
                flat_grads = FlatGradients([mip_net, prop_net], optimizer)
        self.flat_grads = flat_grads if flat_grads is not False else None
        if self.flat_grads is not None and getattr(self.flat_grads, "_dead", False):
            raise ValueError("nerf_amd.training.TrainStep: the FlatGradients passed in was detached by a newer owner of these modules")
        self.grad_clip = float(grad_clip)
        self.graph = None

    def release(self) -> None:
        """Take this step's device-resident noise key off the Ref-NeRF module again (planted by the constructor): a module that outlives
        its TrainStep -- moved to another device, trained by hand -- draws its bottle-neck noise key from torch's CPU generator as before."""
        net = getattr(self, "mip_net", None)
        if net is not None and getattr(self, "is_ref", False) and net.__dict__.get("noise_seed_dev") is getattr(self, "seed", None):
            net.__dict__.pop("noise_seed_dev", None)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    # ---------------------------------------------------------------------------------------------------------------- the iteration
    def _body(self):
        pixels, coords = randomFromOneImage(self.image, self.crop_xy)                     # pure indexing on the device (cached table)
        pts, z_c, rgb_tgt, rays = ops.sample_training_rays_dev(pixels, coords, self.pose, self.fx, self.fy, self.near, self.far, self.ray_num,
                                                               self.coarse_pnum, self.seed)            # train.py:160-162
        dirs = rays[:, 3:]
        if self.prop_normal:
            pts.requires_grad_(True)                                                                    # train.py:165
        density = self.prop_net.forward(pts, contract=True) if self.contract else self.prop_net.forward(pts)
        if self.prop_normal:
            from .ref_model import RefNeRF
            coarse_grad = -RefNeRF.get_grad(density, pts)                                               # :167-168
        density = F.softplus(density)                                                                   # :169
        prop_w = maxBlurFilter(ProposalNetwork.get_weights(density, z_c, dirs), 0.01)                   # :170-171
        u = ops.philox_uniforms((self.ray_num, self.fine_pnum + 1), seed_dev=self.seed)
        z_f, below = inverseSample(prop_w, z_c, self.fine_pnum + 1, sort=True, u=u)                     # :174
        extra = 0.0
        if self.is_ref:                                                                                 # :175-187
            from .ref_model import BackFaceLoss, RefNeRF, WeightedNormalLoss
            samples, z_f, below, sort_ids = NeRF.coarseFineMerge(rays, z_c, z_f, below)
            pos, fine_dir = samples.split((3, 3), dim=-1)                                               # (views: RefNeRF.forward reads `samples` itself)
            pos.requires_grad_(True)
            rgbo, pred_normal = self.mip_net.forward(pos, fine_dir, contract=True) if self.contract else self.mip_net.forward(pos, fine_dir)
            density_grad = -RefNeRF.get_grad(rgbo[..., -1], pos)
            rgbo[..., -1] = F.softplus(rgbo[..., -1] + 0.5)
            # train.py:182 passes mip_net.density_act POSITIONALLY, i.e. into `mul_norm`: the depths are not scaled by |d| and the
            # density activation stays the default ReLU (a no-op after the softplus) -- reproduced, like the oracle's ref_train_step
            rendered, weights, _ = NeRF.render(rgbo, z_f, dirs, self.mip_net.density_act, white_bkg=self.white_bkg)
            extra = 4e-4 * WeightedNormalLoss()(weights, density_grad, pred_normal) + 0.1 * BackFaceLoss()(weights, pred_normal, fine_dir)
            if self.prop_normal:
                picked = RefNeRF.coarse_grad_select(density_grad, sort_ids, self.coarse_pnum)
                extra = extra + 4e-5 * WeightedNormalLoss()(prop_w, picked.detach(), coarse_grad)       # 4e-4 * 0.1 (:198)
        else:
            if self.ipe_radius is not None:                      # the fine_pnum frusta between the fine_pnum + 1 sorted depths
                rgbo = self.mip_net.forward_rays(rays, z_f, self.fine_pnum, ipe_radius=self.ipe_radius, contract=self.contract)
                z_f = z_f[..., :-1].contiguous()
            else:
                z_f = z_f[..., :-1].contiguous()                                                        # :188
                rgbo = (self.mip_net.forward_rays(rays, z_f, self.fine_pnum, contract=True) if self.contract
                        else self.mip_net.forward(NeRF.length2pts(rays, z_f)))                          # :189-190
            rendered, weights, _ = NeRF.render(rgbo, z_f, dirs, white_bkg=self.white_bkg)               # :191
        bounds = getBounds(prop_w, below)                                                               # :192
        if self.flat_grads is not None:
            self.flat_grads.bind(); self.flat_grads.begin_step()                                        # (the kernels overwrite: no zeroing pass)
        else:
            self.opt.zero_grad(set_to_none=True)
        img_loss = torch.mean((rendered - rgb_tgt) ** 2)                                                # :194 (nn.MSELoss)
        loss = self.prop_loss_fn(bounds, weights.detach()) + img_loss + extra                           # :196-198
        loss.backward()
        if self._reduce:
            self.flat_grads.all_reduce()                                                                # ddp_train.py:98, as one collective
        if self.grad_hook is not None:
            self.grad_hook()
        if self.grad_clip > 0.0:                                                                        # train.py:217 grad_clip_func
            if self.flat_grads is not None:                                                             # one norm + one scale over the flat buffer
                self.flat_grads.finalize_window()
                flat = self.flat_grads.flat
                flat.mul_(torch.clamp(self.grad_clip / (torch.linalg.vector_norm(flat) + 1e-6), max=1.0))
            else:
                torch.nn.utils.clip_grad_norm_(list(self.mip_net.parameters()) + list(self.prop_net.parameters()), self.grad_clip)
        self.opt.step()
        ops.advance_seed(self.seed)
        self.loss.copy_(loss.detach())
        self.img_loss.copy_(img_loss.detach())

    # ---------------------------------------------------------------------------------------------------------------- driving it
    def set_image(self, img: torch.Tensor, pose: torch.Tensor) -> None:
        """img (3,H,W) / (1,3,H,W), pose (3,4) / (1,3,4) -- device tensors; asynchronous device-to-device copies into the step's inputs"""
        self.image.copy_(img.reshape(self.image.shape), non_blocking=True)
        self.pose.copy_(pose.reshape(-1)[:12].reshape(3, 4), non_blocking=True)

    def set_crop(self, crop_xy) -> None:
        """train.py:155 switches from the centre crop to the full image after `center_crop_iter` iterations: the pixel table changes
        shape, so a captured graph is dropped (call capture() again; the eager path needs nothing)."""
        crop_xy = tuple(crop_xy)
        if crop_xy != self.crop_xy:
            self.crop_xy = crop_xy
            self.graph = None

    def capture(self, warmup: int = 2) -> None:
        """Run `warmup` eager iterations on the current image (lazy kernel attributes, optimizer state, allocator pools), then record
        the iteration into a hipGraph.  The warm-up iterations are real training steps."""
        if self.grad_hook is not None:
            raise RuntimeError("nerf_amd.training.TrainStep: an iteration with a grad_hook (collective) is not captured; run it eagerly")
        self.prop_net.train(); self.mip_net.train()
        for _ in range(max(1, warmup)):
            self._body()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        kw = {}
        if self._reduce and torch.distributed.is_available() and torch.distributed.is_initialized():
            if torch.distributed.get_backend(self.flat_grads.group) != "nccl":
                raise RuntimeError("nerf_amd.training.TrainStep: only an RCCL (backend 'nccl') all-reduce can be captured into the hipGraph")
            kw["capture_error_mode"] = "thread_local"                                     # (the process group's watchdog thread polls events meanwhile)
        with torch.cuda.graph(self.graph, **kw):
            self._body()                                                                  # (recorded, not executed)

    def __call__(self, img: Optional[torch.Tensor] = None, pose: Optional[torch.Tensor] = None):
        if img is not None:
            self.set_image(img, pose)
        if self.graph is not None:
            self.opt.sync_lr()                                                            # a scheduler may have rewritten param_groups' lr
            self.graph.replay()
            ops.parameters_changed()                                                      # (the captured Adam launch ran: packed caches are stale)
        else:
            self._body()
        return self.loss, self.img_loss
