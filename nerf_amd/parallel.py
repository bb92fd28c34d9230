"""Multi-GPU glue: one process per GPU over torch.distributed (backend "nccl" = RCCL on
ROCm; "gloo" in the CPU tests).

* Rendering shards rays: every rank renders a contiguous, tile-aligned slice of the image's ray list with its own
  replica of both networks (3 MB).  There is NO collective on the data path; one all_gather of the 16 B/ray result at
  the end (or none, for throughput runs).
* Training (what the reference's ddp_train.py does with DistributedDataParallel, ddp_train.py:98): all gradients --
  fine AND proposal network; the reference leaves the proposal net un-reduced (ddp_train.py:97-99), a deviation noted
  in DESIGN.md -- are flattened into ONE buffer and reduced with ONE all_reduce per step: 2.98 MB is latency-bound on
  xGMI, so a single collective beats per-bucket machinery.
"""
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int, align: int = 1) -> Tuple[int, int]:
    """Contiguous [start, end) slice of `n_items` for `rank`; boundaries are multiples of `align` (e.g. the 2500-ray
    render tile) except the last one.  Slices differ by at most one `align` unit."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    units = (n_items + align - 1) // align
    base, rem = divmod(units, world)
    start_u = rank * base + min(rank, rem)
    end_u = start_u + base + (1 if rank < rem else 0)
    return min(start_u * align, n_items), min(end_u * align, n_items)


def gather_shards(local: torch.Tensor, n_items: int, align: int = 1, group=None) -> torch.Tensor:
    """all_gather of ragged per-rank slices (dim 0) back into the full (n_items, ...) tensor on every rank."""
    world = dist.get_world_size(group)
    if local.is_cuda and dist.get_backend(group) == "gloo":        # gloo has no device all_gather: stage through the host
        return gather_shards(local.cpu(), n_items, align, group).to(local.device)
    sizes = [shard_range(n_items, r, world, align) for r in range(world)]
    longest = max(e - s for s, e in sizes)
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[: e - s] for o, (s, e) in zip(out, sizes)], dim=0)


def allreduce_gradients(modules: Sequence[torch.nn.Module], group=None, average: bool = True) -> int:
    """One flat-buffer all_reduce over the gradients of every parameter of `modules` (missing grads count as zero).
    Returns the number of elements reduced."""
    params: List[torch.nn.Parameter] = [p for m in modules for p in m.parameters() if p.requires_grad]
    if not params:
        return 0
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off: off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return off


class FlatGradients:
    """ONE persistent flat fp32 gradient buffer over the parameters of `modules` (fine AND proposal network: 744 069 elements = 2.98 MB);
    every ``p.grad`` is a view of it for the lifetime of the object.

    * the weight-gradient kernels of attached nerf_amd modules write straight into the views (``PackedWeightsMixin.grad_sinks``): the
      backward has no per-tensor accumulate launches and nothing is ever concatenated or copied back;
    * ``all_reduce()`` is ONE collective on the buffer (ddp_train.py:98 reduces per bucket through DDP's hooks; the reference leaves the
      proposal network un-reduced, ddp_train.py:97-99 -- both are reduced here, deviation noted in DESIGN.md).  RCCL: ReduceOp.AVG, one
      launch; gloo (CPU tests / two ranks on one GPU): SUM + one scale.  The call uses no host state, so a training step that contains
      it is captured into a hipGraph like any other (``nerf_amd.training.TrainStep(flat_grads=...)``);
    * ``nerf_amd.optim.Adam`` (or any optimizer) reads the same views.

    ``begin_step()`` opens a new accumulation window (the first backward of a module in a window OVERWRITES its gradients -- no zeroing
    pass --, further ones add); it runs automatically after every ``optimizer.step()`` of an optimizer passed as `optimizer`.  Gradients
    arriving through ordinary autograd (a RefNeRF, a zero-padded narrow network, any other module in `modules`) accumulate into the
    views as usual; those ranges are zeroed by ``begin_step()``.

    The reference loop calls ``opt.zero_grad()`` every iteration (train.py:200), which sets ``p.grad = None``:
      * an attached module's next backward re-binds its views and overwrites them (``sinks_for``);
      * a module on the autograd path then receives FRESH gradient tensors outside the buffer.  ``all_reduce()`` and the optimizer
        pre-step hook therefore close the window with ``finalize_window()``: every ``p.grad`` that is not its view is copied into the view
        and re-bound, a parameter left without a gradient has its range zeroed -- the buffer always holds this window's gradients when the
        collective, the clip or the optimizer read it;
      * an attached module that received NO backward in the window (a loss without the proposal term, a fine-network-only phase) has its
        ranges zeroed by ``finalize_window()`` instead of keeping the previous window's values (torch would skip such a parameter; Adam
        with a zero gradient only decays its moments)."""

    def __init__(self, modules: Sequence[torch.nn.Module], optimizer: Optional[torch.optim.Optimizer] = None, group=None):
        self.modules = list(modules)
        self.group = group
        self.params: List[torch.nn.Parameter] = [p for m in self.modules for p in m.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradients: no trainable parameters")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("FlatGradients: parameters must be fp32 on one device")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self.views, off = {}, 0
        for p in self.params:
            n = p.numel()
            self.views[p] = self.flat[off: off + n].view_as(p)
            off += n
        self._fresh, self._autograd_views, self._autograd_params, self._direct_views = {}, [], [], {}
        for m in self.modules:
            direct = getattr(m, "_supports_grad_sinks", False)
            if direct:
                prev = m.__dict__.get("_grad_owner")
                if prev is not None and prev is not self:
                    prev.detach()                                        # one owner per module: the older buffer would silently go stale
                m.__dict__["_grad_owner"] = self
                direct = m.grad_sinks() is not None                      # (None: a zero-padded narrow network / a frozen parameter -> ordinary autograd path)
                if not direct:
                    m.__dict__.pop("_grad_owner", None)
            if direct:
                self._fresh[m] = True
                self._direct_views[m] = [self.views[p] for p in m.parameters() if p.requires_grad]
            else:                                                        # gradients arrive through autograd's accumulate: zeroed per window
                self._autograd_params += [p for p in m.parameters() if p.requires_grad]
                self._autograd_views += [self.views[p] for p in m.parameters() if p.requires_grad]
        self.bind()
        self.begin_step()
        self._hooks = []
        if optimizer is not None:
            self._hooks.append(optimizer.register_step_pre_hook(lambda *_: self.finalize_window()))
            self._hooks.append(optimizer.register_step_post_hook(lambda *_: self.begin_step()))

    def covers(self, modules) -> bool:
        """True when this buffer is the live gradient owner of exactly these modules"""
        mods = list(modules)
        return len(mods) == len(self.modules) and all(a is b for a, b in zip(sorted(mods, key=id), sorted(self.modules, key=id))) and \
            all(m.__dict__.get("_grad_owner", self) is self for m in mods)

    def bind(self) -> None:
        """(re-)point every p.grad at its view -- e.g. after a zero_grad(set_to_none=True)"""
        self._alive("bind")
        for p in self.params:
            if p.grad is not self.views[p]:
                p.grad = self.views[p]

    def begin_step(self) -> None:
        """open a new accumulation window: views re-bound, autograd-path ranges zeroed, attached modules overwrite on their first backward"""
        self._alive("begin_step")
        self.bind()
        for m in self._fresh:
            self._fresh[m] = True
        if self._autograd_views:
            torch._foreach_zero_(self._autograd_views)

    def finalize_window(self) -> None:
        """make the flat buffer hold THIS window's gradients before anything reads it (collective, clip, optimizer): see the class docstring"""
        self._alive("finalize_window")
        stale = [v for m, fresh in self._fresh.items() if fresh for v in self._direct_views[m]]      # attached, but no backward this window
        for p in self._autograd_params:
            v = self.views[p]
            if p.grad is None:                                           # zero_grad(set_to_none=True) and no gradient since
                stale.append(v)
                p.grad = v
            elif p.grad is not v:                                        # autograd allocated a fresh tensor outside the buffer
                v.copy_(p.grad)
                p.grad = v
        if stale:
            torch._foreach_zero_(stale)
        for m in self._direct_views:                                     # (an un-written module's p.grad may be None as well)
            if self._fresh[m]:
                for p in m.parameters():
                    if p.requires_grad and p.grad is not self.views[p]:
                        p.grad = self.views[p]

    def sinks_for(self, module, layers):
        """(weight views, bias views, overwrite?) of an attached module -- called by its backward; None when a layer's parameter is not in
        the buffer (frozen): the module then takes the ordinary autograd path"""
        if getattr(self, "_dead", False) or any(l.weight not in self.views or l.bias not in self.views for l in layers):
            return None
        first = self._fresh.get(module, False)
        self._fresh[module] = False
        for l in layers:
            for p in (l.weight, l.bias):
                if p.grad is not self.views[p]:                          # zero_grad(set_to_none=True) (or a foreign tensor) since the last
                    p.grad = self.views[p]                               # backward: torch's meaning is "start over", so this backward
                    first = True                                         # overwrites
        return [self.views[l.weight] for l in layers], [self.views[l.bias] for l in layers], first

    def zero_(self) -> None:
        self.flat.zero_()

    def detach(self) -> None:
        """stop owning the modules' gradients (their kernels return gradients to autograd again) and drop the optimizer hooks"""
        for m in self.modules:
            if m.__dict__.get("_grad_owner") is self:
                m.__dict__.pop("_grad_owner", None)
        for h in getattr(self, "_hooks", []):
            h.remove()
        self._hooks = []
        # a detached owner is DEAD: a TrainStep that still holds it would otherwise zero the old views in finalize_window()
        # and re-bind every p.grad to them AFTER the backward wrote into the newer owner's buffer -- all-zero gradients, silently
        self._dead = True
        self._fresh, self._direct_views, self._autograd_params, self._autograd_views = {}, {}, [], []

    def _alive(self, what: str) -> None:
        if getattr(self, "_dead", False):
            raise RuntimeError("nerf_amd.parallel.FlatGradients.%s: this buffer was detached (a newer FlatGradients owns the modules' gradients); "
                               "build the TrainStep / optimizer hooks on the live owner" % what)

    def all_reduce(self, average: bool = True) -> int:
        """One collective over every gradient; returns the element count.  No-op (after closing the window) outside a process group."""
        self.finalize_window()
        if not (dist.is_available() and dist.is_initialized()):
            return self.flat.numel()
        world = dist.get_world_size(self.group)
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG if average else dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            if average and world > 1:
                self.flat.mul_(1.0 / world)
        return self.flat.numel()


def broadcast_parameters(modules: Sequence[torch.nn.Module], src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s weights (one flat broadcast)."""
    params = [p for m in modules for p in m.parameters()]
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(flat[off: off + n].view_as(p))
            off += n


def render_image_sharded(network, prop_net, render_pose, image_size, focal, near, far, sample_num=128, white_bkg=False,
                         render_depth=False, gather: bool = True, seed: int = 0, group=None, render_normal=False, contract: bool = False,
                         ipe=False) -> dict:
    """Ray-sharded whole-image render: rank r renders a contiguous 256-aligned slice of the image's (tile-ordered) ray list through
    ``procedures.render_image``'s own body -- MipNeRF, Ref-NeRF and layer-by-layer networks alike.  Every uniform is drawn in the kernels
    with Philox keyed by (``seed``, GLOBAL ray index), so the gathered image does not depend on the world size: it is bit-identical to
    ``render_image(..., seed=seed)`` in one process (round 4 drew per-rank ``torch.rand`` tensors: 494 MB / N materialised and an image
    that changed with N).  No collective on the data path; one all_gather of rgb (+ depth, + normal) at the end, or none (gather=False:
    ``{"rgb_rays", "depth_rays", "normal_rays", "range"}`` of this rank's slice)."""
    from .procedures import get_patch_size, render_image
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    H, W = (image_size, image_size) if not isinstance(image_size, (tuple, list)) else image_size
    sz, patch_num = get_patch_size((H, W))
    n = H * W if sz is None else patch_num[0] * patch_num[1] * sz * sz               # (rows beyond the last whole tile are not rendered, like the reference)
    start, end = shard_range(n, rank, world, align=256)
    part = render_image(network, prop_net, render_pose, image_size, focal, near, far, sample_num, white_bkg, render_depth, render_normal,
                        rng="philox", contract=contract, ipe=ipe, seed=int(seed), _shard=(start, end))
    if not gather:
        part.pop("to_image")
        return part
    to_image = part["to_image"]
    out = {"rgb": to_image(gather_shards(part["rgb_rays"], n, 256, group), 3)}
    if render_depth:
        out["depth_img"] = to_image(gather_shards(part["depth_rays"].unsqueeze(-1), n, 256, group), 1).expand(3, -1, -1).contiguous()
    if part["normal_rays"] is not None:
        out["normal_img"] = to_image(gather_shards(part["normal_rays"].unsqueeze(-1), n, 256, group), 1).expand(3, -1, -1).contiguous()
    return out
