"""Adam as ONE HIP launch over all parameters (nerf_amd_adam_step), with torch.optim.Adam's update rule, state layout and
``state_dict`` format (train.py:117-118 builds ``optim.Adam`` over both networks; train.py:200-218 steps it and
nerf_base.DecayLrScheduler rewrites ``param_groups[i]['lr']`` every iteration -- both work on this class unchanged, and a checkpoint's
``'optimizer'`` entry written by either implementation loads into the other).

torch's own foreach implementation is ~10 launches per step over the 32 parameter tensors; here the step counter lives on the device,
so a training step captured in a hipGraph replays correctly; with ``lr_on_device=True`` the learning rate is read from a device scalar
as well (``set_lr`` / the next eager ``step()`` refresh it from ``param_groups[i]['lr']``), so a replayed graph follows
DecayLrScheduler; otherwise it is a launch argument.
"""
from typing import Iterable, Optional

import torch

from . import ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False, grad_scale: float = 1.0, lr_on_device: bool = False):
        if weight_decay != 0.0 or amsgrad:
            raise NotImplementedError("nerf_amd.optim.Adam: weight_decay / amsgrad are not built (the reference uses neither)")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                        differentiable=False, fused=None)
        super().__init__(params, defaults)
        self.grad_scale = float(grad_scale)
        self.lr_on_device = bool(lr_on_device)

    def sync_lr(self):
        """Copy every group's host-side 'lr' into its device scalar (lr_on_device): call it before replaying a captured step whenever a
        scheduler has rewritten param_groups[i]['lr'] (one tiny async fill per group, only when the value changed)."""
        for group in self.param_groups:
            dev_lr = group.get("_lr_dev")
            if dev_lr is not None and group.get("_lr_host") != group["lr"]:
                dev_lr.fill_(float(group["lr"]))
                group["_lr_host"] = group["lr"]

    def _group_state(self, group):
        params = [p for p in group["params"] if p.grad is not None]
        if not params:
            return None
        dev = params[0].device
        step = group.get("_step_dev")
        for p in params:
            st = self.state[p]
            if len(st) == 0:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if step is None:
                # one device counter per group, seeded from the LARGEST per-parameter step of a loaded state (torch.optim.Adam keeps one per
                # parameter; they agree whenever all parameters stepped together, which is the only regime the one-launch kernel supports:
                # a parameter that first receives a gradient later inherits the group's count)
                prevs = [float(self.state[q]["step"]) for q in group["params"] if "step" in self.state[q]]
                step = torch.full((1,), max(prevs) if prevs else 0.0, dtype=torch.float32, device=dev)
                group["_step_dev"] = step
        for p in params:
            self.state[p]["step"] = step                                 # one shared device counter (same value for every tensor)
        return params, step

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            gs = self._group_state(group)
            if gs is None:
                continue
            params, step = gs
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in params]
            b1, b2 = group["betas"]
            lr_dev = None
            if self.lr_on_device:
                if group.get("_lr_dev") is None:
                    group["_lr_dev"] = torch.full((1,), float(group["lr"]), dtype=torch.float64, device=params[0].device)
                    group["_lr_host"] = group["lr"]
                elif not torch.cuda.is_current_stream_capturing():
                    self.sync_lr()
                lr_dev = group["_lr_dev"]
            ops.adam_step([p.data for p in params], grads, [self.state[p]["exp_avg"] for p in params],
                          [self.state[p]["exp_avg_sq"] for p in params], step, group["lr"], b1, b2, group["eps"], self.grad_scale, lr_dev=lr_dev)
            # the kernel wrote the parameters through raw pointers: bump their version counters, which the packed-weight caches
            # (nerf_amd/_packed.py) and autograd's saved-tensor checks are keyed on
            torch._C._autograd._unsafe_set_version_counter(tuple(params), tuple(p._version + 1 for p in params))
        return loss

    def state_dict(self):
        """torch.optim.Adam's format: every parameter's state carries its own 0-d CPU `step` tensor (the live state shares ONE device
        counter between all tensors; written out as-is, torch's foreach step would increment that shared tensor once per parameter)."""
        sd = super().state_dict()
        for g in sd["param_groups"]:
            for k in ("_step_dev", "_lr_dev", "_lr_host"):
                g.pop(k, None)
        sd["state"] = {k: dict(v) for k, v in sd["state"].items()}
        for st in sd["state"].values():
            if "step" in st:
                st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            for k in ("_step_dev", "_lr_dev", "_lr_host"):
                g.pop(k, None)
