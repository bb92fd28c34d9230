"""Parameter communication of the model-averaging regime -- mirror of the reference's ``nerf/param_com.py`` (same names, arguments
and resulting parameter values; model_average.py:230-260 drives them).

The reference issues one collective / point-to-point call PER PARAMETER TENSOR (22 for MipNeRF, 10 for the proposal net); every call
here flattens the module's parameters into ONE contiguous buffer first, so a model is one message: on xGMI (point-to-point links,
latency-bound at these sizes: 2.1 MB + 0.86 MB) that is 22x / 10x fewer RCCL launches for the same bytes.  The arithmetic per element
is the reference's, in the reference's order.
"""
from typing import List

import torch
from torch import distributed as dist


def _flat(model: torch.nn.Module) -> torch.Tensor:
    return torch.cat([p.data.reshape(-1) for p in model.parameters()])


def _host_staged(flat: torch.Tensor, group) -> bool:
    """RCCL (backend "nccl": what ddp_train.py / model_average.py initialise) moves device buffers directly.  gloo -- the CPU tests and
    the two-ranks-on-one-GPU tests -- has no device send / recv / reduce, so there a device buffer is staged through the host."""
    return flat.is_cuda and dist.get_backend(group) == "gloo"


def _unflat(model: torch.nn.Module, flat: torch.Tensor) -> None:
    off = 0
    for p in model.parameters():
        n = p.numel()
        p.data.copy_(flat[off: off + n].view_as(p))
        off += n
    _invalidate(model)


def _invalidate(model: torch.nn.Module) -> None:
    """parameters were written through .data: drop the packed-weight caches of the HIP kernels"""
    if hasattr(model, "invalidate_packed"):
        model.invalidate_packed()


def param_send(model, dist_ranks: list, group=None):
    """Send the parameters to specific ranks (param_com.py:13-17)."""
    flat = _flat(model)
    if _host_staged(flat, group):
        flat = flat.cpu()
    for rank in dist_ranks:
        dist.send(tensor=flat, dst=rank, group=group)


def param_recv(model, source_rank, group=None):
    """Receive the parameters from one rank (param_com.py:19-22)."""
    flat = _flat(model)
    if _host_staged(flat, group):
        host = flat.cpu()
        dist.recv(tensor=host, src=source_rank, group=group)
        flat.copy_(host)
    else:
        dist.recv(tensor=flat, src=source_rank, group=group)
    _unflat(model, flat)


def param_recv_avg(model, tmp, weights: list, source_ranks: list, self_rank: int = 0, group=None):
    """Receive parameters and form the weighted average (param_com.py:24-34): p = w[self] p + sum_src w[src] p_src; `tmp` (a module of
    the same shape) is left holding the last received model, like in the reference."""
    acc = _flat(model)
    acc *= weights[self_rank]
    buf = _flat(tmp)
    host = buf.cpu() if _host_staged(buf, group) else None
    for src in source_ranks:
        if host is not None:
            dist.recv(tensor=host, src=src, group=group)
            buf.copy_(host)
        else:
            dist.recv(tensor=buf, src=src, group=group)
        acc += weights[src] * buf
    _unflat(tmp, buf)
    _unflat(model, acc)


def param_reduce(model, weights: list, self_rank: int, dst_rank: int = 0, group=None):
    """Weight the parameters and reduce them onto `dst_rank` (param_com.py:36-42).  Every rank's parameters end up multiplied by its
    weight; `dst_rank` additionally holds the sum."""
    flat = _flat(model)
    flat *= weights[self_rank]
    if dist.get_backend(group) == "gloo":
        # gloo uses a non-root rank's buffer as scratch (its contents after the call are unspecified); RCCL leaves it alone, which is what
        # the reference's callers see -- so reduce a copy and take the result on the destination only
        buf = flat.cpu() if flat.is_cuda else flat.clone()
        dist.reduce(tensor=buf, dst=dst_rank, group=group)
        if dist.get_rank(group) == dst_rank:
            flat.copy_(buf)
    else:
        dist.reduce(tensor=flat, dst=dst_rank, group=group)
    _unflat(model, flat)


def param_broadcast(model, src_rank: int = 0, group=None):
    """Broadcast the parameters of `src_rank` (param_com.py:44-47)."""
    flat = _flat(model)
    dist.broadcast(tensor=flat, src=src_rank, group=group)
    _unflat(model, flat)


def param_all_reduce(model, group=None):
    """One-step model average: all-reduce of the (already weighted) parameters (param_com.py:49-54)."""
    flat = _flat(model)
    dist.all_reduce(tensor=flat, group=group)
    _unflat(model, flat)
