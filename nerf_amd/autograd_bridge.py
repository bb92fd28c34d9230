"""Autograd bridge: makes the HIP ops usable under ``loss.backward()``.

Forward is ALWAYS a HIP kernel, and so is backward: every op carries a hand-written HIP backward -- the
sampling / compositing adjoints, the MLPs' fused dgrad chains and MFMA weight gradients on the training forward's activation dump.
``HipOp.backward`` has exactly two outcomes: the HIP backward ran, or ``NotImplementedError``.  There is no torch re-evaluation of an
op anywhere in this package: the torch expressions the kernels are specified by -- and tested against -- live with the tests
(``tests/torch_spec.py``), together with everything that differentiates them.  Calls the HIP backward does not cover (a compositing
row longer than ``ops.BWD_MAX_SAMPLES``, MipNeRF positions that require a gradient) are refused when the FORWARD is called
(``unsupported``), not deep inside ``loss.backward()``.

Gradient parity with the reference is pinned by golden G14 / G17 (tests/test_gpu_parity.py::test_train_step_gradients).
"""
from typing import Callable

import torch


class _VJP:
    """`inputs_only` while RefNeRF.get_grad asks for d(out)/d(positions): the networks' HIP backward then runs the dgrad-only density
    chain (nerf_amd_density_grad) and forms no parameter gradient."""
    inputs_only = False


# Gradients w.r.t. sample POSITIONS.  The reference's training loss never uses them (the fine depths are detached, utils.py:35-36; the
# position leaves of train.py:165,179 exist for RefNeRF.get_grad only), so by default the HIP backward of ProposalNetwork / RefNeRF forms
# a position gradient ONLY inside get_grad (inputs_only_grad) and `loss.backward()` leaves `pts.grad` untouched -- a dgrad chain and an
# encoding adjoint per network saved per step.  Set POSITION_GRADS = True to have ProposalNetwork's loss backward also return
# d loss / d pts (pose / sample refinement); RefNeRF's full position gradient is not built (it raises when asked for).
POSITION_GRADS = False


class inputs_only_grad:
    """with inputs_only_grad(): torch.autograd.grad(y, positions, ...) -- HipOp backward passes inside differentiate the non-parameter
    inputs only (the density-gradient normals of ref_model.py:119-125 need no parameter gradient)."""

    def __enter__(self):
        self.prev, _VJP.inputs_only = _VJP.inputs_only, True

    def __exit__(self, *exc):
        _VJP.inputs_only = self.prev


def unsupported(what: str):
    """A differentiable call outside the HIP backward's coverage: refuse it at forward time."""
    raise NotImplementedError("nerf_amd: %s has no HIP backward and this package contains no torch fallback -- wrap the call in torch.no_grad() "
                              "if no gradient is needed (the ops' torch specifications, for anyone who wants autograd through them, are tests/torch_spec.py)" % what)


class HipOp(torch.autograd.Function):
    """forward = ``hip_fn(*tensors)`` (HIP kernels); backward = ``hip_bwd(grad, *tensors)`` (HIP kernels) -> one gradient (or None) per
    argument.  The first ``n_diff`` outputs of hip_fn are differentiable -- hip_bwd then receives a tuple of gradients (None for outputs
    the loss does not reach) when n_diff > 1 --; further outputs are returned as-is (non-differentiable)."""

    @staticmethod
    def forward(ctx, hip_fn: Callable, hip_bwd: Callable, n_diff: int, *tensors):
        for t in tensors:
            if isinstance(t, torch.Tensor) and not t.is_cuda:
                raise RuntimeError("nerf_amd: tensors must live on the HIP device")
        ctx.hip_bwd = hip_bwd
        ctx.save_for_backward(*[t for t in tensors if isinstance(t, torch.Tensor)])
        ctx.is_tensor = [isinstance(t, torch.Tensor) for t in tensors]
        ctx.consts = [t for t in tensors if not isinstance(t, torch.Tensor)]
        ctx.n_diff = int(n_diff)
        ctx.set_materialize_grads(False)                      # an output the loss does not reach arrives as None, not as zeros
        with torch.no_grad():
            out = hip_fn(*[t.detach() if isinstance(t, torch.Tensor) else t for t in tensors])
        if isinstance(out, tuple) and len(out) > ctx.n_diff:
            ctx.mark_non_differentiable(*[o for o in out[ctx.n_diff:] if isinstance(o, torch.Tensor)])
        return out

    @staticmethod
    def backward(ctx, *out_grads):
        n_args = len(ctx.is_tensor)
        if all(g is None for g in out_grads[:ctx.n_diff]):
            return (None, None, None, *[None] * n_args)
        if ctx.n_diff == 1:
            grad = out_grads[0].contiguous()
        else:
            grad = tuple(g.contiguous() if g is not None else None for g in out_grads[:ctx.n_diff])
        it_t, it_c = iter(ctx.saved_tensors), iter(ctx.consts)
        full = [next(it_t) if is_t else next(it_c) for is_t in ctx.is_tensor]
        with torch.no_grad():
            grads = ctx.hip_bwd(grad, *full)
        if grads is None:                                       # (the forward-time checks make this unreachable for this package's own ops)
            raise NotImplementedError("nerf_amd: the op's HIP backward declined this call and there is no torch fallback")
        return (None, None, None, *grads)


def needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)
