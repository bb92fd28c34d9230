"""Autograd bridge: makes the forward-only HIP ops usable under ``loss.backward()``.

Forward is ALWAYS the HIP kernel.  Backward: ops that have a HIP backward (`with_hip_backward`: the
sampling/compositing kernels, the MLPs' GEMM-chain backward on the training forward's activation dump) use it; the torch
expressions below are their specification -- what the tests compare them with -- and the remaining path: gradients w.r.t. sample
positions are obtained by re-evaluating the expression ON THE DEVICE under ``enable_grad`` and asking torch.autograd for the
vector-Jacobian product.  That is not a fallback of the forward path: it never runs unless ``backward`` is called, never touches
the CPU, and raises if the tensors are not on the HIP device.

Gradient parity with the reference is pinned by golden G14 (tests/test_gpu_parity.py::test_train_step_gradients).
"""
from typing import Callable, Sequence

import math

import torch
import torch.nn.functional as F

from . import ops


def _pe(x: torch.Tensor, L: int) -> torch.Tensor:
    """[sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] (nerf_helper.py:38-48) in five device ops instead of 4L + 1."""
    freq = torch.pow(2.0, torch.arange(L, dtype=x.dtype, device=x.device))
    a = x.unsqueeze(-2) * freq[:, None]                              # (..., L, 3)
    return torch.stack((torch.sin(a), torch.cos(a)), dim=-2).reshape(x.shape[:-1] + (6 * L,))


class _VJP:
    """State of the device-side VJP (HipOp.backward): `active` while an expression is re-evaluated for its vector-Jacobian
    product (the Linear layers then use _Linear's backward), `inputs_only` while RefNeRF.get_grad asks for d(out)/d(positions)
    (parameter gradients are then not formed at all)."""
    active = False
    inputs_only = False
    bf16 = False          # the op ran in BF16 precision: the re-evaluated Linear layers use bf16 operands with fp32 accumulation too


# The torch VJP of an op's specification expression (re-evaluated on the device with library GEMMs) is NOT a product path: every shape the
# reference's training loops produce has a hand-written HIP backward.  An op that reaches the VJP -- a compositing row longer than
# ops.BWD_MAX_SAMPLES, MipNeRF positions that require a gradient, an empty batch -- raises unless this switch is on (tests that
# differentiate the specification itself, and callers who knowingly want the slow generic path, set it).
TORCH_VJP_FALLBACK = False


class allow_torch_vjp:
    """with allow_torch_vjp(): ... -- HipOp backward passes inside may fall back to the torch VJP of the specification expression"""

    def __enter__(self):
        global TORCH_VJP_FALLBACK
        self.prev, TORCH_VJP_FALLBACK = TORCH_VJP_FALLBACK, True

    def __exit__(self, *exc):
        global TORCH_VJP_FALLBACK
        TORCH_VJP_FALLBACK = self.prev


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


_WG_SPLIT, _WG_SLICES = 4096, 128


class _Linear(torch.autograd.Function):
    """F.linear for the VJP pass.  torch's own backward hands delta^T @ x -- a (O x M) @ (M x I) product with millions of rows as the
    reduction dimension -- to a library kernel that runs on a few dozen workgroups, and sums the bias gradient with a
    one-row-per-thread reduction; here the reduction dimension is split into ~128 slices (one batched GEMM, partial products summed)
    and the bias gradient is a full-width reduction.  First order only (like every use in this package)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.k = x.shape[-1]
        if _VJP.bf16 and w.shape[0] >= 16:                    # the kernels' arithmetic: bf16 operands, fp32 accumulate, fp32 elementwise
            x, w = x.to(torch.bfloat16), w.to(torch.bfloat16)   # (narrow heads stay fp32: the library has no good bf16 kernels for them)
            if ctx.k % 8:                                     # ... nor for odd leading dimensions: zero-pad the reduction dimension
                x, w = F.pad(x, (0, 8 - ctx.k % 8)), F.pad(w, (0, 8 - ctx.k % 8))
            ctx.save_for_backward(x, w)
            return torch.mm(x.reshape(-1, x.shape[-1]), w.t(), out_dtype=torch.float32).reshape(x.shape[:-1] + (w.shape[0],)) + b
        x = x.to(w.dtype)                                     # (a bf16 activation of _LinearReLU feeding a narrow fp32 head)
        ctx.save_for_backward(x, w)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        low = w.dtype == torch.bfloat16
        f32 = {"out_dtype": torch.float32} if low else {}
        g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
        gx = gw = gb = None
        if ctx.needs_input_grad[2] and not _VJP.inputs_only:
            M, O = g2.shape
            k = 256 // O if O <= 256 and 256 % O == 0 else 1          # narrow heads: fold k rows into one 256-wide row
            main = M // k * k
            gb = g2[:main].reshape(-1, k * O).sum(0).view(k, O).sum(0)
            if main < M:
                gb = gb + g2[main:].sum(0)
        if low:
            g2 = g2.to(torch.bfloat16)
        if ctx.needs_input_grad[0]:
            gx = torch.mm(g2, w, **f32).reshape(x.shape)
        if ctx.needs_input_grad[1] and not _VJP.inputs_only:
            M = g2.shape[0]
            split = max(_WG_SPLIT, M // _WG_SLICES // _WG_SPLIT * _WG_SPLIT)
            n = M // split
            if n < 2:
                gw = torch.mm(g2.t(), x2, **f32)
            else:
                main = n * split
                gw = torch.bmm(g2[:main].view(n, split, -1).transpose(1, 2), x2[:main].view(n, split, -1), **f32).sum(0)
                if main < M:
                    gw = gw + torch.mm(g2[main:].t(), x2[main:], **f32)
        if gx is not None and gx.shape[-1] != ctx.k:
            gx = gx[..., :ctx.k]
        if gw is not None and gw.shape[-1] != ctx.k:
            gw = gw[:, :ctx.k]
        return gx, gw, gb


class _LinearReLU(torch.autograd.Function):
    """relu(F.linear) of a hidden layer for the VJP pass, in three device passes instead of ~ten: bias inside the GEMM, ReLU in place;
    backward = the HIP mask + bias-gradient kernel (nerf_amd_relu_mask_bias) on the incoming gradient, dgrad GEMM, split-K wgrad.  In
    BF16 mode activations stay bf16 from layer to layer like in the forward kernels."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.k, ctx.xshape, ctx.xdtype = x.shape[-1], x.shape, x.dtype
        low = _VJP.bf16
        dt = torch.bfloat16 if low else torch.float32
        x2, w2 = x.reshape(-1, ctx.k).to(dt), w.to(dt)
        if low and ctx.k % 8:                                  # the library's bf16 kernels want aligned leading dimensions
            x2, w2 = F.pad(x2, (0, 8 - ctx.k % 8)), F.pad(w2, (0, 8 - ctx.k % 8))
        y = torch.addmm(b.to(dt), x2, w2.t()).relu_()
        ctx.save_for_backward(x2, w2, y)
        ctx.precision = ops.BF16 if low else ops.F32
        return y.reshape(x.shape[:-1] + (w.shape[0],))

    @staticmethod
    def backward(ctx, g):
        x2, w2, y = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1]).to(y.dtype).contiguous()
        if g2.data_ptr() == g.data_ptr():                      # never write into autograd's gradient buffer
            g2 = g2.clone()
        g2, gb = ops.relu_mask_bias_(g2, y, ctx.precision)
        f32 = {"out_dtype": torch.float32} if y.dtype == torch.bfloat16 else {}
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.mm(g2, w2, **({} if ctx.xdtype == y.dtype else f32))[:, :ctx.k].reshape(ctx.xshape)
        if ctx.needs_input_grad[1] and not _VJP.inputs_only:
            M = g2.shape[0]
            split = max(_WG_SPLIT, M // _WG_SLICES // _WG_SPLIT * _WG_SPLIT)
            n = M // split
            if n < 2:
                gw = torch.mm(g2.t(), x2, **f32)
            else:
                main = n * split
                gw = torch.bmm(g2[:main].view(n, split, -1).transpose(1, 2), x2[:main].view(n, split, -1), **f32).sum(0)
                if main < M:
                    gw = gw + torch.mm(g2[main:].t(), x2[main:], **f32)
            gw = gw[:, :ctx.k]
        return gx, gw, (gb if ctx.needs_input_grad[2] and not _VJP.inputs_only else None)


def _lin(x, w, b):
    return _Linear.apply(x, w, b) if _VJP.active else F.linear(x, w, b)


def _lin_relu(x, w, b):
    """relu(linear): hidden layers.  Outside the VJP pass this is the plain torch expression (the specification the tests use)."""
    width_ok = w.shape[0] in (128, 256)                        # what nerf_amd_relu_mask_bias takes
    if _VJP.active and _VJP.bf16 and width_ok:                 # (fp32 mode: measured no faster than the separate ops)
        return _LinearReLU.apply(x, w, b)
    return F.relu(_lin(x, w, b))


def contract_expr(pts):
    """Mip-NeRF 360 scene contraction (eq. 10) of the position columns of (..., 3) / (..., 6) samples -- the kernels' `contract` flag."""
    x = pts[..., :3]
    n = x.norm(dim=-1, keepdim=True).clamp(min=1.0)                   # inside the unit ball: factor (2 - 1) / 1 = 1
    xc = x * ((2.0 - 1.0 / n) / n)
    return torch.cat((xc, pts[..., 3:]), dim=-1) if pts.shape[-1] > 3 else xc


def proposal_expr(pts, w, b):
    """ProposalNetwork.forward as torch ops (addtional.py:88-96); w, b = lists in state_dict order."""
    h = torch.cat((pts, _pe(pts, 10)), dim=-1)
    for i in range(4):
        h = _lin_relu(h, w[i], b[i])
    return _lin(h, w[4], b[4]).squeeze(-1)


def mip_expr(pts, w, b):
    """MipNeRF.forward as torch ops (mip_model.py:41-60); tensors in the order of MipNeRF._linear_layers()."""
    x, d = pts[..., :3], pts[..., 3:6]
    d = d / d.norm(dim=-1, keepdim=True)
    ex = torch.cat((x, _pe(x, 10)), dim=-1)
    ed = torch.cat((d, _pe(d, 4)), dim=-1)
    h = ex
    for i in range(4):
        h = _lin_relu(h, w[i], b[i])
    g = torch.cat((ex, h), dim=-1)
    for i in range(4, 7):
        g = _lin_relu(g, w[i], b[i])
    bott = _lin(g, w[7], b[7])
    sigma = _lin(g, w[8], b[8])
    c = _lin_relu(torch.cat((bott, ed), dim=-1), w[9], b[9])
    rgb = torch.sigmoid(_lin(c, w[10], b[10]))
    return torch.cat((rgb, sigma), dim=-1)


def ref_expr(pos, d, noise, P, ide_fn, use_srgb: bool = False):
    """RefNeRF.forward as torch ops (ref_model.py:68-106); P = {state_dict key: tensor}; `noise` = the train-mode
    perturbation of the bottle-neck vector or None.  Returns cat(rgb, density, normal) (..., 7)."""
    lin = lambda name, t: _lin(t, P[name + ".weight"], P[name + ".bias"])
    lin_relu = lambda name, t: _lin_relu(t, P[name + ".weight"], P[name + ".bias"])
    ex = torch.cat((pos, _pe(pos, 10)), dim=-1)
    h = ex
    for i in (0, 2, 4, 6):
        h = lin_relu("spa_block1.%d" % i, h)
    g = torch.cat((ex, h), dim=-1)
    for i in (0, 2, 4, 6):
        g = lin_relu("spa_block2.%d" % i, g)
    normal, diffuse, tint = lin("norm_col_tint_head", g).split((3, 3, 3), dim=-1)
    rough, density = lin("rho_tau_head", g).split((1, 1), dim=-1)
    rough = F.softplus(rough - 1.0)
    b = lin("bottle_neck", g)
    if noise is not None:
        b = b + noise
    normal = -normal / (normal.norm(dim=-1, keepdim=True) + 1e-7)
    refl = d - 2.0 * torch.sum(d * normal, dim=-1, keepdim=True) * normal
    allin = torch.cat((b, ide_fn(refl, rough), torch.sum(normal * d, dim=-1, keepdim=True)), dim=-1)
    r = allin
    for i in (0, 2, 4, 6):
        r = lin_relu("dir_block1.%d" % i, r)
    r = torch.cat((allin, r), dim=-1)
    for i in (0, 2, 4, 6):
        r = lin_relu("dir_block2.%d" % i, r)
    spec = torch.sigmoid(lin("spec_rgb_head.0", r)) * torch.sigmoid(tint)
    if use_srgb:                                                                   # ref_model.py:100-102
        from .nerf_helper import linear_to_srgb
        rgb = linear_to_srgb(spec + torch.sigmoid(diffuse - math.log(3.0)))
    else:
        rgb = spec + torch.sigmoid(diffuse)
    return torch.cat((rgb, density, normal), dim=-1)


def weights_expr(sigma, z, act_code: int):
    """sigma -> alpha -> exclusive transmittance product (nerf_base.py:80-86); z already scaled."""
    big = torch.full((z.shape[0], 1), 1e10, dtype=z.dtype, device=z.device)
    delta = torch.cat((z[:, 1:] - z[:, :-1], big), dim=-1)
    dens = F.relu(sigma) if act_code == ops.ACT_RELU else (F.softplus(sigma) if act_code == ops.ACT_SOFTPLUS else sigma)
    m = torch.exp(-dens * delta)
    ones = torch.ones((z.shape[0], 1), dtype=z.dtype, device=z.device)
    T = torch.cumprod(torch.cat((ones, m + 1e-10), dim=-1), dim=-1)[:, :-1]
    return (1.0 - m) * T


def max_blur_expr(w, alpha):
    mx = torch.maximum(w[..., :-1], w[..., 1:])
    return 0.5 * (torch.cat((w[..., :1], mx), dim=-1) + torch.cat((mx, w[..., -1:]), dim=-1)) + alpha


def bounds_expr(w, inds):
    sat = torch.cat((torch.zeros(w.shape[0], 1, device=w.device), torch.cumsum(w, dim=-1)), dim=-1)
    return torch.gather(sat, -1, inds[:, 1:] + 1) - torch.gather(sat, -1, inds[:, :-1])


def with_hip_backward(expr_fn: Callable, hip_bwd: Callable) -> Callable:
    """Attach a HIP backward to an expression: hip_bwd(grad, *args) -> one gradient (or None) per argument.  HipOp then calls it
    instead of re-evaluating `expr_fn` (which stays as the specification the HIP backward is tested against)."""
    expr_fn.hip_bwd = hip_bwd
    return expr_fn


class HipOp(torch.autograd.Function):
    """forward = `hip_fn(*tensors)` (HIP kernels);  backward = `expr_fn.hip_bwd` (HIP kernels) when the op has one, else the VJP
    of `expr_fn(*tensors)` (torch, on the device).  The first `expr_fn.n_diff` outputs of hip_fn (default 1) are differentiable --
    `expr_fn` returns as many, hip_bwd then receives a tuple of gradients with None for outputs the loss does not reach --; further
    outputs are returned as-is (non-differentiable)."""

    @staticmethod
    def forward(ctx, hip_fn: Callable, expr_fn: Callable, n_extra: int, *tensors):
        for t in tensors:
            if isinstance(t, torch.Tensor) and not t.is_cuda:
                raise RuntimeError("nerf_amd: tensors must live on the HIP device")
        ctx.expr_fn = expr_fn
        ctx.save_for_backward(*[t for t in tensors if isinstance(t, torch.Tensor)])
        ctx.is_tensor = [isinstance(t, torch.Tensor) for t in tensors]
        ctx.is_param = [isinstance(t, torch.nn.Parameter) for t in tensors]
        ctx.bf16 = ops.current_precision() == ops.BF16
        ctx.consts = [t for t in tensors if not isinstance(t, torch.Tensor)]
        ctx.n_diff = int(getattr(expr_fn, "n_diff", 1))
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
        saved = list(ctx.saved_tensors)
        consts = list(ctx.consts)
        hip_bwd = getattr(ctx.expr_fn, "hip_bwd", None)
        if hip_bwd is not None:
            it_t, it_c = iter(saved), iter(consts)
            full = [next(it_t) if is_t else next(it_c) for is_t in ctx.is_tensor]
            with torch.no_grad():
                grads = hip_bwd(grad, *full)
            if grads is not None:                                   # None = "not supported for these sizes": fall through to the VJP
                return (None, None, None, *grads)
        if not TORCH_VJP_FALLBACK:
            raise NotImplementedError(
                "nerf_amd: no HIP backward for this call (%s); the torch re-evaluation of the specification is off by default -- "
                "`with nerf_amd.autograd_bridge.allow_torch_vjp():` enables it" %
                ("the op's HIP backward declined these sizes" if hip_bwd is not None else "the op differentiates inputs only the generic VJP covers"))
        # only the inputs autograd actually asks for become leaves: e.g. the sample positions of MipNeRF carry no gradient, which
        # spares the VJP the first layer's dgrad and the whole sin/cos backward
        cache = getattr(ctx, "vjp_cache", None)
        ctx.vjp_cache = None
        prev, prev16 = _VJP.active, _VJP.bf16
        _VJP.active, _VJP.bf16 = True, ctx.bf16
        try:
            if cache is None:
                args, leaves, wanted = [], [], []
                for k, is_t in enumerate(ctx.is_tensor):
                    if is_t:
                        t = saved.pop(0)
                        want = t.is_floating_point() and ctx.needs_input_grad[3 + k]
                        if want:
                            t = t.detach().requires_grad_(True)
                            leaves.append(t)
                        wanted.append(want)
                        args.append(t)
                    else:
                        wanted.append(False)
                        args.append(consts.pop(0))
                if not leaves:
                    return (None, None, None, *[None] * len(args))
                with torch.enable_grad():
                    y = ctx.expr_fn(*args)
            else:
                y, leaves, wanted = cache
            if ctx.n_diff > 1:                                  # several differentiable outputs: keep the ones a gradient arrived for
                pairs = [(yy, gg) for yy, gg in zip(y, grad) if gg is not None]
                y, grad = tuple(p_[0] for p_ in pairs), tuple(p_[1] for p_ in pairs)
            if _VJP.inputs_only:
                # RefNeRF.get_grad (retain_graph): differentiate the non-parameter inputs only -- the Linear layers skip their weight
                # and bias gradients -- and keep the re-evaluated graph: the loss backward that follows on the same op re-uses it
                # instead of evaluating the expression a second time
                sel = [w and not p_ for w, p_ in zip(wanted, ctx.is_param)]
                it = iter(leaves)
                sub = [t for w, s_ in zip(wanted, sel) for t in ([next(it)] if w else []) if s_]
                part = iter(torch.autograd.grad(y, sub, grad, allow_unused=True, retain_graph=True)) if sub else iter(())
                ctx.vjp_cache = (y, leaves, wanted)
                return (None, None, None, *[next(part) if s_ else None for s_ in sel])
            grads = torch.autograd.grad(y, leaves, grad, allow_unused=True)
        finally:
            _VJP.active, _VJP.bf16 = prev, prev16
        gi = iter(grads)
        return (None, None, None, *[next(gi) if w else None for w in wanted])


def needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)
