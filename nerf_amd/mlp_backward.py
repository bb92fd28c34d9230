"""Backward of the two MLPs from the HIP training forward's activation dump.

The forward kernels (nerf_amd_{proposal,mip}_forward_train) leave every hidden layer's post-ReLU activations in HBM; the
gradient is then a chain of plain GEMMs -- dgrad `delta @ W`, wgrad `delta^T @ activation` -- which are library GEMMs
(hipBLASLt through torch.matmul) in the kernels' own arithmetic (bf16 operands with fp32 accumulation, or fp32), plus ReLU masks
taken from the stored activations.  Nothing of the forward is re-evaluated except the positional encodings (elementwise) and,
for MipNeRF, the bottle-neck vector the folded forward kernel never materialises (one GEMM).

Reference semantics: what torch.autograd computes for addtional.py:88-96 and mip_model.py:41-60.
Tensor order of `weights` / returned gradients = the modules' `_linear_layers()` order (weights first, then biases).
"""
from typing import List, Sequence, Tuple

import torch

from . import ops


_SPLIT = 4096        # minimum rows per split-K slice of a wgrad GEMM
_SLICES = 128        # ... and the number of slices aimed at for large M (128 slices x 2 output tiles fill the 256 CUs)


def _pad16(t: torch.Tensor) -> torch.Tensor:
    """(M, O < 16) -> (M, 16) with zero columns: the library has no efficient kernels for 1- or 3-wide operands."""
    out = torch.zeros((t.shape[0], 16), dtype=t.dtype, device=t.device)
    out[:, : t.shape[1]] = t
    return out


def _encode(x: torch.Tensor, L: int, precision: int, normalize: bool = False) -> torch.Tensor:
    """[x | PE_L(x)] zero-padded to a multiple of 8 columns (the library's bf16 kernels want aligned leading dimensions; odd ones
    fall into a path with milliseconds of host-side search per call) -- one HIP kernel (nerf_amd_encode_rows)."""
    return ops.encode_rows(x, L, precision, normalize)


def _bgrad_narrow(d16: torch.Tensor, n: int) -> torch.Tensor:
    """Column sums of an (M, 16) matrix whose first n columns matter: summed as (M/16, 256) so that the reduction runs over
    full-width rows (a 3- or 16-wide column reduce of millions of rows is a millisecond on its own)."""
    M = d16.shape[0]
    main = M // 16 * 16
    out = d16[:main].view(-1, 256).sum(0, dtype=torch.float32).view(16, 16).sum(0)
    if main < M:
        out = out + d16[main:].sum(0, dtype=torch.float32)
    return out[:n]


def _wgrad(delta: torch.Tensor, act: torch.Tensor) -> torch.Tensor:
    """delta^T @ act with the sample dimension as K.  The library picks a 16-workgroup kernel for a (256 x M) @ (M x 256) product, so
    K is split by hand: one batched GEMM over slices of >= _SPLIT rows (about _SLICES of them for large M), partial products leave the GEMM in fp32 and are summed in fp32."""
    M, O = delta.shape
    split = max(_SPLIT, M // _SLICES // _SPLIT * _SPLIT)
    n = M // split
    f32 = {} if delta.dtype == torch.float32 else {"out_dtype": torch.float32}    # bf16 operands: products leave the GEMM unrounded
    if n < 2:
        return torch.mm(delta.t(), act, **f32)
    if O < 16:                                            # the 1- and 3-wide head gradients: pad to a real GEMM shape
        return _wgrad(_pad16(delta), act)[:O]
    main = n * split
    out = torch.bmm(delta[:main].view(n, split, -1).transpose(1, 2), act[:main].view(n, split, -1), **f32).sum(0)
    if main < M:
        out += torch.mm(delta[main:].t(), act[main:], **f32)
    return out


def _bgrad(delta: torch.Tensor) -> torch.Tensor:
    return delta.sum(0, dtype=torch.float32)


def proposal_backward(g_density: torch.Tensor, pts: torch.Tensor, dump: torch.Tensor, precision: int,
                      weights: Sequence[torch.Tensor]) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """g_density (M,), pts (M,3), weights = [layers.0, .2, .4, .6, .8].weight -> ([dW]*5, [db]*5)"""
    M = pts.shape[0]
    dt = torch.bfloat16 if precision == ops.BF16 else torch.float32
    W = [w.detach().to(dt) for w in weights]
    # one pass per hidden layer: its activations as rows (the wgrad operand) + the ReLU mask of the incoming delta + the bias gradient
    rows_mask = lambda l, d: ops.train_dump_rows_mask_(dump, ops.NET_PROPOSAL, precision, l, d)
    gW, gb = [None] * 5, [None] * 5
    g = g_density.reshape(M, 1).to(dt)
    h, delta, gb[3] = rows_mask(3, g * W[4])
    gW[4], gb[4] = _wgrad(g, h), g_density.sum().reshape(1)
    for l in (3, 2, 1):
        prev, nxt, gb[l - 1] = rows_mask(l - 1, torch.mm(delta, W[l]))
        gW[l] = _wgrad(delta, prev)
        delta = nxt
    gW[0] = _wgrad(delta, _encode(pts, 10, precision))[:, :63]
    return gW, gb


def mip_backward(g_rgbo: torch.Tensor, rgbo: torch.Tensor, pts: torch.Tensor, dump: torch.Tensor, precision: int,
                 weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor]) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """g_rgbo, rgbo (M,4), pts (M,6); weights/biases in MipNeRF._linear_layers() order (lin_block1.{0,2,4,6}, lin_block2.{0,2,4},
    bottle_neck.0, opacity_head.0, rgb_layer.{0,2}) -> ([dW]*11, [db]*11)"""
    M = pts.shape[0]
    dt = torch.bfloat16 if precision == ops.BF16 else torch.float32
    W = [w.detach().to(dt) for w in weights]
    rows_mask = lambda l, d: ops.train_dump_rows_mask_(dump, ops.NET_MIP, precision, l, d)
    gW, gb = [None] * 11, [None] * 11
    ed = _encode(pts[:, 3:6], 4, precision, normalize=True)                         # (M, 27 -> 32), d / |d| first (mip_model.py:52)
    # colour head: rgb = sigmoid(rgb_layer.2(c)), c = relu(rgb_layer.0(cat(bottle_neck(g6), ed)))
    rgb = rgbo[:, :3]
    d10p = _pad16((g_rgbo[:, :3] * rgb * (1.0 - rgb)).to(dt))
    w10 = torch.zeros((16, W[10].shape[1]), dtype=dt, device=W[10].device)
    w10[:3] = W[10]
    c, dc, gb[9] = rows_mask(7, torch.mm(d10p, w10))                                # 128 features
    gW[10], gb[10] = _wgrad(d10p, c)[:3], _bgrad_narrow(d10p, 3)
    dbott = torch.mm(dc, W[9][:, :256].contiguous())
    dsig = g_rgbo[:, 3:4].to(dt)
    g6, delta, gb[6] = rows_mask(6, torch.addmm(dsig * W[8], dbott, W[7]))
    bott = torch.addmm(biases[7].detach().to(dt), g6, W[7].t())                   # the folded forward never forms it
    gW[9] = torch.cat((_wgrad(dc, bott), _wgrad(dc, ed)[:, :27]), dim=1)          # cat(bottle_neck, dir_enc) column blocks
    gW[7], gb[7] = _wgrad(dbott, g6), _bgrad(dbott)
    gW[8], gb[8] = _wgrad(dsig, g6), g_rgbo[:, 3].sum().reshape(1)
    del bott, dbott, dc, c, g6
    for l in (6, 5):
        prev, nxt, gb[l - 1] = rows_mask(l - 1, torch.mm(delta, W[l]))
        gW[l] = _wgrad(delta, prev)
        delta = nxt
    ex = _encode(pts[:, :3], 10, precision)                                         # (M, 63 -> 64)
    h3, nxt, gb[3] = rows_mask(3, torch.mm(delta, W[4][:, 63:].contiguous()))
    gW[4] = torch.cat((_wgrad(delta, ex)[:, :63], _wgrad(delta, h3)), dim=1)      # skip layer: cat(encoded_x, h)
    delta = nxt
    del h3
    for l in (3, 2, 1):
        prev, nxt, gb[l - 1] = rows_mask(l - 1, torch.mm(delta, W[l]))
        gW[l] = _wgrad(delta, prev)
        delta = nxt
    gW[0] = _wgrad(delta, ex)[:, :63]
    return gW, gb
