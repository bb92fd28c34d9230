"""Backward of the two MLPs from the HIP training forward's activation dump -- every product runs in
hand-written gfx950 kernels (nerf_amd/csrc/bwd_kernels.hip); torch only owns the buffers:

  1. dgrad chain (nerf_amd_{proposal,mip}_backward_chain): delta stays in registers from the heads to the first hidden layer, the
     transposed weights stream through LDS, the ReLU adjoint comes from the activation dump; every layer's delta is dumped in the
     same fragment order;
  2. weight gradients (nerf_amd_{proposal,mip}_weight_grads): delta^T . activations contracted over the samples on the matrix cores,
     per-workgroup partials summed in a fixed order (bit-reproducible), bias gradients on the side, written in the reference's
     (out, in) layout; the bottle_neck.0 / rgb_layer.0 pair the forward folds together is un-folded in parameter space.

Reference semantics: what torch.autograd computes for addtional.py:88-96 and mip_model.py:41-60.
Tensor order of `weights` / returned gradients = the modules' `_linear_layers()` order.
"""
from typing import List, Optional, Sequence, Tuple

import torch

from . import ops


def proposal_backward(g_density: torch.Tensor, pts: torch.Tensor, dump: torch.Tensor, precision: int, weights: Sequence[torch.Tensor],
                      packed_bwd: Optional[torch.Tensor] = None, out=None) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """g_density (M,), pts (M,3) [unused: the encoding operand is in the dump], weights = layers.{0,2,4,6,8}.weight -> ([dW]*5, [db]*5);
    `out` = (weight sinks, bias sinks) to write instead of fresh tensors"""
    M = g_density.numel()
    if packed_bwd is None:
        packed_bwd = ops.pack_weights_backward(ops.NET_PROPOSAL, ops.BF16 if precision == ops.BF16_F8 else precision, weights)
    delta = ops.proposal_backward_chain(packed_bwd, precision, g_density, dump)
    return ops.proposal_weight_grads(precision, M, dump, delta, out=out)


def mip_backward(g_rgbo: torch.Tensor, rgbo: torch.Tensor, pts: torch.Tensor, dump: torch.Tensor, precision: int,
                 weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
                 packed_bwd: Optional[torch.Tensor] = None, out=None) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """g_rgbo, rgbo (M,4); weights/biases in MipNeRF._linear_layers() order -> ([dW]*11, [db]*11); `out` as in proposal_backward"""
    M = g_rgbo.numel() // 4
    if packed_bwd is None:
        packed_bwd = ops.pack_weights_backward(ops.NET_MIP, ops.BF16 if precision == ops.BF16_F8 else precision, weights)
    delta = ops.mip_backward_chain(packed_bwd, precision, g_rgbo, rgbo, dump)
    return ops.mip_weight_grads(precision, M, dump, delta, weights, biases, out=out)
