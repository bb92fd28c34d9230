"""Networks LARGER than the shapes the fused MLP kernels are compiled for -- hidden widths above 256 (`--nerf_net_width` /
`--prop_net_width`, procedures.py:176-177) or more than 10 position octaves (mip_model.py:15-18, addtional.py:61) -- evaluated layer by
layer on the device: every `nn.Linear` (+ activation) of mip_model.py:41-60 / addtional.py:88-96 is one launch of the hand-written MFMA
GEMM `nerf_amd_gemm` (nerf_amd/csrc/generic_kernels.hip; explicit strides, so `W.t()` and column slices of a concatenated input are
views, never copies), the encodings are the stand-alone HIP encoders (`nerf_amd_positional_encoding`, `nerf_amd_encode_rows`), and the
backward is the same GEMM in its two other stride forms (input gradient with the ReLU mask in the epilogue; weight gradient = a
contraction over the samples, split over workgroups and summed in a fixed order) -- what torch.autograd computes for the reference's
modules, with no torch arithmetic and no library GEMM.  torch only owns the buffers (and concatenates / slices them).

This is the COMPATIBILITY path of the shape arguments: activations make a round trip through HBM per layer (fp32 rows), so it runs at a
fraction of the fused kernels' rate -- every shape the fused kernels are compiled for (widths <= 256, <= 10 octaves, either cat_origin)
keeps them.  Sample positions get no gradient here (the reference's loss never uses it, utils.py:35-36); scene contraction and the
integrated PE are flags of the fused kernels' sample fetch only.
"""
from typing import List, Tuple

import torch

from . import autograd_bridge as ab
from . import ops

RELU, SIGMOID = 1, 2


def _encode_positions(x: torch.Tensor, levels: int, cat_origin: bool) -> torch.Tensor:
    """(M,3) -> [x | sin 2^0 x | cos 2^0 x | ...] (nerf_helper.py:38-48 behind the raw position, mip_model.py:50-51)"""
    x = x.contiguous()
    pe = ops.positional_encoding(x, levels)
    return torch.cat((x, pe), dim=-1) if cat_origin else pe


def _encode_directions(d: torch.Tensor, cat_origin: bool) -> torch.Tensor:
    """(M,3) raw directions -> [d/|d| | PE_4(d/|d|)] (mip_model.py:45-47,52) as a view of the encoder's 32-column rows"""
    rows = ops.encode_rows(d.contiguous(), 4, ops.F32, normalize=True)
    return rows[:, :27] if cat_origin else rows[:, 3:27]


def _linear(prec: int, x: torch.Tensor, layer: torch.nn.Linear, act: int = 0, out: torch.Tensor = None) -> torch.Tensor:
    return ops.gemm(prec, x, layer.weight.detach().t(), out=out, bias=layer.bias.detach(), act=act)


def _param_grads(prec: int, delta: torch.Tensor, x: torch.Tensor, ones: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """nn.Linear's parameter gradients: dW = delta^T x (out, in), db = delta^T 1"""
    return ops.gemm(prec, delta.t(), x), ops.gemm(prec, delta.t(), ones).reshape(-1)


def _chain_back(prec: int, delta: torch.Tensor, layers: List[torch.nn.Linear], inputs: List[torch.Tensor], ones: torch.Tensor, first_cols: int = 0):
    """Backward through Linear+ReLU layers `layers` (applied in this order) whose inputs were `inputs`; `delta` = gradient w.r.t. the LAST
    layer's post-ReLU-masked pre-activation.  -> ([dW], [db]) in the same order and the gradient w.r.t. the pre-activation that produced
    inputs[0][:, first_cols:] (None when first_cols < 0: the first layer's input is an encoding, nothing upstream)."""
    gW, gb = [None] * len(layers), [None] * len(layers)
    for k in range(len(layers) - 1, -1, -1):
        gW[k], gb[k] = _param_grads(prec, delta, inputs[k], ones)
        if k > 0:                                               # d(input) = (delta W) . [input > 0]: the ReLU that produced this layer's input
            delta = ops.gemm(prec, delta, layers[k].weight.detach(), mask=inputs[k])
        elif first_cols >= 0:                                   # the hidden part of a cat(encoding, hidden) input
            delta = ops.gemm(prec, delta, layers[0].weight.detach()[:, first_cols:], mask=inputs[0][:, first_cols:])
        else:
            delta = None
    return gW, gb, delta


# ---------------------------------------------------------------------------------------------------------------- ProposalNetwork
def proposal_forward(net, pts: torch.Tensor) -> torch.Tensor:
    """ProposalNetwork.forward (addtional.py:88-96) for a generic-shape module: pts (N,C,3) -> density (N,C)."""
    prec = ops.current_precision()
    layers = net._linear_layers()
    params = [l.weight for l in layers] + [l.bias for l in layers]
    shape = pts.shape[:-1]

    def run(p, keep=None):
        x = _encode_positions(p.reshape(-1, 3).float(), net.position_flevel, net.cat_origin)
        acts = [x]
        for l in layers[:4]:
            acts.append(_linear(prec, acts[-1], l, RELU))
        out = _linear(prec, acts[-1], layers[4])
        if keep is not None:
            keep["acts"] = acts
        return out.view(shape)

    if not ab.needs_grad(pts, *params):
        return run(pts)
    if pts.requires_grad:
        ab.unsupported("a generic-shape ProposalNetwork (hidden width > 256 or > 10 octaves) with sample positions that require a gradient")
    held = {}

    def bwd(g, p, *wb):
        acts = held.pop("acts")
        ones = torch.ones((acts[0].shape[0], 1), dtype=torch.float32, device=g.device)
        delta = g.reshape(-1, 1).float().contiguous()
        gW4, gb4 = _param_grads(prec, delta, acts[4], ones)
        delta = ops.gemm(prec, delta, layers[4].weight.detach(), mask=acts[4])
        gW, gb, _ = _chain_back(prec, delta, layers[:4], acts[:4], ones, first_cols=-1)
        return (None, *gW, gW4, *gb, gb4)
    return ab.HipOp.apply(lambda p, *wb: run(p, held), bwd, 1, pts, *params)


# ---------------------------------------------------------------------------------------------------------------- MipNeRF
def mip_forward(net, pts: torch.Tensor) -> torch.Tensor:
    """MipNeRF.forward (mip_model.py:41-60) for a generic-shape module: pts (N,S,6) = [position | raw direction] -> (N,S,4)."""
    prec = ops.current_precision()
    L = net._linear_layers()             # lin_block1 x4, lin_block2 x3, bottle_neck, opacity_head, rgb_layer.0, rgb_layer.2
    params = net._params()
    shape = pts.shape[:-1]

    def run(p, keep=None):
        p2 = p.reshape(-1, 6).float()
        M = p2.shape[0]
        ex = _encode_positions(p2[:, :3], net.position_flevel, net.cat_origin)
        ed = _encode_directions(p2[:, 3:6], net.cat_origin)
        E, W = ex.shape[1], net.hidden_unit
        a = [ex]
        for l in L[:3]:
            a.append(_linear(prec, a[-1], l, RELU))
        skip = torch.empty((M, E + W), dtype=torch.float32, device=p.device)      # cat(encoded_x, tmp) (mip_model.py:55): the last layer of
        skip[:, :E] = ex                                                        # lin_block1 writes straight into its column range
        _linear(prec, a[-1], L[3], RELU, out=skip[:, E:])
        b = [skip]
        for l in L[4:7]:
            b.append(_linear(prec, b[-1], l, RELU))
        g = b[-1]                                                               # (M, 256)
        out = torch.empty((M, 4), dtype=torch.float32, device=p.device)
        _linear(prec, g, L[8], out=out[:, 3:4])                                 # opacity_head (:57)
        head = torch.empty((M, 256 + ed.shape[1]), dtype=torch.float32, device=p.device)
        head[:, 256:] = ed
        _linear(prec, g, L[7], out=head[:, :256])                               # bottle_neck, no activation (:58)
        c = _linear(prec, head, L[9], RELU)                                     # rgb_layer.0 on cat(bottle-neck, encoded_r) (:59)
        _linear(prec, c, L[10], SIGMOID, out=out[:, :3])
        if keep is not None:
            keep.update(a=a, b=b, head=head, c=c, out=out, E=E)
        return out.view(*shape, 4)

    if not ab.needs_grad(pts, *params):
        return run(pts)
    if pts.requires_grad:                                                        # the reference's loss never differentiates the fine positions (utils.py:35-36)
        ab.unsupported("MipNeRF.forward with sample positions that require a gradient")
    held = {}

    def bwd(gr, p, *wb):
        a, b, head, c, out, E = (held.pop(k) for k in ("a", "b", "head", "c", "out", "E"))
        gr = gr.reshape(-1, 4).float().contiguous()
        M = gr.shape[0]
        ones = torch.ones((M, 1), dtype=torch.float32, device=gr.device)
        gW, gb = [None] * 11, [None] * 11
        d_rgb = ops.sigmoid_backward(gr[:, :3], out[:, :3])                      # rgb_layer.2 + sigmoid
        gW[10], gb[10] = _param_grads(prec, d_rgb, c, ones)
        d_c = ops.gemm(prec, d_rgb, L[10].weight.detach(), mask=c)               # through rgb_layer.0's ReLU
        gW[9], gb[9] = _param_grads(prec, d_c, head, ones)
        # bottle_neck (no activation) and opacity_head both hang off g: one product [d_bottle | d_sigma] . [W_bottle ; W_opacity], masked by g's ReLU
        dcat = torch.empty((M, 257), dtype=torch.float32, device=gr.device)
        ops.gemm(prec, d_c, L[9].weight.detach()[:, :256], out=dcat[:, :256])
        dcat[:, 256] = gr[:, 3]
        g = b[-1]
        gW[7], gb[7] = _param_grads(prec, dcat[:, :256], g, ones)
        gW[8], gb[8] = _param_grads(prec, dcat[:, 256:257], g, ones)
        wcat = torch.cat((L[7].weight.detach(), L[8].weight.detach()), dim=0)
        delta = ops.gemm(prec, dcat, wcat, mask=g)
        w2, b2, delta = _chain_back(prec, delta, L[4:7], b[:3], ones, first_cols=E)     # lin_block2; on to the hidden half of the skip input
        gW[4:7], gb[4:7] = w2, b2
        w1, b1, _ = _chain_back(prec, delta, L[:4], a[:4], ones, first_cols=-1)          # lin_block1
        gW[:4], gb[:4] = w1, b1
        return (None, *gW, *gb)
    return ab.HipOp.apply(lambda p, *wb: run(p, held), bwd, 1, pts, *params)
