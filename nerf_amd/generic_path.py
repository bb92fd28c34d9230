"""Networks LARGER than the shapes the fused MLP kernels are compiled for -- hidden widths above 256 (`--nerf_net_width` /
`--prop_net_width`, procedures.py:176-177) or more than 10 position octaves (mip_model.py:15-18, addtional.py:61) -- evaluated layer by
layer on the device: every `nn.Linear` (+ activation) of mip_model.py:41-60 / addtional.py:88-96 is one launch of the hand-written MFMA
GEMM `nerf_amd_gemm` (nerf_amd/csrc/generic_kernels.hip; explicit strides, so `W.t()` and column slices of a concatenated input are
views, never copies), the encodings are the stand-alone HIP encoders (`nerf_amd_positional_encoding`, `nerf_amd_encode_rows`), and the
backward is the same GEMM in its two other stride forms (input gradient with the ReLU mask in the epilogue; weight gradient = a
contraction over the samples, split over workgroups and summed in a fixed order) -- what torch.autograd computes for the reference's
modules, with no torch arithmetic and no library GEMM.  torch only owns the buffers (and concatenates / slices them).

This is the COMPATIBILITY path of the shape arguments: activations make a round trip through HBM per layer, so it runs at a fraction of
the fused kernels' rate -- every shape the fused kernels are compiled for (widths <= 256, <= 10 octaves, either cat_origin) keeps them.
Two routes: fp32 rows on `nerf_amd_gemm` (everything that is differentiated, and the fp32 parity mode), and -- round 5 -- bf16 rows on
`nerf_amd_rows_gemm` for forwards nobody differentiates under bf16 precision (rendering): the same values at a third of the traffic and
~3x the rate (`_rows_route`, `_skip_rows` below).  Sample positions get a gradient inside RefNeRF.get_grad only (d density / d position: a dgrad-only chain + the encoding's
adjoint, like on the fused path; the reference's loss never uses another, utils.py:35-36); scene contraction is a stage of its own in front of the
encoder here (`ops.contract_positions`, round 5); the integrated PE comes from the stand-alone encoder (`ops.ipe_feature`, with `contract=` since round 6: MipNeRF.forward_rays).  RefNeRF takes this path as well (`ref_forward`: hidden width > 256, > 10 octaves, or
`--ide_level 5`, whose 36 spherical-harmonic terms the fused kernel's three IDE K groups do not hold).
"""
from typing import List, Tuple

import torch

from . import autograd_bridge as ab
from . import ops

RELU, SIGMOID = 1, 2

# The inference route on bf16 rows (round 5; nerf_amd_rows_gemm): a forward that nobody differentiates, under bf16 precision, carries its
# activations from layer to layer as bf16 rows -- the values the fp32-row route computes (it rounds the same operands to bf16 on their way
# into LDS) at a third of the HBM traffic and 3x the product rate.  False = the fp32-row route everywhere (A/B measurements, tests).
ROWS_ROUTE = True


def _rows_route(prec: int, keep, layers) -> bool:
    """bf16 precision, no activations to keep for a backward, and every hidden width a multiple of 8 (16-byte row pieces)"""
    return ROWS_ROUTE and keep is None and (prec & 0xff) == ops.BF16 and all(l.out_features % 8 == 0 for l in layers)


def _packed(net, key, layers, columns=None) -> "ops.PackedLinear":
    """the layer's parameters in nerf_amd_rows_gemm's layout, with PackedWeightsMixin's cache rules: eval mode -- cached under (data_ptr,
    _version) of its tensors + ops.PARAM_GENERATION (torch's optimizer steps and load_state_dict bump `_version`, this package's Adam and
    replayed graphs bump the generation; `invalidate_packed()` / a train() / eval() switch drop
    the cache); train mode -- packed on every call (a hipGraph-replayed step changes parameters without `_version` moving).  Several
    `layers` = their rows stacked into one product (Ref-NeRF's heads)."""
    layers = layers if isinstance(layers, (list, tuple)) else [layers]
    cache = net.__dict__.setdefault("_rows_packed", {})
    stamp = (ops.PARAM_GENERATION[0],) + tuple((t.data_ptr(), t._version, str(t.device)) for l in layers for t in (l.weight, l.bias))
    hit = None if net.training else cache.get(key)
    if hit is None or hit[0] != stamp:
        w = torch.cat([l.weight.detach() for l in layers], dim=0) if len(layers) > 1 else layers[0].weight.detach()
        b = torch.cat([l.bias.detach() for l in layers], dim=0) if len(layers) > 1 else layers[0].bias.detach()
        hit = (stamp, ops.PackedLinear(w, b, columns))
        if not net.training:
            cache[key] = hit
    return hit[1]


def _bf16_rows(src: torch.Tensor, width: int = 0, col0: int = 0) -> torch.Tensor:
    """fp32 rows -> a fresh bf16 buffer (M, max(width, col0 + roundup(cols, 8))) holding them at column col0 (zero padding to the next 8)"""
    M, cols = src.shape
    buf = torch.empty((M, max(width, col0 + ops._pad(cols, 8))), dtype=torch.bfloat16, device=src.device)
    return ops.rows_to_bf16(src, buf, col0)


def _skip_rows(net, key_prefix, first, second, ex: torch.Tensor):
    """encoding -> `first` (4 x Linear+ReLU) -> cat(encoding, hidden) -> `second` (Linear+ReLU each) on bf16 rows: mip_model.py:53-56 /
    ref_model.py:74-77.  The concatenation keeps the hidden features FIRST (aligned column 0 for the product that writes them); the packed
    weight of second[0] has its input columns re-ordered to match."""
    M, E = ex.shape
    W = first[3].out_features
    h = _bf16_rows(ex)[:, :E]
    for i, l in enumerate(first[:3]):
        h = ops.rows_gemm(h, _packed(net, (key_prefix, 1, i), l), RELU)
    skip = torch.empty((M, W + ops._pad(E, 8)), dtype=torch.bfloat16, device=ex.device)
    ops.rows_gemm(h, _packed(net, (key_prefix, 1, 3), first[3]), RELU, out=skip[:, :W])
    ops.rows_to_bf16(ex, skip, W)
    h = ops.rows_gemm(skip[:, :W + E], _packed(net, (key_prefix, 2, 0), second[0], columns=[(E, W), (0, E)]), RELU)
    for i, l in enumerate(second[1:]):
        h = ops.rows_gemm(h, _packed(net, (key_prefix, 2, i + 1), l), RELU)
    return h


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
def proposal_forward(net, pts: torch.Tensor, contract: bool = False) -> torch.Tensor:
    """ProposalNetwork.forward (addtional.py:88-96) for a generic-shape module: pts (N,C,3) -> density (N,C).  `contract` (round 5):
    Mip-NeRF 360 scene contraction as a stage in front of the encoder (ops.contract_positions; the fused kernels do it in their sample
    fetch); get_grad pulls the encoding's gradient back through its Jacobian."""
    prec = ops.current_precision()
    layers = net._linear_layers()
    params = [l.weight for l in layers] + [l.bias for l in layers]
    shape = pts.shape[:-1]

    def run(p, keep=None):
        xc = p.reshape(-1, 3).float()
        if contract:
            xc = ops.contract_positions(xc)
        x = _encode_positions(xc, net.position_flevel, net.cat_origin)
        if _rows_route(prec, keep, layers[:4]):
            h = _bf16_rows(x)[:, :x.shape[1]]
            for i, l in enumerate(layers[:4]):
                h = ops.rows_gemm(h, _packed(net, ("prop", i), l), RELU)
            return ops.rows_gemm(h, _packed(net, ("prop", 4), layers[4]), out_dtype=torch.float32).view(shape)
        acts = [x]
        for l in layers[:4]:
            acts.append(_linear(prec, acts[-1], l, RELU))
        out = _linear(prec, acts[-1], layers[4])
        if keep is not None:
            keep["acts"] = acts
        return out.view(shape)

    if not ab.needs_grad(pts, *params):
        return run(pts)
    held = {}

    def bwd(g, p, *wb):
        if "acts" not in held:
            raise RuntimeError("nerf_amd: the activations of this forward were already consumed (backward twice over the same graph)")
        if ab._VJP.inputs_only:                                  # RefNeRF.get_grad(density, coarse_samples) (train.py:165-168, `prop_normal`)
            acts = held["acts"]
            delta = ops.gemm(prec, g.reshape(-1, 1).float().contiguous(), layers[4].weight.detach(), mask=acts[4])
            d_enc, _ = _dgrad_only(prec, delta, layers[:4], acts[:4], acts[0].shape[1])
            x_raw = p.reshape(-1, 3).float().contiguous()
            gx = ops.positional_encoding_backward(d_enc, ops.contract_positions(x_raw) if contract else x_raw, net.position_flevel, net.cat_origin)
            if contract:
                gx = ops.contract_positions(x_raw, grad=gx)
            return (gx.view(p.shape), *[None] * len(wb))
        acts = held.pop("acts")
        ones = torch.ones((acts[0].shape[0], 1), dtype=torch.float32, device=g.device)
        delta = g.reshape(-1, 1).float().contiguous()
        gW4, gb4 = _param_grads(prec, delta, acts[4], ones)
        delta = ops.gemm(prec, delta, layers[4].weight.detach(), mask=acts[4])
        gW, gb, _ = _chain_back(prec, delta, layers[:4], acts[:4], ones, first_cols=-1)
        return (None, *gW, gW4, *gb, gb4)
    return ab.HipOp.apply(lambda p, *wb: run(p, held), bwd, 1, pts, *params)


# ---------------------------------------------------------------------------------------------------------------- MipNeRF
def mip_forward(net, pts: torch.Tensor, contract: bool = False, encoded_x: torch.Tensor = None) -> torch.Tensor:
    """MipNeRF.forward (mip_model.py:41-60) for a generic-shape module: pts (N,S,6) = [position | raw direction] -> (N,S,4); `contract`:
    scene contraction of the positions in front of the encoder (they carry no gradient on this path); `encoded_x` (N,S,6 L): the
    encoding columns that follow the position, given instead of positional_encoding(position) -- the integrated PE of
    MipNeRF.forward_rays(ipe_radius=...) (mip_methods.py:47-58 through nerf_amd_ipe_feature), with the frustum mean as the position."""
    prec = ops.current_precision()
    L = net._linear_layers()             # lin_block1 x4, lin_block2 x3, bottle_neck, opacity_head, rgb_layer.0, rgb_layer.2
    params = net._params()
    shape = pts.shape[:-1]

    def run(p, keep=None):
        p2 = p.reshape(-1, 6).float()
        M = p2.shape[0]
        if encoded_x is not None:
            enc = encoded_x.reshape(M, -1).float()
            ex = torch.cat((p2[:, :3], enc), dim=-1) if net.cat_origin else enc.contiguous()
        else:
            ex = _encode_positions(ops.contract_positions(p2[:, :3]) if contract else p2[:, :3], net.position_flevel, net.cat_origin)
        ed = _encode_directions(p2[:, 3:6], net.cat_origin)
        E, W = ex.shape[1], net.hidden_unit
        if _rows_route(prec, keep, L[:8] + [L[9]]):
            g = _skip_rows(net, "mip", L[:4], L[4:7], ex)
            out = torch.empty((M, 4), dtype=torch.float32, device=p.device)
            ops.rows_gemm(g, _packed(net, ("mip", 8), L[8]), out=out[:, 3:4])                           # opacity_head (:57)
            Bn, Ed = L[7].out_features, ed.shape[1]
            head = torch.empty((M, Bn + ops._pad(Ed, 8)), dtype=torch.bfloat16, device=p.device)
            ops.rows_gemm(g, _packed(net, ("mip", 7), L[7]), out=head[:, :Bn])                          # bottle_neck (:58)
            ops.rows_to_bf16(ed, head, Bn)
            c = ops.rows_gemm(head[:, :Bn + Ed], _packed(net, ("mip", 9), L[9]), RELU)                  # rgb_layer.0 on cat(bottle-neck, encoded_r) (:59)
            ops.rows_gemm(c, _packed(net, ("mip", 10), L[10]), SIGMOID, out=out[:, :3])
            return out.view(*shape, 4)
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
        if "a" not in held:
            raise RuntimeError("nerf_amd: the activations of this forward were already consumed (backward twice over the same graph)")
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


# ---------------------------------------------------------------------------------------------------------------- RefNeRF
def _dgrad_only(prec: int, delta: torch.Tensor, layers, inputs, enc_cols: int):
    """dgrad-only chain (no parameter gradients: RefNeRF.get_grad) through Linear+ReLU `layers` whose first input is cat(encoding
    [enc_cols], hidden) (enc_cols == its whole width: a plain encoding).  -> (gradient w.r.t. the encoding columns, gradient w.r.t. the
    pre-activation that produced the hidden columns or None)."""
    for k in range(len(layers) - 1, 0, -1):
        delta = ops.gemm(prec, delta, layers[k].weight.detach(), mask=inputs[k])
    w0 = layers[0].weight.detach()
    d_enc = ops.gemm(prec, delta, w0[:, :enc_cols])
    d_hid = ops.gemm(prec, delta, w0[:, enc_cols:], mask=inputs[0][:, enc_cols:]) if w0.shape[1] > enc_cols else None
    return d_enc, d_hid


def ref_forward(net, pos: torch.Tensor, dirs: torch.Tensor, noise, contract: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """RefNeRF.forward (ref_model.py:68-106) for a module the fused Ref-NeRF kernel is not compiled for (hidden width > 256, > 10 position
    octaves, ide_level 5): pos, dirs (N,S,3) -> ((N,S,4) = [rgb | raw density], predicted normal (N,S,3)).  Layer products on nerf_amd_gemm,
    the stages between them (normal / reflection / IDE / colour combination) on the element-wise kernels of generic_ref_kernels.hip; the
    backward is the same GEMM in its other stride forms + those kernels' adjoints.  `noise` = the train-mode bottle-neck perturbation
    (ref_model.py:84-85) or None.  Position gradients exist for RefNeRF.get_grad only (d density / d position), like on the fused path."""
    prec = ops.current_precision()
    S1 = [net.spa_block1[i] for i in (0, 2, 4, 6)]
    S2 = [net.spa_block2[i] for i in (0, 2, 4, 6)]
    D1 = [net.dir_block1[i] for i in (0, 2, 4, 6)]
    D2 = [net.dir_block2[i] for i in (0, 2, 4, 6)]
    nct, rt, bn, sph = net.norm_col_tint_head, net.rho_tau_head, net.bottle_neck, net.spec_rgb_head[0]
    named = list(net.named_parameters())
    names = [n for n, _ in named]
    params = [p for _, p in named]
    shape = pos.shape[:-1]
    deg, Bd, T = net.sh_max_level, net.bottle_neck_dim, net.dir_enc_dim // 2
    Din = Bd + 2 * T + 1
    flags = net.kernel_flags
    table = net._ide_table(pos.device, deg)

    def run(p, dd, keep=None):
        x_raw = p.reshape(-1, 3).float().contiguous()
        x = ops.contract_positions(x_raw) if contract else x_raw             # (`contract`: the stage in front of the encoder, see proposal_forward)
        dv = dd.reshape(-1, 3).float().contiguous()
        M, dev = x.shape[0], x.device
        ex = _encode_positions(x, net.position_flevel, net.cat_origin)
        E, W = ex.shape[1], net.hidden_unit
        if noise is None and Bd % 8 == 0 and _rows_route(prec, keep, S1 + S2 + D1 + D2):
            g = _skip_rows(net, "spa", S1, S2, ex)
            heads = ops.rows_gemm(g, _packed(net, "heads", [nct, rt]), out_dtype=torch.float32)
            Wd = D1[3].out_features
            cat2 = torch.empty((M, Wd + ops._pad(Din, 8)), dtype=torch.bfloat16, device=dev)            # [r_tmp | bottle-neck | IDE real | IDE imag | n.d]
            ops.rows_gemm(g, _packed(net, "bn", bn), out=cat2[:, Wd:Wd + Bd])
            dir_in = torch.empty((M, Din - Bd), dtype=torch.float32, device=dev)
            normal = ops.ref_dir_inputs(heads, dv, deg, table, dir_in)
            ops.rows_to_bf16(dir_in, cat2, Wd + Bd)
            h = cat2[:, Wd:Wd + Din]
            for i, l in enumerate(D1[:3]):
                h = ops.rows_gemm(h, _packed(net, ("dir", 1, i), l), RELU)
            ops.rows_gemm(h, _packed(net, ("dir", 1, 3), D1[3]), RELU, out=cat2[:, :Wd])
            h = ops.rows_gemm(cat2[:, :Wd + Din], _packed(net, ("dir", 2, 0), D2[0], columns=[(Din, Wd), (0, Din)]), RELU)
            for i, l in enumerate(D2[1:]):
                h = ops.rows_gemm(h, _packed(net, ("dir", 2, i + 1), l), RELU)
            spec = ops.rows_gemm(h, _packed(net, "spec", sph), SIGMOID, out_dtype=torch.float32)
            return torch.cat((ops.ref_combine(heads, spec, flags), normal), dim=-1).view(*shape, 7)
        a = [ex]
        for l in S1[:3]:
            a.append(_linear(prec, a[-1], l, RELU))
        skip = torch.empty((M, E + W), dtype=torch.float32, device=dev)          # cat(encoded_x, x_tmp) (ref_model.py:76)
        skip[:, :E] = ex
        _linear(prec, a[-1], S1[3], RELU, out=skip[:, E:])
        b = [skip]
        for l in S2:
            b.append(_linear(prec, b[-1], l, RELU))
        g = b[-1]                                                               # `intermediate` (M, output_dim)
        # the three heads hang off g: one product on cat(norm_col_tint_head, rho_tau_head) (11 rows), one for the bottle-neck, written
        # straight into its columns of the directional input vector [bottle-neck | IDE real | IDE imag | n.d] = the front of cat(all_inputs, r_tmp)
        w_heads = torch.cat((nct.weight.detach(), rt.weight.detach()), dim=0)
        b_heads = torch.cat((nct.bias.detach(), rt.bias.detach()), dim=0)
        heads = ops.gemm(prec, g, w_heads.t(), bias=b_heads)
        cat2 = torch.empty((M, Din + W), dtype=torch.float32, device=dev)
        _linear(prec, g, bn, out=cat2[:, :Bd])
        if noise is not None:
            ops.add_rows_(cat2[:, :Bd], noise.reshape(-1, Bd).float())
        normal = ops.ref_dir_inputs(heads, dv, deg, table, cat2[:, Bd:Din])
        r = [cat2[:, :Din]]
        for l in D1[:3]:
            r.append(_linear(prec, r[-1], l, RELU))
        _linear(prec, r[-1], D1[3], RELU, out=cat2[:, Din:])
        q = [cat2]
        for l in D2:
            q.append(_linear(prec, q[-1], l, RELU))
        spec = _linear(prec, q[-1], sph, SIGMOID)
        rgbo = ops.ref_combine(heads, spec, flags)
        if keep is not None:
            keep.update(x=x, x_raw=x_raw, dv=dv, a=a, b=b, heads=heads, r=r, q=q, spec=spec, E=E, w_heads=w_heads)
        return torch.cat((rgbo, normal), dim=-1).view(*shape, 7)

    if not ab.needs_grad(pos, dirs, *params):
        out = run(pos, dirs)
        return out[..., :4].contiguous(), out[..., 4:].contiguous()
    held = {}

    def bwd(gr, p, dd, *wb):
        if "heads" not in held:
            raise RuntimeError("nerf_amd: the activations of this forward were already consumed (backward twice over the same graph)")
        g7 = gr.reshape(-1, 7).float().contiguous()
        M, dev = g7.shape[0], g7.device
        a, b, E = held["a"], held["b"], held["E"]
        g = b[-1]
        if ab._VJP.inputs_only:                                  # RefNeRF.get_grad: d density / d position, a dgrad-only chain
            delta = ops.gemm(prec, g7[:, 3:4], rt.weight.detach()[1:2, :], mask=g)
            d_enc, d_hid = _dgrad_only(prec, delta, S2, b[:4], E)
            d_enc1, _ = _dgrad_only(prec, d_hid, S1, a[:4], E)
            ops.add_rows_(d_enc, d_enc1)
            gx = ops.positional_encoding_backward(d_enc, held["x"], net.position_flevel, net.cat_origin)
            if contract:
                gx = ops.contract_positions(held["x_raw"], grad=gx)
            return (gx.view(p.shape), None, *[None] * len(wb))
        heads, r, q, spec, w_heads = (held.pop(k) for k in ("heads", "r", "q", "spec", "w_heads"))
        x, dv = held.pop("x"), held.pop("dv")
        held.pop("x_raw", None)
        held.pop("a"); held.pop("b")
        ones = torch.ones((M, 1), dtype=torch.float32, device=dev)
        G = {}
        dhb = torch.empty((M, 11 + Bd), dtype=torch.float32, device=dev)        # [d heads 11 | d bottle-neck]: one product back into g
        d_spec = ops.ref_combine_backward(g7[:, :4], heads, spec, flags, dhb)
        G["spec_rgb_head.0.weight"], G["spec_rgb_head.0.bias"] = _param_grads(prec, d_spec, q[4], ones)
        delta = ops.gemm(prec, d_spec, sph.weight.detach(), mask=q[4])
        # dir_block2 (its first input is cat(all_inputs [Din], r_tmp)): the all_inputs columns get an unmasked gradient, the hidden ones r_tmp's ReLU
        for k in (3, 2, 1):
            G["dir_block2.%d.weight" % (2 * k)], G["dir_block2.%d.bias" % (2 * k)] = _param_grads(prec, delta, q[k], ones)
            delta = ops.gemm(prec, delta, D2[k].weight.detach(), mask=q[k])
        G["dir_block2.0.weight"], G["dir_block2.0.bias"] = _param_grads(prec, delta, q[0], ones)
        w20 = D2[0].weight.detach()
        d_all = ops.gemm(prec, delta, w20[:, :Din])
        delta = ops.gemm(prec, delta, w20[:, Din:], mask=q[0][:, Din:])
        for k in (3, 2, 1):
            G["dir_block1.%d.weight" % (2 * k)], G["dir_block1.%d.bias" % (2 * k)] = _param_grads(prec, delta, r[k], ones)
            delta = ops.gemm(prec, delta, D1[k].weight.detach(), mask=r[k])
        G["dir_block1.0.weight"], G["dir_block1.0.bias"] = _param_grads(prec, delta, r[0], ones)
        ops.add_rows_(d_all, ops.gemm(prec, delta, D1[0].weight.detach()))
        ops.ref_dir_inputs_backward(heads, dv, deg, table, d_all[:, Bd:], g7[:, 4:7], dhb)
        dhb[:, 11:] = d_all[:, :Bd]                                              # the bottle-neck has no activation (and the noise no gradient)
        gh_w, gh_b = _param_grads(prec, dhb[:, :11], g, ones)
        G["norm_col_tint_head.weight"], G["norm_col_tint_head.bias"] = gh_w[:9], gh_b[:9]
        G["rho_tau_head.weight"], G["rho_tau_head.bias"] = gh_w[9:11], gh_b[9:11]
        G["bottle_neck.weight"], G["bottle_neck.bias"] = _param_grads(prec, dhb[:, 11:], g, ones)
        delta = ops.gemm(prec, dhb, torch.cat((w_heads, bn.weight.detach()), dim=0), mask=g)
        w2, b2, delta = _chain_back(prec, delta, S2, b[:4], ones, first_cols=E)
        w1, b1, _ = _chain_back(prec, delta, S1, a[:4], ones, first_cols=-1)
        for i, k in enumerate((0, 2, 4, 6)):
            G["spa_block2.%d.weight" % k], G["spa_block2.%d.bias" % k] = w2[i], b2[i]
            G["spa_block1.%d.weight" % k], G["spa_block1.%d.bias" % k] = w1[i], b1[i]
        return (None, None, *[G[n].reshape(wb[i].shape) for i, n in enumerate(names)])

    out = ab.HipOp.apply(lambda p, dd, *wb: run(p, dd, held), bwd, 1, pos, dirs, *params)
    return out[..., :4].clone(), out[..., 4:].clone()                            # (callers write into rgbo[..., -1] in place)
