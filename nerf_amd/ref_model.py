"""Host-side mirror of the reference's ``nerf/ref_model.py``: RefNeRF with the reference's module tree (identical
``state_dict`` keys) whose ``forward`` runs the fused HIP kernel (spatial MLP -> heads -> normals / reflection / IDE in
registers -> directional MLP), plus the two scalar losses."""
from typing import Optional, Tuple

import torch
from torch import nn
from torch.nn import functional as F

from . import autograd_bridge as ab
from . import ops
from ._packed import PackedWeightsMixin, require_no_grad
from .nerf_base import NeRF
from .nerf_helper import makeMLP
from .ref_func import generate_ide_fn, ide_table


class RefNeRF(PackedWeightsMixin, NeRF):
    _net_id = ops.NET_REF

    def __init__(self, position_flevel, sh_max_level, bottle_neck_dim=128, hidden_unit=256, output_dim=256, use_srgb=False,
                 cat_origin=True, perturb_bottle_neck_w=0.1) -> None:
        super().__init__(position_flevel, cat_origin, lambda x: x)            # density is not activated during render
        self.sh_max_level = sh_max_level
        self.bottle_neck_dim = bottle_neck_dim
        self.hidden_unit, self.output_dim = hidden_unit, output_dim
        self.dir_enc_dim = ((1 << sh_max_level) - 1 + sh_max_level) << 1
        in_dim = 6 * position_flevel + (3 if cat_origin else 0)
        spa1 = makeMLP(in_dim, hidden_unit)
        for _ in range(3):
            spa1.extend(makeMLP(hidden_unit, hidden_unit))
        self.spa_block1 = nn.Sequential(*spa1)
        self.spa_block2 = nn.Sequential(*makeMLP(hidden_unit + in_dim, hidden_unit), *makeMLP(hidden_unit, hidden_unit),
                                        *makeMLP(hidden_unit, hidden_unit), *makeMLP(hidden_unit, output_dim))
        self.rho_tau_head = nn.Linear(output_dim, 2)                           # roughness, density
        self.norm_col_tint_head = nn.Linear(output_dim, 9)                     # normal, diffuse colour, tint
        self.bottle_neck = nn.Linear(output_dim, bottle_neck_dim)
        self.spec_rgb_head = nn.Sequential(*makeMLP(output_dim, 3, nn.Sigmoid()))
        dir_in = 1 + bottle_neck_dim + self.dir_enc_dim
        dir1 = makeMLP(dir_in, hidden_unit)
        for _ in range(3):
            dir1.extend(makeMLP(hidden_unit, hidden_unit))
        self.dir_block1 = nn.Sequential(*dir1)
        self.dir_block2 = nn.Sequential(*makeMLP(hidden_unit + dir_in, hidden_unit), *makeMLP(hidden_unit, hidden_unit),
                                        *makeMLP(hidden_unit, output_dim), *makeMLP(hidden_unit, output_dim))
        self.use_srgb = use_srgb
        self.perturb_bottle_neck_w = perturb_bottle_neck_w
        self.integrated_dir_enc = generate_ide_fn(sh_max_level)
        self.apply(self.init_weight)

    def _check_config(self):
        # hidden widths below 256 (`--nerf_net_width`, train.py:80) run on the same kernels through exact zero-padding, like MipNeRF
        # (_packed.py); the reference itself needs hidden_unit == output_dim (dir_block2.6 takes hidden_unit inputs from an output_dim-wide layer)
        # `--ide_level` 1..3 (procedures.py:211, train.py:80) run on the level-4 kernel as well: the (m, l) terms of level d are the first
        # T_d = 2, 5, 10 of level 4's 19 (ref_func.py:56-58 orders them by l), so the directional layers' weights are embedded into the
        # level-4 column layout [bottle-neck 128 | real 19 | imag 19 | n.d] with zeros on the terms the module does not have.  Level 5
        # (36 terms) does not fit the kernel's three IDE K groups.
        # position_flevel < 10 / cat_origin=False: zero weights on the position-encoding columns the module lacks (_embed_pos, as in _packed.py)
        # Shapes the fused kernel is NOT compiled for -- hidden width above 256, more than 10 position octaves, `--ide_level 5` (36 terms), another
        # bottle-neck width -- run layer by layer on the generic MFMA GEMM + the element-wise Ref-NeRF stages (nerf_amd/generic_path.py
        # `ref_forward`, generic_ref_kernels.hip), forward and backward.
        if not (self.position_flevel >= 1 and 1 <= self.sh_max_level <= 5 and self.bottle_neck_dim >= 1 and self.hidden_unit >= 1):
            raise NotImplementedError("nerf_amd: RefNeRF needs position_flevel >= 1, ide_level 1..5 (ref_func.py:63: higher levels are numerically "
                                      "unstable in the reference too) and positive widths")
        if self.output_dim != self.hidden_unit:
            raise NotImplementedError("nerf_amd: RefNeRF needs hidden_unit == output_dim (the reference's own dir_block2.6 takes hidden_unit inputs "
                                      "from an output_dim-wide layer, ref_model.py:56-58)")

    def _generic(self) -> bool:
        return self.hidden_unit > 256 or self.position_flevel > 10 or self.sh_max_level > 4 or self.bottle_neck_dim != 128

    def _pos_segment(self):
        return self.encoding_segment(self.position_flevel, self.cat_origin)

    def _embed_pos(self, w: torch.Tensor) -> torch.Tensor:
        """spa_block{1,2}.0 weight (rows, enc [+ hidden]) -> (rows, 63 [+ hidden]): the module's encoding columns placed in the level-10 layout"""
        if self.position_flevel == 10 and self.cat_origin:
            return w
        kc, mc, n = self._pos_segment()
        out = torch.zeros((w.shape[0], 63 + w.shape[1] - n), dtype=w.dtype, device=w.device)
        out[:, kc: kc + n] = w[:, :n]
        out[:, 63:] = w[:, n:]
        return out

    def _extract_pos(self, g: torch.Tensor) -> torch.Tensor:
        if self.position_flevel == 10 and self.cat_origin:
            return g
        kc, mc, n = self._pos_segment()
        return torch.cat((g[:, kc: kc + n], g[:, 63:]), dim=1)

    def _dir_cols(self):
        """column of the level-4 directional input vector (167 wide) that each of the module's 128 + 2 T + 1 directional inputs occupies"""
        T = self.dir_enc_dim // 2
        return list(range(128)) + [128 + t for t in range(T)] + [128 + 19 + t for t in range(T)] + [166]

    def _embed_dir(self, w: torch.Tensor) -> torch.Tensor:
        """dir_block{1,2}.0 weight (rows, 128 + 2 T + 1 [+ hidden]) -> (rows, 167 [+ hidden]) in the kernel's column layout"""
        if self.sh_max_level == 4:
            return w
        n_in = 129 + self.dir_enc_dim
        out = torch.zeros((w.shape[0], 167 + w.shape[1] - n_in), dtype=w.dtype, device=w.device)
        out[:, self._dir_cols()] = w[:, :n_in]
        out[:, 167:] = w[:, n_in:]
        return out

    def _extract_dir(self, g: torch.Tensor) -> torch.Tensor:
        if self.sh_max_level == 4:
            return g
        return torch.cat((g[:, self._dir_cols()], g[:, 167:]), dim=1)

    # kernel shapes of the 20 packed tensors (spatial 0..7, bottle_neck 8, heads 9, directional 10..17, spec head 18; the IDE table is not padded)
    _KERNEL_SHAPES = ([(256, 63)] + [(256, 256)] * 3 + [(256, 319)] + [(256, 256)] * 3 + [(128, 256), (11, 256), (256, 167)] + [(256, 256)] * 3 +
                      [(256, 423)] + [(256, 256)] * 3 + [(3, 256)])

    @staticmethod
    def _pad_to(t, shape):
        if tuple(t.shape) == tuple(shape):
            return t
        out = torch.zeros(tuple(shape), dtype=t.dtype, device=t.device)
        out[tuple(slice(0, n) for n in t.shape)] = t
        return out

    @property
    def kernel_flags(self) -> int:
        """ref_flags of the C-ABI's Ref-NeRF entry points (NERF_AMD_REF_SRGB = use_srgb, ref_model.py:100-102)"""
        return ops.REF_SRGB if self.use_srgb else 0

    def _pack_tensors(self):
        nct, rt = self.norm_col_tint_head, self.rho_tau_head
        # head rows in kernel order: normal(3) roughness(1) | diffuse(3) density(1) | tint(3)   (mlp_layout.h)
        hw = torch.cat((nct.weight[0:3], rt.weight[0:1], nct.weight[3:6], rt.weight[1:2], nct.weight[6:9]), dim=0).detach().contiguous()
        hb = torch.cat((nct.bias[0:3], rt.bias[0:1], nct.bias[3:6], rt.bias[1:2], nct.bias[6:9]), dim=0).detach().contiguous()
        lin = [self.spa_block1[0], self.spa_block1[2], self.spa_block1[4], self.spa_block1[6],
               self.spa_block2[0], self.spa_block2[2], self.spa_block2[4], self.spa_block2[6], self.bottle_neck]
        tail = [self.dir_block1[0], self.dir_block1[2], self.dir_block1[4], self.dir_block1[6],
                self.dir_block2[0], self.dir_block2[2], self.dir_block2[4], self.dir_block2[6], self.spec_rgb_head[0]]
        tables = self.__dict__.setdefault("_ide_table_on", {})   # constant: built (float64 host loops) and uploaded once per device
        if hw.device not in tables:
            tables[hw.device] = ide_table(4).to(hw.device).contiguous()
        table = tables[hw.device]
        ws = [l.weight for l in lin] + [hw] + [l.weight for l in tail]
        bs = [l.bias for l in lin] + [hb] + [l.bias for l in tail]
        if self.sh_max_level != 4:
            ws[10], ws[14] = self._embed_dir(ws[10].detach()), self._embed_dir(ws[14].detach())       # dir_block1.0, dir_block2.0
        ws[0], ws[4] = self._embed_pos(ws[0].detach()), self._embed_pos(ws[4].detach())               # spa_block1.0, spa_block2.0
        if self.hidden_unit != 256:                          # zero-padded to the compiled 256-wide shapes: the same function (hidden features are
            with torch.no_grad():                            # always the LAST column segment of a layer's input, so the padding goes at the end)
                ws = [self._pad_to(w.detach(), sh) for w, sh in zip(ws, self._KERNEL_SHAPES)]
                bs = [self._pad_to(b.detach(), (sh[0],)) for b, sh in zip(bs, self._KERNEL_SHAPES)]
        return ws + [table], bs + [hb]

    def _pack_now(self, precision: int, narrow: bool = False) -> torch.Tensor:
        ws, bs = self._pack_tensors()
        return ops.pack_weights(self._net_id, precision, ws, bs)

    def forward(self, pts: torch.Tensor, ray_d: Optional[torch.Tensor] = None, contract: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """pts (N,S,6) [or (N,S,3) + ray_d (N,S,3)] -> ((N,S,4) = [rgb | raw density], normal (N,S,3))  (ref_model.py:68-106).
        ``contract`` (an addition; BASELINE configs[4] shapes): the positions go through the Mip-NeRF 360 scene contraction before the
        encoding -- a flag of the kernels' sample fetch; RefNeRF.get_grad then returns d density / d (uncontracted position)."""
        self._check_config()
        pos, d = pts[..., :3], (pts[..., 3:6] if ray_d is None else ray_d)
        prec = ops.current_precision()
        # ref_model.py:84-85, `spa_info_b + torch.normal(0, w, shape)` in training mode.  noise_rng (an addition, like render_image's rng):
        #   "philox" (default): N(0, w) deviates from Philox4x32-10 + Box-Muller keyed by (one 62-bit draw from torch's CPU generator per
        #       forward -- torch.manual_seed governs it -- or the device scalar `noise_seed_dev`, which nerf_amd.training.TrainStep points
        #       at its own per-step key so that a captured hipGraph replays fresh noise) and the sample index.  The training forward draws
        #       them INSIDE the kernel: no (M, 128) tensor is written and read back (1.6 GB + a 0.5 ms launch per 2^14-ray step);
        #   "torch": torch.normal on the device generator, as a tensor handed to the kernel (round 1-4 behaviour).
        noise, noise_kw = None, {}
        fused_train = (not self._generic()) and ab.needs_grad(pos, d, *self.parameters())
        if self.training and self.perturb_bottle_neck_w > 0:
            if getattr(self, "noise_rng", "philox") == "philox" and self.bottle_neck_dim == 128:
                seed_dev = self.__dict__.get("noise_seed_dev")
                if seed_dev is not None and seed_dev.device != pos.device:
                    seed_dev = None                          # a TrainStep's key on ANOTHER device (module moved since): host-drawn key instead
                seed = 0 if seed_dev is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
                if fused_train:
                    noise_kw = dict(noise_std=float(self.perturb_bottle_neck_w), noise_seed=seed, noise_seed_dev=seed_dev)
                else:
                    noise = ops.philox_normal(pos.numel() // 3, self.perturb_bottle_neck_w, seed, seed_dev, device=pos.device).view(pos.shape[:-1] + (128,))
            else:
                noise = torch.normal(0, self.perturb_bottle_neck_w, pos.shape[:-1] + (self.bottle_neck_dim,), device=pos.device)
        if self._generic():
            from . import generic_path
            return generic_path.ref_forward(self, pos, d, noise, contract=contract)
        named = list(self.named_parameters())
        params = [p for _, p in named]
        if ab.needs_grad(pos, d, *params):
            # training (train.py:176-186): HIP training forward (activation dump); the backward is hand-written kernels too --
            #   RefNeRF.get_grad (d density / d position, inputs_only mode): a dgrad-only chain to the encoded position + the encoding's
            #     derivative (nerf_amd_density_grad);
            #   loss.backward(): single-layer dgrad launches, the element-wise IDE / normal / head stage, MFMA weight gradients
            #     (nerf_amd_ref_backward).  Gradients w.r.t. the positions are only produced for get_grad (the reference discards the rest).
            names = [n for n, _ in named]
            held = {}
            shape = pos.shape[:-1]

            def hip(p, dd, *wb):
                # train.py:177 hands over the two halves of ONE (N, S, 6) sample tensor (`fine_samples.split((3, 3), dim=-1)`): when the
                # arguments are exactly those views the kernel reads the tensor they came from -- no cat, no copy
                if (p.dim() >= 2 and p.stride() == dd.stride() and p.stride(-1) == 1 and p.stride(-2) == 6 and p.shape == dd.shape and
                        p.untyped_storage().data_ptr() == dd.untyped_storage().data_ptr() and dd.storage_offset() == p.storage_offset() + 3 and
                        all(p.stride(k) == p.stride(k + 1) * p.shape[k + 1] for k in range(p.dim() - 2))):
                    pts6 = torch.as_strided(p, tuple(p.shape[:-1]) + (6,), p.stride(), p.storage_offset())
                else:
                    pts6 = torch.cat((p, dd), dim=-1).contiguous()
                rgbo, normal, held["dump"], held["aux"] = ops.ref_forward_train(self.packed(prec), prec, pts6, noise, self.kernel_flags, contract=contract, **noise_kw)
                held["pts"] = pts6.view(-1, 6)
                return rgbo, normal                           # two differentiable outputs (callers write into rgbo[..., -1] in place)

            def bwd(g, p, dd, *wb):
                if "dump" not in held:
                    raise RuntimeError("nerf_amd: the activation dump of this forward was already consumed (backward twice over the same graph)")
                g_rgbo, g_nrm = g
                if "bwd_blob" not in held:
                    held["bwd_blob"] = self.packed_backward(prec)
                if ab._VJP.inputs_only:                       # RefNeRF.get_grad: the density channel's gradient w.r.t. the positions
                    if g_rgbo is None:
                        return (torch.zeros_like(p), None, *[None] * len(wb))
                    gx = ops.density_grad(ops.NET_REF, held["bwd_blob"], prec, held["dump"], held["pts"], scale=g_rgbo.reshape(-1, 4)[:, 3], contract=contract)
                    return (gx.view(p.shape), None, *[None] * len(wb))
                M_ = held["pts"].shape[0]
                g2 = torch.cat((g_rgbo.reshape(-1, 4) if g_rgbo is not None else torch.zeros((M_, 4), dtype=torch.float32, device=p.device),
                                g_nrm.reshape(-1, 3) if g_nrm is not None else torch.zeros((M_, 3), dtype=torch.float32, device=p.device)), dim=-1)
                gw, gb = ops.ref_backward(held["bwd_blob"], prec, held.pop("dump"), held.pop("aux"), held["pts"][:, 3:], g2, self._ide_table(p.device), self.kernel_flags)
                by_name = self._grads_by_name(gw, gb)
                if self.sh_max_level != 4:
                    for k_ in ("dir_block1.0.weight", "dir_block2.0.weight"):
                        by_name[k_] = self._extract_dir(by_name[k_])
                for k_ in ("spa_block1.0.weight", "spa_block2.0.weight"):
                    by_name[k_] = self._extract_pos(by_name[k_])
                shapes = {n: p_.shape for n, p_ in named}                 # (narrow networks: the padded rows / columns are the discarded part)
                return (None, None, *[by_name[n][tuple(slice(0, k) for k in shapes[n])] if tuple(by_name[n].shape) != tuple(shapes[n]) else by_name[n]
                                      for n in names])
            rgbo, normal = ab.HipOp.apply(hip, bwd, 2, pos, d, *params)
            return rgbo, normal
        return ops.ref_forward(self.packed(prec), prec, torch.cat((pos, d), dim=-1).contiguous(), noise=noise, flags=self.kernel_flags, contract=contract)

    def _ide_table(self, device, level: int = 4):
        """the (2^(level-1) + 1, T) coefficient matrix of ref_func.py:60-74 on `device` (constant: float64 host loops, uploaded once)"""
        tables = self.__dict__.setdefault("_ide_table_on", {})
        key = device if level == 4 else (device, level)
        if key not in tables:
            tables[key] = ide_table(level).to(device).contiguous()
        return tables[key]

    def packed_backward(self, precision: int) -> torch.Tensor:
        ws, _ = self._pack_tensors()
        return ops.pack_weights_backward(self._net_id, precision, ws)

    @staticmethod
    def _grads_by_name(gw, gb):
        """kernel tensor order (include/nerf_amd.h, nerf_amd_ref_backward) -> parameter names"""
        out = {}
        for i, l in enumerate((0, 2, 4, 6)):
            out["spa_block1.%d.weight" % l], out["spa_block1.%d.bias" % l] = gw[i], gb[i]
            out["spa_block2.%d.weight" % l], out["spa_block2.%d.bias" % l] = gw[4 + i], gb[4 + i]
            out["dir_block1.%d.weight" % l], out["dir_block1.%d.bias" % l] = gw[11 + i], gb[11 + i]
            out["dir_block2.%d.weight" % l], out["dir_block2.%d.bias" % l] = gw[15 + i], gb[15 + i]
        out["bottle_neck.weight"], out["bottle_neck.bias"] = gw[8], gb[8]
        out["norm_col_tint_head.weight"], out["norm_col_tint_head.bias"] = gw[9], gb[9]
        out["rho_tau_head.weight"], out["rho_tau_head.bias"] = gw[10], gb[10]
        out["spec_rgb_head.0.weight"], out["spec_rgb_head.0.bias"] = gw[19], gb[19]
        return out

    @staticmethod
    def coarse_grad_select(fine_grads: torch.Tensor, sort_inds: torch.Tensor, c_pnum: int) -> torch.Tensor:
        """Pick the gradients that belong to the coarse samples after the coarse/fine merge sort (ref_model.py:108-117)."""
        if fine_grads.is_cuda and not fine_grads.requires_grad:      # one launch (nerf_amd_coarse_grad_select): ranks by ballot, no sort
            return ops.coarse_grad_select(fine_grads, sort_inds, c_pnum)
        n, total, _ = fine_grads.shape
        sel = torch.cat((torch.zeros((n, total - c_pnum), dtype=torch.bool, device=fine_grads.device),
                         torch.ones((n, c_pnum), dtype=torch.bool, device=fine_grads.device)), dim=-1)
        sel = torch.gather(sel, -1, sort_inds)
        # = fine_grads[sel].reshape(n, c_pnum, -1) without the boolean-mask gather (whose output size is data dependent: a device
        # synchronisation, and not capturable in a hipGraph): every row has exactly c_pnum selected entries, in sorted order
        pos = torch.sort(sel.to(torch.int8), dim=-1, descending=True, stable=True)[1][:, :c_pnum]
        return torch.gather(fine_grads, 1, pos[:, :, None].expand(-1, -1, fine_grads.shape[-1]))

    @staticmethod
    def get_grad(func_val: torch.Tensor, inputs: torch.Tensor) -> torch.Tensor:
        """Normalised d(func)/d(inputs) (ref_model.py:119-125): first-order only, like the reference (no create_graph).  Works for
        every differentiable op of this package -- e.g. the proposal density w.r.t. its sample positions (train.py:165-168,
        `prop_normal`), whose position gradient is the dgrad-only density chain (nerf_amd_density_grad), like RefNeRF's own."""
        with ab.inputs_only_grad():                              # no parameter gradient is needed for d(func)/d(inputs)
            grad, = torch.autograd.grad(func_val, inputs, torch.ones_like(func_val), retain_graph=True)
        grad_norm = grad.norm(dim=-1, keepdim=True)
        return grad / torch.maximum(torch.full_like(grad_norm, 1e-5), grad_norm)


def _dot_loss(weight: torch.Tensor, a: torch.Tensor, b: torch.Tensor, mode: int, scale: float, expr) -> torch.Tensor:
    """scale * sum w f(<a, b>) on the device: one streaming kernel + a fixed-order sum forward, one kernel backward
    (nerf_amd_weighted_dot_loss[_backward]) instead of five element-wise launches each way; `expr` = the reference's torch expression, the
    specification (and what CPU tensors -- host-side unit tests -- evaluate)."""
    if not weight.is_cuda:
        return expr(weight, a, b)
    if not ab.needs_grad(weight, a, b):
        return ops.weighted_dot_loss(weight, a, b, mode, scale)
    need = (weight.requires_grad, a.requires_grad, b.requires_grad)

    def bwd(g, w_, a_, b_):
        d_w, d_a, d_b = ops.weighted_dot_loss_backward(g, w_, a_, b_, mode, scale, need)
        return (d_w.view(w_.shape) if d_w is not None else None, d_a.view(a_.shape) if d_a is not None else None,
                d_b.view(b_.shape) if d_b is not None else None)
    return ab.HipOp.apply(lambda w_, a_, b_: ops.weighted_dot_loss(w_, a_, b_, mode, scale), bwd, 1, weight, a, b)


class WeightedNormalLoss(nn.Module):
    def __init__(self, size_average=False):
        super().__init__()
        self.size_average = size_average

    def forward(self, weight: torch.Tensor, d_norm: torch.Tensor, p_norm: torch.Tensor) -> torch.Tensor:
        """sum / mean of w (1 - <n_density, n_pred>)  (ref_model.py:127-135)."""
        def expr(w, d, p):
            diff = 1. - torch.sum(d * p, dim=-1)
            return torch.mean(w * diff) if self.size_average else torch.sum(w * diff)
        return _dot_loss(weight, d_norm, p_norm, 0, 1.0 / max(weight.numel(), 1) if self.size_average else 1.0, expr)


class BackFaceLoss(nn.Module):
    def forward(self, weight: torch.Tensor, normal: torch.Tensor, ray_d: torch.Tensor) -> torch.Tensor:
        """mean of w relu(<n, d>)  (ref_model.py:137-143)."""
        return _dot_loss(weight, normal, ray_d, 1, 1.0 / max(weight.numel(), 1), lambda w, n, d: torch.mean(w * F.relu(torch.sum(n * d, dim=-1))))
