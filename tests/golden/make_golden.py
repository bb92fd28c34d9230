#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REAL reference (Enigmatisms/NeRF,
mounted read-only at /root/reference) on CPU.  Run in the build container only:

    python tests/golden/make_golden.py

The reference never travels to the GPU box; only the .npz files written here do.  They hold
inputs + the reference's outputs (data, not source).  Network weights are produced by the
closed-form generator ``tests/weights.py`` (integer hash, libm-free) and loaded into the reference
modules with ``load_state_dict``, so they are not stored.

Import shims (in-process, no reference file is edited):
  1. ``torch.Tensor.cuda`` / ``nn.Module.cuda`` -> identity (the reference hard-codes ``.cuda()``);
  2. ``numpy.math = math`` (ref_func.py uses ``np.math.factorial``);
  3. stub modules for torchvision / natsort / tensorboard (imported at module top, never run).
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"


def install_shims():
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    np.math = math
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional",
                 "torchvision.utils", "natsort", "tensorboard", "torch.utils.tensorboard"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    sys.modules["torchvision.transforms.functional"].resize = None
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision.utils"].save_image = None
    sys.modules["natsort"].natsorted = sorted
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    sys.path.insert(0, REF)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def write_signatures():
    """G20: inspect.signature of every public function / class / method of the reference modules the entry scripts import
    (train.py:12-22,77,82; ddp_train.py:17-27; model_average.py:16-29) -- the call-surface contract of SURVEY.md section 8b as data."""
    import importlib
    import json
    import sigtools
    rec = {}
    for m in sigtools.MODULES:
        rec[m] = sigtools.module_signatures(importlib.import_module("nerf." + m), m)
    with open(os.path.join(HERE, "g20_signatures.json"), "w") as f:
        json.dump(rec, f, indent=0, sort_keys=True)
    print("wrote g20_signatures.json (%d names)" % sum(len(v) for v in rec.values()))


def write_entry_imports():
    """G23: the names the reference's entry scripts import from `nerf.*` (train.py:12-20, ddp_train.py:17-25, model_average.py:16-27),
    `*` resolved to the public names the star import binds -- the list a `nerf` stand-in package must serve (SURVEY.md section 8b:
    "importable as nerf.*").  Data (module and attribute names), no source text."""
    import importlib
    import json
    import sigtools
    rec = {}
    for script in sigtools.ENTRY_SCRIPTS:
        imp = sigtools.entry_imports(os.path.join(REF, script))
        for m, names in imp.items():
            if "*" in names:
                mod = importlib.import_module("nerf." + m)
                public = getattr(mod, "__all__", None) or [n for n, o in vars(mod).items()
                                                           if not n.startswith("_") and getattr(o, "__module__", None) == mod.__name__]
                imp[m] = sorted(set(n for n in names if n != "*") | set(public))
        rec[script] = imp
    with open(os.path.join(HERE, "g23_entry_imports.json"), "w") as f:
        json.dump(rec, f, indent=0, sort_keys=True)
    print("wrote g23_entry_imports.json (%d names)" % sum(len(v) for s in rec.values() for v in s.values()))


def write_config0_train_step():
    """G24: BASELINE configs[0] at ITS shape -- one training iteration of train.py:151-218 (non-ref) on a 200 x 200 image, 256 rays, 32 + 64
    samples, run by the REAL reference: randomFromOneImage -> validSampler -> ProposalNetwork -> get_weights -> maxBlurFilter ->
    inverseSample -> MipNeRF -> NeRF.render -> getBounds / ProposalLoss / MSE -> backward -> Adam step at the reference's learning-rate
    rule (train.py:56: lr * sample_ray_num / 512; scheduler value of iteration 0, nerf_base.py:115-134).  Stored: the draws (ray indices,
    stratified and inverse-CDF uniforms), the step's outputs, slices of the parameter gradients and of the parameters AFTER the step."""
    from nerf import nerf_base, mip_model, addtional, utils, mip_methods
    import torch.nn.functional as F
    import weights as W
    torch.set_num_threads(8)                                 # (like main(): the summation order of the CPU GEMMs is part of the fixture)
    near, far, N, C_, F_ = 2.0, 6.0, 256, 32, 64
    pose = utils.pose_spherical(52.0, -30.0, 4.0)[:3]
    focal = utils.fov2Focal(0.6911112070083618, (200, 200))
    img = torch.rand(3, 200, 200, generator=torch.Generator().manual_seed(24))
    prop = addtional.ProposalNetwork(10, 256); prop.load_state_dict(W.proposal_state("small")); prop.train()
    mip = mip_model.MipNeRF(10, 4, 256); mip.load_state_dict(W.mip_state("small")); mip.train()
    lr = 5e-4 * N / 512
    opt = torch.optim.Adam(list(mip.parameters()) + list(prop.parameters()), lr=lr, betas=(0.9, 0.999))
    sch = nerf_base.DecayLrScheduler(0.01, 0.1, 100000, lr, 500)
    pix, coords = utils.randomFromOneImage(img, (1.0, 1.0))
    torch.manual_seed(240)
    pts_c, len_c, rgb_tgt, rays_c = utils.validSampler(pix, coords, pose, N, C_, focal, near, far, True)
    torch.manual_seed(240)
    idx = torch.randint(0, coords.shape[0], (N,)); u_strat = torch.rand(N, C_)
    density = F.softplus(prop.forward(pts_c))
    pw = mip_methods.maxBlurFilter(addtional.ProposalNetwork.get_weights(density, len_c, rays_c[:, 3:]), 0.01)
    torch.manual_seed(241)
    fl, below = utils.inverseSample(pw, len_c, F_ + 1, sort=True)
    torch.manual_seed(241)
    u_inv = torch.rand(N, F_ + 1)
    fl = fl[..., :-1]
    rgbo = mip.forward(nerf_base.NeRF.length2pts(rays_c, fl))
    rend, wts, _ = nerf_base.NeRF.render(rgbo, fl, rays_c[:, 3:])
    img_loss = torch.nn.MSELoss()(rend, rgb_tgt)
    p_loss = addtional.ProposalLoss()(addtional.getBounds(pw, below), wts.detach())
    opt.zero_grad()
    (p_loss + img_loss).backward()
    _, lr0 = sch.update_opt_lr(0, opt)
    grads = {"g_mip_l1": mip.lin_block1[0].weight.grad[:8].clone(), "g_mip_skip": mip.lin_block2[0].weight.grad[:8].clone(),
             "g_mip_rgb": mip.rgb_layer[2].weight.grad.clone(), "g_mip_sigma": mip.opacity_head[0].weight.grad.clone(),
             "g_prop_l0": prop.layers[0].weight.grad[:8].clone(), "g_prop_head": prop.layers[8].weight.grad.clone()}
    opt.step()
    npz("g24_config0_train_step", pose=pose, focal=np.array(focal), img=img, idx=idx, u_strat=u_strat, u_inv=u_inv, rays=rays_c, rgb_tgt=rgb_tgt,
        z_coarse=len_c, z_fine=fl, below=below, rendered=rend, weights=wts, img_loss=img_loss, prop_loss=p_loss, lr=np.float64(lr0),
        p_mip_rgb_after=mip.rgb_layer[2].weight.detach(), p_mip_l1_after=mip.lin_block1[0].weight.detach()[:8],
        p_prop_head_after=prop.layers[8].weight.detach(), p_prop_l0_after=prop.layers[0].weight.detach()[:8], **grads)


def write_shallow_encodings():
    """G21: the reference's three networks built with FEWER encoding octaves and / or cat_origin=False (constructor arguments,
    mip_model.py:15-18, addtional.py:61, ref_model.py:17-24) -- forward values and (the first 8 rows of) parameter gradients of sum(out * G) from the REAL modules.
    States: oracle.init_linear_params on the oracle's shape tables (deterministic), so the test can rebuild them."""
    from nerf import mip_model, addtional, ref_model
    from oracle import nerf_oracle as O
    g = torch.Generator().manual_seed(2121)
    pts = torch.cat((torch.rand(6, 9, 3, generator=g) * 3 - 1.5, torch.randn(6, 9, 3, generator=g)), dim=-1)
    pts[..., 3:] = pts[..., 3:] / pts[..., 3:].norm(dim=-1, keepdim=True)
    G4, G1, G3 = torch.randn(6, 9, 4, generator=g), torch.randn(6, 9, generator=g), torch.randn(6, 9, 3, generator=g)
    out = {"pts": pts, "G4": G4, "G1": G1, "G3": G3}
    for L, cat, width in ((6, True, 256), (10, False, 256), (4, False, 96)):
        tag = "L%d_%s_w%d" % (L, "cat" if cat else "nocat", width)
        seed = 100 * L + width + int(cat)
        mip = mip_model.MipNeRF(L, 4, hidden_unit=width, cat_origin=cat)
        mip.load_state_dict(O.init_linear_params(O.mip_shapes(L, 4, width, cat), seed, std=0.08, bias_std=0.05))
        y = mip.forward(pts)
        (y * G4).sum().backward()
        out[tag + "_mip"], out[tag + "_mip_g0"], out[tag + "_mip_gskip"], out[tag + "_mip_grgb"] = (
            y, mip.lin_block1[0].weight.grad[:8], mip.lin_block2[0].weight.grad[:8], mip.rgb_layer[0].weight.grad[:8])
        prop = addtional.ProposalNetwork(L, hidden_unit=width, cat_origin=cat)
        prop.load_state_dict(O.init_linear_params(O.proposal_shapes(L, width, cat), seed + 1, std=0.08, bias_std=0.05))
        d = prop.forward(pts[..., :3])
        (d * G1).sum().backward()
        out[tag + "_prop"], out[tag + "_prop_g0"] = d, prop.layers[0].weight.grad[:8]
        ref = ref_model.RefNeRF(L, 4, hidden_unit=width, output_dim=width, cat_origin=cat); ref.eval()
        ref.load_state_dict(O.init_linear_params(O.ref_shapes(L, 4, width, 128, width, cat), seed + 2, std=0.08, bias_std=0.05))
        rgbo, nrm = ref.forward(pts)
        ((rgbo * G4).sum() + (nrm * G3).sum()).backward()
        out[tag + "_ref_rgbo"], out[tag + "_ref_normal"], out[tag + "_ref_g0"], out[tag + "_ref_gskip"] = (
            rgbo, nrm, ref.spa_block1[0].weight.grad[:8], ref.spa_block2[0].weight.grad[:8])
    npz("g21_shallow_encodings", **out)


def write_generic_refnerf():
    """G22: the reference's RefNeRF in shapes the fused HIP kernel is not compiled for -- `--ide_level 5` (procedures.py:211: 36 spherical-harmonic
    terms), hidden width 320 (`--nerf_net_width`, train.py:80), 11 position octaves with use_srgb -- forward values (rgbo, normal), the first 8 rows of
    four parameter gradients of sum(rgbo * G4) + sum(normal * G3), and RefNeRF.get_grad of the density w.r.t. the positions (ref_model.py:119-125),
    from the REAL modules.  States: oracle.init_linear_params on the oracle's shape table (deterministic), so the test can rebuild them."""
    from nerf import ref_model
    from oracle import nerf_oracle as O
    g = torch.Generator().manual_seed(2222)
    pos = torch.rand(5, 7, 3, generator=g) * 3 - 1.5
    dirs = torch.randn(5, 7, 3, generator=g)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True) * (0.6 + 0.8 * torch.rand(5, 7, 1, generator=g))      # (ray directions are not unit vectors, utils.py:78-85)
    G4, G3 = torch.randn(5, 7, 4, generator=g), torch.randn(5, 7, 3, generator=g)
    out = {"pos": pos, "dirs": dirs, "G4": G4, "G3": G3}
    for L, deg, width, srgb in ((10, 5, 256, False), (10, 4, 320, False), (11, 3, 288, True)):
        tag = "L%d_d%d_w%d" % (L, deg, width)
        ref = ref_model.RefNeRF(L, deg, hidden_unit=width, output_dim=width, use_srgb=srgb); ref.eval()
        ref.load_state_dict(O.init_linear_params(O.ref_shapes(L, deg, width, 128, width, True), 3000 + 10 * L + deg + width, std=0.07, bias_std=0.05))
        p = pos.clone().requires_grad_(True)
        rgbo, nrm = ref.forward(p, dirs)
        grad = ref_model.RefNeRF.get_grad(rgbo[..., -1], p)
        ((rgbo * G4).sum() + (nrm * G3).sum()).backward()
        out[tag + "_rgbo"], out[tag + "_normal"], out[tag + "_density_grad"] = rgbo, nrm, grad
        out[tag + "_g_spa0"], out[tag + "_g_dir0"], out[tag + "_g_dirskip"], out[tag + "_g_heads"] = (
            ref.spa_block1[0].weight.grad[:8], ref.dir_block1[0].weight.grad[:8], ref.dir_block2[0].weight.grad[:8], ref.norm_col_tint_head.weight.grad)
        out[tag + "_g_rho_tau"], out[tag + "_g_bottle"] = ref.rho_tau_head.weight.grad, ref.bottle_neck.weight.grad[:8]
    npz("g22_generic_refnerf", **out)


def main():
    install_shims()
    if "--signatures-only" in sys.argv:
        write_entry_imports()
        return write_signatures()
    if "--config0-only" in sys.argv:
        return write_config0_train_step()
    if "--shallow-only" in sys.argv:
        return write_shallow_encodings()
    if "--generic-ref-only" in sys.argv:
        return write_generic_refnerf()
    from nerf import nerf_helper, nerf_base, mip_methods, mip_model, addtional, utils, procedures
    import weights as W

    torch.set_num_threads(8)
    T = torch.tensor
    near, far = 2.0, 6.0

    # ---------------- G1 ray generation (+ fov2Focal, pose_spherical) ----------------
    pose = utils.pose_spherical(37.0, -30.0, 4.0)[:3]
    fs = utils.fov2Focal(0.6911112070083618, (100, 100))
    ft = utils.fov2Focal((0.6911112070083618, 0.5), (60, 100))
    ft_img = utils.fov2Focal((0.6911112070083618, 0.5), (100, 150))
    # render_image's prologue (procedures.py:43-51) is not a separate function, so the image-path ray table is
    # captured from the REAL render_image as a black box: NeRF.length2pts receives ``camera_rays`` of every tile.
    def ref_ray_raw(pose, image_size, focal):
        rec = []
        orig = nerf_base.NeRF.length2pts
        def spy(rays, z):
            rec.append(rays.clone())
            return orig(rays, z)
        nerf_base.NeRF.length2pts = staticmethod(spy)
        try:
            prop = addtional.ProposalNetwork(10, 256); mip = mip_model.MipNeRF(10, 4, 256)
            with torch.no_grad():
                procedures.render_image(mip, prop, pose, image_size, focal, near, far, 2)
        finally:
            nerf_base.NeRF.length2pts = staticmethod(orig)
        H, Wd = image_size
        sz, (pr, pc) = procedures.get_patch_size(image_size)
        out = torch.zeros(H, Wd, 3)
        for t, rays in enumerate(rec):
            k, j = divmod(t, pc)
            assert torch.equal(rays[:, :3], pose[:, -1].expand(sz * sz, -1))
            out[sz * k: sz * (k + 1), sz * j: sz * (j + 1)] = rays[:, 3:].view(sz, sz, 3)
        return out
    img = torch.rand(3, 60, 100, generator=torch.Generator().manual_seed(3))
    pix, coords = utils.randomFromOneImage(img, (1.0, 1.0))
    pixc, coordsc = utils.randomFromOneImage(img, (0.5, 0.5))
    torch.manual_seed(11)
    rgb_s, rays_s = utils.validSampler(pix, coords, pose, 16, 32, ft, near, far, False)
    torch.manual_seed(11)
    idx_s = torch.randint(0, coords.shape[0], (16,))
    npz("g01_raygen", pose=pose, focal_sq=np.array(fs), focal_tuple=np.array(ft),
        pose_full=utils.pose_spherical(37.0, -30.0, 4.0),
        ray_raw_sq=ref_ray_raw(pose, (100, 100), fs), ray_raw_scalar=ref_ray_raw(pose, (100, 100), 138.5),
        focal_tuple_img=np.array(ft_img), ray_raw_tuple=ref_ray_raw(pose, (100, 150), ft_img),
        img=img, pix=pix, coords=coords, pix_crop=pixc, coords_crop=coordsc,
        sampler_idx=idx_s, sampler_rgb=rgb_s, sampler_rays=rays_s)

    # ---------------- G2 stratified z / pts (train + render flavours) ----------------
    torch.manual_seed(5)
    pts_t, len_t, rgb_t, rays_t = utils.validSampler(pix, coords, pose, 8, 32, ft, near, far, True)
    torch.manual_seed(5)
    idx_t = torch.randint(0, coords.shape[0], (8,))
    u_t = torch.rand((8, 32))
    # render flavour (procedures.py:52,59,65-66): captured from the REAL render_image (one 50x50 tile) by spying
    # on ProposalNetwork.forward (receives pts) and ProposalNetwork.get_weights (receives sampled_lengths).
    cap = {}
    prop = addtional.ProposalNetwork(10, 256); mip = mip_model.MipNeRF(10, 4, 256)
    orig_fwd, orig_gw = addtional.ProposalNetwork.forward, addtional.ProposalNetwork.get_weights
    def spy_fwd(self, pts, encoded_pt=None):
        cap["pts"] = pts.clone(); return orig_fwd(self, pts, encoded_pt)
    def spy_gw(density, zvals, ray_dirs=None):
        cap["z"] = zvals.clone(); cap["dirs"] = ray_dirs.clone(); return orig_gw(density, zvals, ray_dirs)
    addtional.ProposalNetwork.forward = spy_fwd
    addtional.ProposalNetwork.get_weights = staticmethod(spy_gw)
    try:
        f50 = utils.fov2Focal(0.6911112070083618, (50, 50))
        torch.manual_seed(6)
        with torch.no_grad():
            procedures.render_image(mip, prop, pose, 50, f50, near, far, 96)
    finally:
        addtional.ProposalNetwork.forward = orig_fwd
        addtional.ProposalNetwork.get_weights = staticmethod(orig_gw)
    torch.manual_seed(6)
    u_r = torch.rand((50, 50, 64)).view(-1, 64)                      # the tile's first draw (procedures.py:65)
    sel = torch.arange(0, 2500, 97)
    npz("g02_stratified", idx=idx_t, u_train=u_t, z_train=len_t, pts_train=pts_t, rays_train=rays_t, rgb_train=rgb_t,
        render_sample_num=np.array(96), render_focal=np.array(f50), u_render=u_r[sel], z_render=cap["z"][sel],
        pts_render=cap["pts"][sel], dirs_render=cap["dirs"][sel], origin=pose[:, -1])

    # ---------------- G3 positional encoding ----------------
    x = (torch.rand(6, 5, 3, generator=torch.Generator().manual_seed(7)) - 0.5) * 14.0
    x[0, 0] = T([6.9, -6.9, 0.0]); x[0, 1] = T([1e-3, -1e-3, 3.14159265])
    npz("g03_pe", x=x, pe10=nerf_helper.positional_encoding(x, 10), pe4=nerf_helper.positional_encoding(x, 4),
        x2d=x.reshape(-1, 3), pe4_2d=nerf_helper.positional_encoding(x.reshape(-1, 3), 4))

    # ---------------- G4 / G9 the two MLPs ----------------
    g = torch.Generator().manual_seed(9)
    rays = torch.cat((pose[:, -1].expand(12, -1), ref_ray_raw(pose, (100, 100), fs)[::9, ::8].reshape(-1, 3)[:12]), -1)
    out = {}
    for tag in ("small", "he"):
        prop = addtional.ProposalNetwork(10, 256); prop.load_state_dict(W.proposal_state(tag)); prop.eval()
        mip = mip_model.MipNeRF(10, 4, 256); mip.load_state_dict(W.mip_state(tag)); mip.eval()
        zc = torch.linspace(near, far, 64) + torch.rand(12, 64, generator=g) * ((far - near) / 128)
        pts_c = rays[:, None, :3] + zc[..., None] * rays[:, None, 3:]
        zf, _ = torch.sort(near + (far - near) * torch.rand(12, 40, generator=g), dim=-1)
        pts_f = nerf_base.NeRF.length2pts(rays, zf)
        with torch.no_grad():
            out[tag + "_density"] = prop.forward(pts_c)
            out[tag + "_rgbo"] = mip.forward(pts_f)
        out[tag + "_pts_c"] = pts_c; out[tag + "_pts_f"] = pts_f
    npz("g04_g09_mlp", rays=rays, **out)

    # ---------------- G5 sigma -> weights ----------------
    sig = torch.randn(10, 64, generator=g) * 3.0
    zz, _ = torch.sort(near + (far - near) * torch.rand(10, 64, generator=g), dim=-1)
    dd = torch.randn(10, 3, generator=g) * 0.7
    npz("g05_weights", sigma=sig, z=zz, dirs=dd,
        w_prop=addtional.ProposalNetwork.get_weights(sig, zz, dd),
        w_prop_nodir=addtional.ProposalNetwork.get_weights(sig, zz, None),
        w_nerf=nerf_base.NeRF.getNormedWeight(sig, zz),
        w_nerf_id=nerf_base.NeRF.getNormedWeight(sig, zz, lambda t: t.abs()))

    # ---------------- G6 max-blur ----------------
    wr = torch.rand(10, 64, generator=g) ** 4
    npz("g06_maxblur", w=wr, out=mip_methods.maxBlurFilter(wr, 0.01), out_a=mip_methods.maxBlurFilter(wr, 0.25))

    # ---------------- G7 inverse sampling (given u) ----------------
    wp = mip_methods.maxBlurFilter(addtional.ProposalNetwork.get_weights(sig, zz, dd), 0.01)
    wp[3] = 0.0; wp[3, 20] = 1.0                                   # a delta: exercises denom < 1e-5 and edge bins
    wp[4] = 0.01                                                    # flat pdf
    torch.manual_seed(21)
    zs_sorted, below_sorted = utils.inverseSample(wp, zz, 129, sort=True)
    torch.manual_seed(21)
    zs_raw = utils.inverseSample(wp, zz, 129, sort=False)
    torch.manual_seed(21)
    u_inv = torch.rand(10, 129)
    torch.manual_seed(22)
    mids = 0.5 * (zz[..., 1:] + zz[..., :-1])
    s_pdf, b_pdf, a_pdf = utils.sample_pdf(mids, wp[..., 1:-1], 33)
    torch.manual_seed(22)
    u_pdf = torch.rand(10, 33)
    npz("g07_inverse", w=wp, z=zz, u=u_inv, z_sorted=zs_sorted, below_sorted=below_sorted, z_raw=zs_raw,
        u_pdf=u_pdf, s_pdf=s_pdf, below_pdf=b_pdf, above_pdf=a_pdf)

    # ---------------- G8 sample assembly ----------------
    zc8 = zz[:, :16].contiguous()
    zf8 = zs_sorted[:, :33].contiguous()
    fi8 = below_sorted[:, :33].contiguous()
    r8 = torch.cat((torch.randn(10, 3, generator=g), dd), -1)
    m2 = nerf_base.NeRF.coarseFineMerge(r8, zc8, zf8)
    m4 = nerf_base.NeRF.coarseFineMerge(r8, zc8, zf8, fi8)
    npz("g08_assembly", rays=r8, zc=zc8, zf=zf8, finds=fi8, l2p=nerf_base.NeRF.length2pts(r8, zf8),
        m2_samples=m2[0], m2_z=m2[1], m4_samples=m4[0], m4_z=m4[1], m4_inds=m4[2], m4_sort=m4[3])

    # ---------------- G10 compositing ----------------
    rgbo = torch.cat((torch.rand(10, 48, 3, generator=g), torch.randn(10, 48, 1, generator=g) * 4), -1)
    z10, _ = torch.sort(near + (far - near) * torch.rand(10, 48, generator=g), dim=-1)
    nrm = torch.randn(10, 48, 3, generator=g)
    camz = pose[:, -2]
    o = {}
    for wb in (False, True):
        for mn in (False, True):
            rgb, w, ex = nerf_base.NeRF.render(rgbo, z10, dd, mul_norm=mn, white_bkg=wb, render_depth=(near, far),
                                               normal_info=(nrm, camz))
            k = "wb%d_mn%d_" % (wb, mn)
            o[k + "rgb"], o[k + "w"], o[k + "depth"], o[k + "normal"] = rgb, w, ex["depth_img"], ex["normal_img"]
    rgb, w, ex = nerf_base.NeRF.render(rgbo, z10, dd, density_act=torch.nn.functional.softplus)
    o["softplus_rgb"], o["softplus_w"] = rgb, w
    npz("g10_composite", rgbo=rgbo, z=z10, dirs=dd, normal=nrm, cam_z=camz, **o)

    # ---------------- G11 render_image end-to-end (reference RNG order) ----------------
    o = {}
    for tag, size, sn in (("small_50", 50, 128), ("he_50", 50, 128), ("small_100", 100, 64), ("small_200", 200, 64)):
        wt = tag.split("_")[0]
        prop = addtional.ProposalNetwork(10, 256); prop.load_state_dict(W.proposal_state(wt)); prop.eval()
        mip = mip_model.MipNeRF(10, 4, 256); mip.load_state_dict(W.mip_state(wt)); mip.eval()
        f = utils.fov2Focal(0.6911112070083618, (size, size))
        torch.manual_seed(1234)
        with torch.no_grad():
            res = procedures.render_image(mip, prop, pose, size, f, near, far, sn, white_bkg=True, render_depth=True)
        o[tag + "_rgb"] = res["rgb"]; o[tag + "_depth"] = res["depth_img"][0]
        o[tag + "_focal"] = np.array(f)
    npz("g11_render_image", pose=pose, **o)

    # ---------------- G12 integrated PE (dead code) ----------------
    z12, _ = torch.sort(near + (far - near) * torch.rand(6, 9, generator=g), dim=-1)
    r12 = torch.cat((torch.randn(6, 3, generator=g), torch.randn(6, 3, generator=g)), -1)
    feat, mu, mu_t = mip_methods.ipe_feature(z12, r12, 6, 0.0015)
    npz("g12_ipe", z=z12, rays=r12, feat=feat, mu=mu, mu_t=mu_t)

    # ---------------- G13 IDE + RefNeRF eval forward + render_image with a RefNeRF ----------------
    from nerf import ref_model, ref_func
    ide_fn = ref_func.generate_ide_fn(4)
    dirs13 = torch.randn(9, 7, 3, generator=g); dirs13 = dirs13 / dirs13.norm(dim=-1, keepdim=True) * (0.5 + torch.rand(9, 7, 1, generator=g))
    rho13 = torch.rand(9, 7, 1, generator=g) * 0.95 + 0.05
    o13 = {"ide_dirs": dirs13, "ide_rho": rho13, "ide": ide_fn(dirs13, rho13)}
    for tag in ("small", "he"):
        net = ref_model.RefNeRF(10, 4); net.load_state_dict(W.ref_state(tag)); net.eval()
        zf13, _ = torch.sort(near + (far - near) * torch.rand(12, 24, generator=g), dim=-1)
        pts13 = nerf_base.NeRF.length2pts(rays, zf13)
        with torch.no_grad():
            rgbo13, nrm13 = net.forward(pts13)
        o13[tag + "_pts"], o13[tag + "_rgbo"], o13[tag + "_normal"] = pts13, rgbo13, nrm13
    prop = addtional.ProposalNetwork(10, 256); prop.load_state_dict(W.proposal_state("small")); prop.eval()
    net = ref_model.RefNeRF(10, 4); net.load_state_dict(W.ref_state("small")); net.eval()
    f50 = utils.fov2Focal(0.6911112070083618, (50, 50))
    torch.manual_seed(4321)
    with torch.no_grad():
        res13 = procedures.render_image(net, prop, pose, 50, f50, near, far, 64, white_bkg=True, render_depth=True, render_normal=True)
    o13["img_rgb"], o13["img_depth"], o13["img_normal"], o13["img_focal"] = res13["rgb"], res13["depth_img"][0], res13["normal_img"][0], np.array(f50)
    npz("g13_refnerf", rays=rays, pose=pose, **o13)
    abi_ref = [[k, list(v.shape)] for k, v in ref_model.RefNeRF(10, 4).state_dict().items()]

    # ---------------- G14 train-step losses + parameter grads (non-ref) ----------------
    prop = addtional.ProposalNetwork(10, 256); prop.load_state_dict(W.proposal_state("small"))
    mip = mip_model.MipNeRF(10, 4, 256); mip.load_state_dict(W.mip_state("small"))
    torch.manual_seed(77)
    pts_c, len_c, rgb_tgt, rays_c = utils.validSampler(pix, coords, pose, 32, 32, ft, near, far, True)
    torch.manual_seed(77)
    idx14 = torch.randint(0, coords.shape[0], (32,)); u14 = torch.rand(32, 32)
    import torch.nn.functional as F
    density = F.softplus(prop.forward(pts_c))                                         # train.py:166-169
    pw_raw = addtional.ProposalNetwork.get_weights(density, len_c, rays_c[:, 3:])
    pw = mip_methods.maxBlurFilter(pw_raw, 0.01)
    torch.manual_seed(78)
    fl, below = utils.inverseSample(pw, len_c, 65, sort=True)
    torch.manual_seed(78)
    u14b = torch.rand(32, 65)
    fl = fl[..., :-1]
    rgbo14 = mip.forward(nerf_base.NeRF.length2pts(rays_c, fl))
    rend, wts, _ = nerf_base.NeRF.render(rgbo14, fl, rays_c[:, 3:])
    bounds = addtional.getBounds(pw, below)
    img_loss = torch.nn.MSELoss()(rend, rgb_tgt)
    p_loss = addtional.ProposalLoss()(bounds, wts.detach())
    (p_loss + img_loss).backward()
    npz("g14_train_step", idx=idx14, u_strat=u14, u_inv=u14b, rays=rays_c, rgb_tgt=rgb_tgt, z_coarse=len_c,
        z_fine=fl, below=below, bounds=bounds, rendered=rend, weights=wts, img_loss=img_loss, prop_loss=p_loss,
        psnr=addtional.LossPSNR()(img_loss),
        g_mip_l1=mip.lin_block1[0].weight.grad[:8, :], g_mip_rgb=mip.rgb_layer[2].weight.grad,
        g_mip_sigma=mip.opacity_head[0].weight.grad, g_prop_l0=prop.layers[0].weight.grad[:8, :],
        g_prop_head=prop.layers[8].weight.grad)

    # ---------------- G16 state_dict ABI (key names + shapes of the reference modules) ----------------
    import json
    abi = {}
    for name, mod in (("mip", mip_model.MipNeRF(10, 4, 256)), ("prop", addtional.ProposalNetwork(10, 256)),
                      ("prop128", addtional.ProposalNetwork(10))):
        abi[name] = [[k, list(v.shape)] for k, v in mod.state_dict().items()]
    abi["ref"] = abi_ref
    with open(os.path.join(HERE, "g16_state_dict_abi.json"), "w") as f:
        json.dump(abi, f, indent=0)
    print("wrote g16_state_dict_abi.json")

    # ---------------- G15 LR schedule ----------------
    sch = nerf_base.DecayLrScheduler(0.01, 0.1, 100000, 3e-4, 500)
    steps = np.array([0, 1, 250, 499, 500, 501, 10000, 100500, 500000, 2000000])
    npz("g15_lr", steps=steps, lr=np.array([sch.update_opt_lr(int(s))[1] for s in steps]))

    # ---------------- G17 Ref-NeRF train step (train.py:164-199, ref branch, prop_normal on) ----------------
    # (kept LAST so that the generator stream of G1..G15 is unchanged; torch.normal is replaced by a recorded tensor)
    N17, C17, F17 = 16, 16, 32
    prop = addtional.ProposalNetwork(10, 256); prop.load_state_dict(W.proposal_state("small")); prop.train()
    net = ref_model.RefNeRF(10, 4); net.load_state_dict(W.ref_state("small")); net.train()
    rays17 = rays[:N17].clone() if rays.shape[0] >= N17 else torch.cat([rays] * 2)[:N17].clone()
    tgt17 = torch.rand(N17, 3, generator=g)
    res17 = (far - near) / C17
    zc17 = torch.linspace(near, far - res17, C17) + torch.rand(N17, C17, generator=g) * res17
    pts17 = (rays17[:, None, :3] + rays17[:, None, 3:] * zc17[:, :, None]).clone()
    u17 = torch.rand(N17, F17 + 1, generator=g)
    noise17 = torch.randn(N17, C17 + F17, 128, generator=g) * 0.1
    real_normal, real_rand = torch.normal, torch.rand
    torch.normal = lambda *a, **k: noise17
    torch.rand = lambda *a, **k: u17                                                    # the one draw inside sample_pdf
    try:
        pts17.requires_grad = True
        dens17 = prop.forward(pts17)
        coarse_grad = -ref_model.RefNeRF.get_grad(dens17, pts17)
        dens17 = F.softplus(dens17)
        pw17 = mip_methods.maxBlurFilter(addtional.ProposalNetwork.get_weights(dens17, zc17, rays17[:, 3:]), 0.01)
        fl17, below17 = utils.inverseSample(pw17, zc17, F17 + 1, sort=True)
        samples17, fl17m, below17m, sort17 = nerf_base.NeRF.coarseFineMerge(rays17, zc17, fl17, below17)
        pos17, dir17 = samples17.split((3, 3), dim=-1)
        pos17.requires_grad = True
        rgbo17, nrm17 = net.forward(pos17, dir17)
        dgrad17 = -ref_model.RefNeRF.get_grad(rgbo17[..., -1], pos17)
        rgbo17_raw = rgbo17.detach().clone()
        rgbo17[..., -1] = F.softplus(rgbo17[..., -1] + 0.5)
        rend17, wts17, _ = nerf_base.NeRF.render(rgbo17, fl17m, rays17[:, 3:], net.density_act)   # (the reference's positional quirk)
        nl17 = ref_model.WeightedNormalLoss()(wts17, dgrad17, nrm17)
        bf17 = ref_model.BackFaceLoss()(wts17, nrm17, dir17)
        cg17 = ref_model.RefNeRF.coarse_grad_select(dgrad17, sort17, C17)
        cnl17 = ref_model.WeightedNormalLoss()(pw17, cg17.detach(), coarse_grad)
        bounds17 = addtional.getBounds(pw17, below17m)
        img17 = torch.nn.MSELoss()(rend17, tgt17)
        pl17 = addtional.ProposalLoss()(bounds17, wts17.detach())
        loss17 = pl17 + img17 + 4e-4 * (nl17 + 0.1 * cnl17) + 0.1 * bf17
        loss17.backward()
    finally:
        torch.normal, torch.rand = real_normal, real_rand
    npz("g17_ref_train_step", rays=rays17, rgb_tgt=tgt17, z_coarse=zc17, u_inv=u17, noise=noise17, z_fine=fl17, z_merged=fl17m,
        below_merged=below17m, sort_ids=sort17, rgbo_raw=rgbo17_raw, pred_normal=nrm17, density_grad=dgrad17, coarse_grad=coarse_grad,
        weights=wts17, rendered=rend17, normal_loss=nl17, bf_loss=bf17, coarse_normal_loss=cnl17, img_loss=img17, prop_loss=pl17,
        loss=loss17, g_spa0=net.spa_block1[0].weight.grad[:8, :], g_rho_tau=net.rho_tau_head.weight.grad,
        g_nct=net.norm_col_tint_head.weight.grad, g_bottle=net.bottle_neck.weight.grad[:8, :], g_dir0=net.dir_block1[0].weight.grad[:8, :],
        g_spec=net.spec_rgb_head[0].weight.grad, g_prop_l0=prop.layers[0].weight.grad[:8, :], g_prop_head=prop.layers[8].weight.grad,
        g_spa2_6=net.spa_block2[6].weight.grad[:8, :], g_rho_tau_bias=net.rho_tau_head.bias.grad)

    # ---------------- G18 integrated PE at the network's size (L = 10) + coneParameters (mip_methods.py:15-58) ----------------
    # (after G17: draws from the end of the generator stream, earlier fixtures unchanged)
    z18, _ = torch.sort(near + (far - near) * torch.rand(48, 33, generator=g), dim=-1)
    o18 = torch.randn(48, 3, generator=g) * 0.5 + T([0.0, 0.0, 4.0])
    d18 = F.normalize(torch.randn(48, 3, generator=g) * 0.3 + T([0.0, 0.0, -1.0]), dim=-1) * (0.9 + 0.3 * torch.rand(48, 1, generator=g))
    r18 = torch.cat((o18, d18), -1)
    rad18 = 2.0 / (12.0 ** 0.5) / 1111.1
    feat18, mu18, mu_t18 = mip_methods.ipe_feature(z18, r18, 10, rad18)
    cp18 = mip_methods.coneParameters(z18, rad18)
    npz("g18_ipe_l10", z=z18, rays=r18, radius=np.float64(rad18), feat=feat18, mu=mu18, mu_t=mu_t18, var_t=cp18[1], var_r=cp18[2],
        dir_norm=r18[:, 3:].norm())

    # ---------------- G19 RefNeRF(use_srgb=True): forward + parameter gradients (ref_model.py:100-102, nerf_helper.py:50-56) ----------------
    # (after G18: draws from the end of the generator stream, earlier fixtures unchanged)
    zf19, _ = torch.sort(near + (far - near) * torch.rand(12, 20, generator=g), dim=-1)
    pts19 = nerf_base.NeRF.length2pts(rays, zf19)
    G19, Gn19 = torch.randn(12, 20, 4, generator=g), torch.randn(12, 20, 3, generator=g)
    o19 = {"pts": pts19, "g_rgbo": G19, "g_normal": Gn19}
    for tag in ("small", "he"):
        net = ref_model.RefNeRF(10, 4, use_srgb=True); net.load_state_dict(W.ref_state(tag)); net.eval()
        rgbo19, nrm19 = net.forward(pts19)
        ((rgbo19 * G19).sum() + (nrm19 * Gn19).sum()).backward()
        o19[tag + "_rgbo"], o19[tag + "_normal"] = rgbo19, nrm19
        o19[tag + "_g_spec"] = net.spec_rgb_head[0].weight.grad
        o19[tag + "_g_nct"] = net.norm_col_tint_head.weight.grad
        o19[tag + "_g_nct_bias"] = net.norm_col_tint_head.bias.grad
        o19[tag + "_g_rho_tau"] = net.rho_tau_head.weight.grad
        o19[tag + "_g_dir2_6"] = net.dir_block2[6].weight.grad[:8, :]
        o19[tag + "_g_spa2_6"] = net.spa_block2[6].weight.grad[:8, :]
        o19[tag + "_g_spa0"] = net.spa_block1[0].weight.grad[:8, :]
    npz("g19_refnerf_srgb", **o19)

    write_shallow_encodings()
    write_generic_refnerf()
    write_entry_imports()
    write_signatures()
    write_config0_train_step()


if __name__ == "__main__":
    main()
