"""Host-side mirror of the reference's ``nerf/procedures.py``: ``render_image`` with the reference's
signature, tile order, RNG protocol and return dict -- but the whole image goes through the HIP
pipeline in four launches instead of a Python loop over 2500-ray tiles."""
import argparse
from collections.abc import Iterable
from typing import Optional

import torch

from . import ops
from .addtional import ProposalNetwork
from .nerf_base import NeRF

POSSIBLE_PATCH_SIZE = [50, 40, 60, 30]
RENDER_COARSE_PNUM = 64


def get_patch_size(image_size):
    """First patch size in (50,40,60,30) that divides the width (procedures.py:24-31); None when there is
    none (the reference raises UnboundLocalError there; this build then renders the image un-tiled)."""
    for p in POSSIBLE_PATCH_SIZE:
        if image_size[1] % p == 0:
            return p, (image_size[0] // p, image_size[1] // p)
    return None, None


def _draw_uniforms(H, W, sample_num, sz, patch_num, device, rng):
    """Uniforms for every ray, in TILE order when tiling applies.
    rng='reference': per tile one (sz,sz,64) draw then one (sz*sz, sample_num+1) draw from the CPU default
    generator -- exactly the reference's order (procedures.py:65, utils.py:115), so a seeded run reproduces it;
    rng='device': two draws on the device generator (fast path, used by bench.py)."""
    if rng == "device":
        n = (patch_num[0] * patch_num[1] * sz * sz) if sz else H * W
        return (torch.rand((n, RENDER_COARSE_PNUM), device=device), torch.rand((n, sample_num + 1), device=device))
    if sz is None:
        return torch.rand((H * W, RENDER_COARSE_PNUM)).to(device), torch.rand((H * W, sample_num + 1)).to(device)
    n_tiles = patch_num[0] * patch_num[1]
    u1 = torch.empty((n_tiles, sz * sz, RENDER_COARSE_PNUM))
    u2 = torch.empty((n_tiles, sz * sz, sample_num + 1))
    for t in range(n_tiles):
        u1[t] = torch.rand((sz, sz, RENDER_COARSE_PNUM)).view(-1, RENDER_COARSE_PNUM)
        u2[t] = torch.rand((sz * sz, sample_num + 1))
    return u1.view(-1, RENDER_COARSE_PNUM).to(device), u2.view(-1, sample_num + 1).to(device)


# rays per chunk of the layer-by-layer route (the reference tiles an image the same way, procedures.py:53-56)
GENERIC_CHUNK_RAYS = 4096


def _render_rays_by_calls(network, prop_net, rays, z_base, u_strat, u_inv, sample_num, near, far, white_bkg, render_depth, chunk: Optional[int] = None,
                          is_ref_model: bool = False, cam_dir=None, seed: Optional[int] = None, ray_offset: int = 0, contract: bool = False,
                          ipe_radius: Optional[float] = None, ipe_dir_norm=None):
    """The tile body of procedures.py:62-85 as the reference writes it -- stratified depths, ProposalNetwork.forward, get_weights,
    maxBlurFilter, inverseSample, NeRF.length2pts, network.forward, NeRF.render -- on chunks of rays: the route of networks the fused
    render entry (nerf_amd_render_rays) has no packed layout for.  Every call is a HIP kernel of this package; uniforms that were not
    given are the render kernels' own Philox streams for `seed` and the chunk's global ray indices (ops.philox_stream): the image of a
    seeded render does not depend on whether a network runs fused or layer by layer."""
    from .mip_methods import maxBlurFilter
    from .utils import inverseSample
    N = rays.shape[0]
    chunk = GENERIC_CHUNK_RAYS if chunk is None else chunk
    rgb = torch.empty((N, 3), dtype=torch.float32, device=rays.device)
    depth = torch.empty((N,), dtype=torch.float32, device=rays.device) if render_depth else None
    normal_px = torch.empty((N,), dtype=torch.float32, device=rays.device) if cam_dir is not None else None
    resolution = (far - near) / sample_num                                           # procedures.py:57
    for s in range(0, N, chunk):
        r = rays[s: s + chunk].contiguous()
        n = r.shape[0]
        u1 = u_strat[s: s + n] if u_strat is not None else ops.philox_stream((n, RENDER_COARSE_PNUM), seed, ray_offset + s, strat=True, device=rays.device)
        u2 = u_inv[s: s + n] if u_inv is not None else ops.philox_stream((n, sample_num + 1), seed, ray_offset + s, device=rays.device)
        z, pts = ops.stratified_points(r, z_base, u1.contiguous(), resolution)      # :65-66
        density = prop_net.forward(pts, contract=True) if contract else prop_net.forward(pts)
        prop_w = maxBlurFilter(ProposalNetwork.get_weights(density, z, r[:, 3:]), 0.01)
        fine, _ = inverseSample(prop_w, z, sample_num + 1, sort=True, u=u2.contiguous())
        normal = None
        if is_ref_model:                                                             # :71-74
            samples, fine = NeRF.coarseFineMerge(r, z, fine)
            rgbo, normal = network.forward(samples, contract=True) if contract else network.forward(samples)
            rgbo[..., -1] = torch.nn.functional.softplus(rgbo[..., -1] + 0.5)
        elif ipe_radius is not None:                                                  # frusta between the sample_num + 1 sorted depths
            rgbo = network.forward_rays(r, fine.contiguous(), sample_num, ipe_radius=ipe_radius, ipe_dir_norm=ipe_dir_norm, contract=contract)
            fine = fine[..., :-1].contiguous()
        else:
            fine = fine[..., :-1].contiguous()
            rgbo = network.forward(NeRF.length2pts(r, fine), contract=True) if contract else network.forward(NeRF.length2pts(r, fine))
        part, _, extras = NeRF.render(rgbo, fine, r[:, 3:], white_bkg=white_bkg, density_act=torch.nn.functional.relu,
                                      render_depth=(near, far) if render_depth else None,
                                      normal_info=(normal, cam_dir) if cam_dir is not None else None)
        rgb[s: s + n] = part
        if render_depth:
            depth[s: s + n] = extras["depth_img"].reshape(-1)
        if cam_dir is not None:
            normal_px[s: s + n] = extras["normal_img"].reshape(-1)
    return rgb, depth, normal_px


def render_image(network: NeRF, prop_net: ProposalNetwork, render_pose: torch.Tensor, image_size, focal,
                 near: float, far: float, sample_num: int = 128, white_bkg: bool = False, render_depth=False,
                 render_normal=False, rng: str = "philox", contract: bool = False, ipe=False, seed: Optional[int] = None,
                 _shard=None) -> dict:
    """Whole-image inference (procedures.py:34-97) -> {"rgb" (3,H,W) [, "depth_img" (3,H,W)]} on
    ``render_pose.device``.  The caller provides ``no_grad``/``eval()`` like for the reference.
    ``rng``, ``contract`` and ``ipe`` are additions.  ``rng``: where the stratified / inverse-CDF uniforms come from --
      "philox" (default): drawn INSIDE the kernels, Philox4x32-10 keyed by one 62-bit seed taken from torch's CPU generator (so
                 ``torch.manual_seed`` makes a render reproducible); no uniform tensor exists, the drop-in call runs at the kernels' rate;
      "reference": the reference's own stream -- per tile one (sz,sz,64) then one (sz*sz, n+1) draw from the CPU default generator
                 (procedures.py:65, utils.py:115) copied to the device: a seeded run reproduces the reference's image bit for bit in
                 its uniforms (0.5 GB over PCIe per 800x800 image: ~1.4 M rays/s);
      "device":  two torch.rand draws on the device generator.
    ``seed`` (philox mode): the Philox key itself instead of a draw from the CPU generator.  ``_shard = (start, end)`` (internal:
    nerf_amd.parallel.render_image_sharded) renders only those rays of the image's TILE-ORDERED ray list -- every uniform is a pure
    function of (key, global index in that list), so shards reproduce the single-process image bit for bit -- and returns the per-ray
    outputs ``{"rgb_rays", "depth_rays", "normal_rays", "range", "to_image"}`` instead of images.
    ``contract=True`` applies the Mip-NeRF 360 scene contraction to every sample
    position before the networks encode it (unbounded scenes, BASELINE config 5; not available for Ref-NeRF).  ``ipe`` (BASELINE
    config 3): the fine network reads the integrated positional encoding of the conical frustum between consecutive fine depths
    (mip_methods.py:15-58: [mu | ipe_feature]) instead of the point encoding; True = the pixel radius 2/sqrt(12) pixel widths of
    Mip-NeRF, a float = that radius.  The reference holds `ipe_feature` but never calls it: its use inside the loop is this build's
    definition (oracle.render_rays(ipe_radius=...)), parity of the function itself is pinned by golden G12."""
    if not isinstance(image_size, Iterable):
        image_size = (image_size, image_size)
    is_ref_model = type(network).__name__ == "RefNeRF"
    render_normal = bool(render_normal) and is_ref_model
    H, W = int(image_size[0]), int(image_size[1])
    dev = render_pose.device
    if dev.type != "cuda":
        raise RuntimeError("nerf_amd.render_image: render_pose must live on the HIP device; there is no CPU path")
    if isinstance(focal, Iterable):
        fx, fy = float(focal[1]), float(focal[0])                                  # procedures.py:45-47
    else:
        fx = fy = float(focal)
    network._check_config()
    prop_net._check_config()
    prec = ops.current_precision()
    rays = ops.generate_rays(render_pose[:3], H, W, fx, fy, dev)                   # (H*W, 6), raster order
    sz, patch_num = get_patch_size((H, W))
    if sz is not None:                                                             # reorder rays tile by tile
        pr, pc = patch_num
        rays = rays.view(H, W, 6)[: pr * sz].reshape(pr, sz, pc, sz, 6).permute(0, 2, 1, 3, 4).reshape(-1, 6).contiguous()
    if rng == "philox":
        u_strat = u_inv = None
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())                     # one draw from the CPU generator: torch.manual_seed governs it
    else:
        if _shard is not None:
            raise ValueError("nerf_amd.render_image: a ray shard needs rng='philox' (uniforms keyed by the global ray index)")
        seed = None
        u_strat, u_inv = _draw_uniforms(H, W, sample_num, sz, patch_num, dev, rng)
    off = 0
    # integrated PE: ONE direction norm over the call's rays (mip_methods.py:31) -- taken BEFORE a shard is cut out, so that every shard
    # encodes with the norm of the whole image and the gathered image equals the single-process one
    ipe_dir_norm = ops.dirs_norm(rays) if ipe else None
    if _shard is not None:
        off = int(_shard[0])
        rays = rays[off: int(_shard[1])].contiguous()
    z_base = torch.linspace(near, far, RENDER_COARSE_PNUM, device="cpu").to(dev)   # procedures.py:52 (CPU linspace bits)
    normal_px = None
    ipe_radius = None
    if ipe:
        if is_ref_model:
            raise NotImplementedError("nerf_amd: the integrated PE is wired for the MipNeRF render path only")
        ipe_radius = (2.0 / (12.0 ** 0.5) / fx) if ipe is True else float(ipe)
    generic = network._generic() or prop_net._generic()
    if generic:
        # a network LARGER than the fused kernels' compiled shapes (hidden width > 256, > 10 octaves): the reference's tile body
        # (procedures.py:62-85) call by call on the mirrored ops -- the networks run layer by layer (nerf_amd/generic_path.py)
        rgb, depth, normal_px = _render_rays_by_calls(network, prop_net, rays, z_base, u_strat, u_inv, sample_num, near, far, white_bkg, bool(render_depth),
                                                      is_ref_model=is_ref_model,
                                                      cam_dir=render_pose[:, -2].contiguous() if (render_normal and is_ref_model) else None, seed=seed,
                                                      ray_offset=off, contract=contract, ipe_radius=ipe_radius,
                                                      ipe_dir_norm=ipe_dir_norm)
    elif not is_ref_model:
        # (a narrow fine network has no integrated-PE kernel: with ipe its 256-wide -- zero-padded -- blob is used)
        rgb, depth, _, _ = ops.render_rays(prop_net.packed(prec), network.packed(prec, wide=bool(ipe)), prec, rays, z_base, u_strat, u_inv,
                                           sample_num, near, far, white_bkg, want_depth=bool(render_depth), contract=contract,
                                           ipe_radius=ipe_radius, seed=seed, rng_ray_offset=off, ipe_dir_norm=ipe_dir_norm)
    else:
        # Ref-NeRF branch (procedures.py:71-74): coarse and fine depths are merged and sorted, the last one dropped,
        # sigma -> softplus(sigma + 0.5) before compositing (nerf_amd_render_rays_ref: six launches; the sort is a merge of two
        # ascending sets).
        rgb, depth, normal_px, _ = ops.render_rays_ref(prop_net.packed(prec), network.packed(prec), prec, rays, z_base, u_strat, u_inv,
                                                       sample_num, near, far, white_bkg, want_depth=bool(render_depth),
                                                       cam_dir=render_pose[:, -2].contiguous() if render_normal else None, flags=network.kernel_flags,
                                                       seed=seed, contract=contract, rng_ray_offset=off)

    def to_image(t, ch):
        if sz is None:
            return t.view(H, W, ch).permute(2, 0, 1).contiguous()
        pr, pc = patch_num
        img = torch.zeros((ch, H, W), dtype=torch.float32, device=dev)
        img[:, : pr * sz] = t.view(pr, pc, sz, sz, ch).permute(4, 0, 2, 1, 3).reshape(ch, pr * sz, pc * sz)
        return img

    if _shard is not None:
        return {"rgb_rays": rgb, "depth_rays": depth, "normal_rays": normal_px, "range": (off, off + rays.shape[0]), "to_image": to_image}
    result = dict()
    result["rgb"] = to_image(rgb, 3)
    if render_depth:
        result["depth_img"] = to_image(depth.unsqueeze(-1), 1).expand(3, -1, -1).contiguous()   # procedures.py:88
    if render_normal:
        result["normal_img"] = to_image(normal_px.unsqueeze(-1), 1).expand(3, -1, -1).contiguous()   # procedures.py:90
    return result


def render_only(args, model_path: str, opt_level: str, dataset_root: str = "../dataset/", output_root: str = "./output/"):
    """Render-only entry point (procedures.py:99-164): load `<model_path><name>_{mip,prop}.pth`, render either the test-set poses
    (`-e`: with loss / PSNR against the ground truth) or a 120-view orbit `pose_spherical(angle, -30, 4)`, and write one PNG
    per view to `<output_root>{given,sphere}/result_%03d.png` (rgb [, depth, normal, ground truth] side by side).
    `opt_level` (apex) is accepted and ignored: `opt_mode == "native"` or `-s` select the bf16 kernels, anything else fp32.
    `dataset_root` / `output_root` default to the reference's hard-coded locations."""
    from tqdm import tqdm

    from .addtional import LossPSNR, SoftL1Loss
    from .dataset import AdaptiveResize, CustomDataSet, save_image, to_tensor
    from .utils import fov2Focal, pose_spherical
    resize = AdaptiveResize(args.img_scale)
    testset = CustomDataSet("%s%s/" % (dataset_root, args.dataset_name), lambda im: to_tensor(resize(im)), args.scene_scale, False, use_alpha=False)
    cam_fov_test, _ = testset.getCameraParam()
    r_c = testset.r_c()
    eval_poses = args.eval_poses
    render_normal = args.render_normal and not eval_poses
    render_depth = args.render_depth and not eval_poses
    if eval_poses:
        all_poses = testset.tfs.cuda()
        loss_func, psnr_func = SoftL1Loss(), LossPSNR()
    else:
        all_poses = torch.stack([pose_spherical(float(angle), -30.0, 4.0) for angle in torch.linspace(-180, 180, 120 + 1)[:-1]], 0).cuda()
    test_focal = fov2Focal(cam_fov_test, r_c)
    if args.ref_nerf:
        from .ref_model import RefNeRF
        mip_net = RefNeRF(10, args.ide_level, hidden_unit=args.nerf_net_width, perturb_bottle_neck_w=args.bottle_neck_noise, use_srgb=args.use_srgb).cuda()
    else:
        from .mip_model import MipNeRF
        mip_net = MipNeRF(10, 4, hidden_unit=args.nerf_net_width).cuda()
    prop_net = ProposalNetwork(10, hidden_unit=args.prop_net_width).cuda()
    mip_net.loadFromFile(model_path + args.name + "_mip.pth", False)
    prop_net.loadFromFile(model_path + args.name + "_prop.pth", False)
    mip_net.eval()
    prop_net.eval()
    low_precision = bool(args.use_scaler) or args.opt_mode == "native"
    with torch.no_grad():
        for i, pose in tqdm(list(enumerate(all_poses))):
            pose = pose.clone()
            pose[:3, -1] *= args.scene_scale
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=low_precision):
                result = render_image(mip_net, prop_net, pose[:3, :], r_c, test_focal, args.near, args.far, 128, white_bkg=args.white_bkg,
                                      render_normal=render_normal, render_depth=render_depth)
            if eval_poses:
                gt_img = testset[i][0].cuda()
                loss = loss_func(result["rgb"], gt_img)
                print("Image loss:%.6f\tPSNR:%.4f" % (loss.item(), psnr_func(loss).item()))
                result["gt_img"] = gt_img
            save_image(list(result.values()), "%s%s/result_%03d.png" % (output_root, "given" if eval_poses else "sphere", i),
                       nrow=1 + render_depth + render_depth + eval_poses)                  # (the reference counts render_depth twice)


def get_parser():
    """The reference's flag set (procedures.py:166-213) so that its entry scripts parse unchanged."""
    p = argparse.ArgumentParser()
    for name, typ, default, hlp in (
        ("--epochs", int, 2400, "Training lasts for . epochs"), ("--max_save", int, 3, "Check point max save number"),
        ("--sample_ray_num", int, 1024, "<x> rays to sample per training time"),
        ("--coarse_sample_pnum", int, 64, "Points to sample in coarse net"),
        ("--fine_sample_pnum", int, 128, "Points to sample in fine net"),
        ("--eval_time", int, 5, "Tensorboard output interval (train time)"),
        ("--output_time", int, 20, "Image output interval (train time)"),
        ("--center_crop_iter", int, 0, "Produce center"), ("--prop_net_width", int, 256, "Width of proposal network"),
        ("--nerf_net_width", int, 256, "Width of nerf network"), ("--near", float, 2., "Nearest sample depth"),
        ("--far", float, 6., "Farthest sample depth"), ("--center_crop_x", float, 0.5, "Center crop x axis ratio"),
        ("--center_crop_y", float, 0.5, "Center crop y axis ratio"), ("--name", str, "model_1", "Model name for loading"),
        ("--dataset_name", str, "lego", "Input dataset name in nerf synthetic dataset"),
        ("--img_scale", float, 0.5, "Scale of the image"), ("--scene_scale", float, 1.0, "Scale of the scene"),
        ("--grad_clip", float, -0.01, "Gradient clipping parameter (Negative number means no clipping)"),
        ("--pe_period_scale", float, 0.5, "Scale of positional encoding"),
        ("--opt_mode", str, "O1", "Optimization mode: none, native (torch amp), O1, O2 (apex amp)"),
        ("--min_ratio", float, 0.01, "Minimum for now_lr / lr"), ("--decay_rate", float, 0.1, "After <decay step>, lr = lr * <decay_rate>"),
        ("--decay_step", int, 100000, "After <decay step>, lr = lr * <decay_rate>"),
        ("--warmup_step", int, 500, "Warm up step (from lowest lr to starting lr)"), ("--lr", float, 1.5e-4, "Start lr"),
        ("--ide_level", int, 4, "Max level of spherical harmonics to be used"),
        ("--bottle_neck_noise", float, 0.02, "Noise std for perturbing bottle_neck vector"),
    ):
        p.add_argument(name, type=typ, default=default, help=hlp)
    for short, long_, hlp in (
        ("-d", "--del_dir", "Delete dir ./logs and start new tensorboard records"), ("-l", "--load", "Load checkpoint or trained model."),
        ("-s", "--use_scaler", "Use AMP scaler to speed up"), ("-b", "--debug", "Code debugging (detect gradient anomaly and NaNs)"),
        ("-v", "--visualize", "Visualize proposal network"), ("-r", "--do_render", "Only render the result"),
        ("-w", "--white_bkg", "Output white background"), ("-t", "--ref_nerf", "Test Ref NeRF"),
        ("-u", "--use_srgb", "Whether to use srgb in the output or not"), ("-e", "--eval_poses", "Whether to use test set poses to render image"),
    ):
        p.add_argument(short, long_, default=False, action="store_true", help=hlp)
    for long_, hlp in (("--render_depth", "Render depth image"), ("--render_normal", "Render normal image"),
                       ("--prop_normal", "(For proposal net) Whether to learn normals")):
        p.add_argument(long_, default=False, action="store_true", help=hlp)
    return p
