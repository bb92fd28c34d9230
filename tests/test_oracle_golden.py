"""Pin the CPU oracle (oracle/nerf_oracle.py) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py, run in the build container).  CPU-only; runs in seconds.

Both sides run the same aten CPU kernels, so most checks are bit-exact (tol 0); the few that are not
(different-but-equivalent op order) use 1e-6, the bar SURVEY.md section 7 step 2 sets.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import weights as W
from conftest import max_abs
from oracle import nerf_oracle as O

NEAR, FAR = 2.0, 6.0


def test_g01_raygen(golden):
    g = golden("g01_raygen")
    pose = g["pose"]
    assert torch.equal(O.pose_spherical(37.0, -30.0, 4.0), g["pose_full"])
    assert np.allclose(O.fov2focal(0.6911112070083618, (100, 100)), g["focal_sq"].numpy(), rtol=0, atol=0)
    assert np.allclose(O.fov2focal((0.6911112070083618, 0.5), (60, 100)), g["focal_tuple"].numpy(), rtol=0, atol=0)
    assert torch.equal(O.ray_dirs_image(pose, 100, 100, tuple(g["focal_sq"].tolist())), g["ray_raw_sq"])
    assert torch.equal(O.ray_dirs_image(pose, 100, 100, 138.5), g["ray_raw_scalar"])
    assert torch.equal(O.ray_dirs_image(pose, 100, 150, tuple(g["focal_tuple_img"].tolist())), g["ray_raw_tuple"])
    pix, coords = O.pixel_table(g["img"], (1.0, 1.0))
    assert torch.equal(pix, g["pix"]) and torch.equal(coords, g["coords"])
    pix, coords = O.pixel_table(g["img"], (0.5, 0.5))
    assert torch.equal(pix, g["pix_crop"]) and torch.equal(coords, g["coords_crop"])
    idx = g["sampler_idx"]
    d = O.ray_dirs_pixels(g["coords"][idx], pose, tuple(g["focal_tuple"].tolist()))
    assert torch.equal(d, g["sampler_rays"][:, 3:])
    assert torch.equal(g["pix"][idx], g["sampler_rgb"])
    assert torch.equal(pose[:, -1].expand(16, -1), g["sampler_rays"][:, :3])


def test_g02_stratified(golden):
    g1, g = golden("g01_raygen"), golden("g02_stratified")
    z = O.stratified_train(NEAR, FAR, 32, g["u_train"])
    assert torch.equal(z, g["z_train"])
    d = O.ray_dirs_pixels(g1["coords"][g["idx"]], g1["pose"], tuple(g1["focal_tuple"].tolist()))
    assert torch.equal(d, g["rays_train"][:, 3:])
    pts = g1["pose"][:, -1] + d[:, None, :] * z[:, :, None]                   # utils.py:90
    assert torch.equal(pts, g["pts_train"])
    zr = O.stratified_render(NEAR, FAR, g["render_sample_num"], g["u_render"])
    assert torch.equal(zr, g["z_render"])
    pr = g["origin"][None, None, :] + zr[..., None] * g["dirs_render"][:, None, :]   # procedures.py:66
    assert torch.equal(pr, g["pts_render"])


def test_g03_positional_encoding(golden):
    g = golden("g03_pe")
    assert torch.equal(O.positional_encoding(g["x"], 10), g["pe10"])
    assert torch.equal(O.positional_encoding(g["x"], 4), g["pe4"])
    assert torch.equal(O.positional_encoding(g["x2d"], 4), g["pe4_2d"])


@pytest.mark.parametrize("tag", ["small", "he"])
def test_g04_g09_mlps(golden, tag):
    g = golden("g04_g09_mlp")
    with torch.no_grad():
        dens = O.proposal_forward(W.proposal_state(tag), g[tag + "_pts_c"])
        rgbo = O.mip_forward(W.mip_state(tag), g[tag + "_pts_f"])
    assert max_abs(dens, g[tag + "_density"]) <= 1e-6 * max(1.0, g[tag + "_density"].abs().max().item())
    assert max_abs(rgbo, g[tag + "_rgbo"]) <= 1e-6 * max(1.0, g[tag + "_rgbo"].abs().max().item())


def test_g05_sigma_to_weights(golden):
    g = golden("g05_weights")
    assert torch.equal(O.sigma_to_weights(g["sigma"], g["z"], g["dirs"]), g["w_prop"])
    assert torch.equal(O.sigma_to_weights(g["sigma"], g["z"], None), g["w_prop_nodir"])
    assert torch.equal(O.sigma_to_weights(g["sigma"], g["z"], None, F.relu), g["w_nerf"])
    assert torch.equal(O.sigma_to_weights(g["sigma"], g["z"], None, lambda t: t.abs()), g["w_nerf_id"])


def test_g06_max_blur(golden):
    g = golden("g06_maxblur")
    assert torch.equal(O.max_blur(g["w"], 0.01), g["out"])
    assert torch.equal(O.max_blur(g["w"], 0.25), g["out_a"])


def test_g07_inverse_sampling(golden):
    g = golden("g07_inverse")
    z, below = O.inverse_sample(g["w"], g["z"], g["u"], sort=True)
    assert torch.equal(z, g["z_sorted"]) and torch.equal(below, g["below_sorted"])
    assert torch.equal(O.inverse_sample(g["w"], g["z"], g["u"], sort=False), g["z_raw"])
    mids = 0.5 * (g["z"][..., 1:] + g["z"][..., :-1])
    s, b, a = O.sample_pdf(mids, g["w"][..., 1:-1], g["u_pdf"])
    assert torch.equal(s, g["s_pdf"]) and torch.equal(b, g["below_pdf"]) and torch.equal(a, g["above_pdf"])


def test_g08_assembly(golden):
    g = golden("g08_assembly")
    assert torch.equal(O.length2pts(g["rays"], g["zf"]), g["l2p"])
    s, z = O.coarse_fine_merge(g["rays"], g["zc"], g["zf"])
    assert torch.equal(s, g["m2_samples"]) and torch.equal(z, g["m2_z"])
    s, z, inds, order = O.coarse_fine_merge(g["rays"], g["zc"], g["zf"], g["finds"])
    assert torch.equal(s, g["m4_samples"]) and torch.equal(z, g["m4_z"])
    assert torch.equal(inds, g["m4_inds"]) and torch.equal(order, g["m4_sort"])


def test_g10_composite(golden):
    g = golden("g10_composite")
    for wb in (False, True):
        for mn in (False, True):
            rgb, w, ex = O.composite(g["rgbo"], g["z"], g["dirs"], mul_norm=mn, white_bkg=wb,
                                     render_depth=(NEAR, FAR), normal_info=(g["normal"], g["cam_z"]))
            k = "wb%d_mn%d_" % (wb, mn)
            assert torch.equal(rgb, g[k + "rgb"]) and torch.equal(w, g[k + "w"])
            assert torch.equal(ex["depth_img"], g[k + "depth"]) and torch.equal(ex["normal_img"], g[k + "normal"])
    rgb, w, _ = O.composite(g["rgbo"], g["z"], g["dirs"], density_act=F.softplus)
    assert torch.equal(rgb, g["softplus_rgb"]) and torch.equal(w, g["softplus_w"])


@pytest.mark.parametrize("tag,size,sn", [("small_50", 50, 128), ("he_50", 50, 128), ("small_100", 100, 64)])
def test_g11_render_image(golden, tag, size, sn):
    """End-to-end incl. the reference's per-tile RNG draw order (CPU default generator)."""
    g = golden("g11_render_image")
    wt = tag.split("_")[0]
    torch.manual_seed(1234)
    with torch.no_grad():
        res = O.render_image(W.proposal_state(wt), W.mip_state(wt), g["pose"], size, tuple(g[tag + "_focal"].tolist()),
                             NEAR, FAR, sn, white_bkg=True, render_depth=True)
    assert max_abs(res["rgb"], g[tag + "_rgb"]) <= 2e-6
    assert max_abs(res["depth_img"][0], g[tag + "_depth"]) <= 2e-6


def test_g12_ipe(golden):
    g = golden("g12_ipe")
    feat, mu, mu_t = O.ipe_feature(g["z"], g["rays"], 6, 0.0015)
    assert max_abs(feat, g["feat"]) <= 1e-6 and max_abs(mu, g["mu"]) <= 1e-6 and max_abs(mu_t, g["mu_t"]) <= 1e-6


def test_g18_ipe_at_network_size(golden):
    """ipe_feature with L = 10 and coneParameters against the real reference (mip_methods.py:15-58), bit for bit."""
    g = golden("g18_ipe_l10")
    feat, mu, mu_t = O.ipe_feature(g["z"], g["rays"], 10, g["radius"])
    assert torch.equal(feat, g["feat"]) and torch.equal(mu, g["mu"]) and torch.equal(mu_t, g["mu_t"])
    cp = O.cone_parameters(g["z"], g["radius"])
    assert torch.equal(cp[0], g["mu_t"]) and torch.equal(cp[1], g["var_t"]) and torch.equal(cp[2], g["var_r"])
    assert float(g["rays"][:, 3:].norm()) == g["dir_norm"]


@pytest.mark.parametrize("n", [1, 7, 8, 14, 30, 62, 63, 64, 127, 254, 255])
def test_cascade_row_sum_is_torchs_cpu_summation_order(n):
    """The HIP inverse-sampling kernels reproduce the ORDER of torch's CPU row sum for the pdf normaliser (utils.py:110-111) so that
    every searchsorted decision matches the reference; O.cascade_row_sum restates that order -- here it is checked against
    torch.sum itself, bit for bit, on the machine running the tests (the order is a property of the ATen build: 8-float vectors)."""
    x = torch.rand(4000, n, generator=torch.Generator().manual_seed(n)) ** 2 + 1e-5
    want = torch.sum(x, -1, keepdim=True)[:, 0].numpy()
    assert (O.cascade_row_sum(x.numpy()) == want).all()


def test_philox_known_answers_and_counter_layout():
    """The oracle twin of the kernels' in-kernel uniform stream: Philox4x32-10 against the published known-answer vectors of the
    Random123 distribution (counter / key all zeros and all ones), and the (ray, sample) -> counter layout of include/nerf_amd.h."""
    import numpy as np
    z, f = np.uint32([0]), np.uint32([0xFFFFFFFF])
    assert [int(x[0]) for x in O.philox4x32_10(z, z, z, z, 0, 0)] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert [int(x[0]) for x in O.philox4x32_10(f, f, f, f, 0xFFFFFFFF, 0xFFFFFFFF)] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    seed, off = 0x0123456789ABCDEF, 5_000_000_000                          # a ray counter beyond 32 bits
    us, ui = O.philox_uniforms(seed, 3, off, 64, 200)
    def block(n, j):
        return O.philox4x32_10(np.uint32([(off + n) & 0xFFFFFFFF]), np.uint32([(off + n) >> 32]), np.uint32([j]), np.uint32([0x5253]),
                               seed & 0xFFFFFFFF, seed >> 32)
    for n, s in ((0, 0), (1, 7), (2, 63)):                                 # u_strat(n, s) = word 0 of block s
        assert float(us[n, s]) == float((int(block(n, s)[0][0]) >> 8) * 2.0 ** -24)
    for n, k, j, w in ((0, 0, 0, 1), (1, 64, 0, 2), (2, 191, 63, 3), (2, 192, 64, 1), (2, 199, 71, 1)):   # u_inv: words 1..3
        assert float(ui[n, k]) == float((int(block(n, j)[w][0]) >> 8) * 2.0 ** -24)
    assert float(us.min()) >= 0.0 and float(us.max()) < 1.0


def test_g14_train_step_forward_and_losses(golden):
    """Forward half of train.py:164-199 (non-ref): softplus'd proposal density, bounds, losses."""
    g1, g = golden("g01_raygen"), golden("g14_train_step")
    prop_sd, mip_sd = W.proposal_state("small"), W.mip_state("small")
    pose = g1["pose"]
    d = O.ray_dirs_pixels(g1["coords"][g["idx"]], pose, tuple(g1["focal_tuple"].tolist()))
    rays = torch.cat((pose[:, -1].expand(32, -1), d), -1)
    assert torch.equal(rays, g["rays"])
    z_c = O.stratified_train(NEAR, FAR, 32, g["u_strat"])
    assert torch.equal(z_c, g["z_coarse"])
    with torch.no_grad():
        dens = F.softplus(O.proposal_forward(prop_sd, pose[:, -1] + d[:, None, :] * z_c[:, :, None]))
        pw = O.max_blur(O.sigma_to_weights(dens, z_c, d), 0.01)
        z_f, below = O.inverse_sample(pw, z_c, g["u_inv"], sort=True)
        z_f = z_f[..., :-1]
        rgbo = O.mip_forward(mip_sd, O.length2pts(rays, z_f))
        rend, wts, _ = O.composite(rgbo, z_f, d)
        bounds = O.get_bounds(pw, below)
        img_loss = torch.mean((rend - g["rgb_tgt"]) ** 2)
        p_loss = O.proposal_loss(bounds, wts)
    assert max_abs(z_f, g["z_fine"]) <= 2e-6
    assert (below != g["below"]).float().mean().item() <= 0.01
    assert max_abs(rend, g["rendered"]) <= 2e-6 and max_abs(wts, g["weights"]) <= 2e-6
    assert max_abs(bounds, g["bounds"]) <= 1e-5
    assert abs(img_loss.item() - g["img_loss"]) <= 1e-6 and abs(p_loss.item() - g["prop_loss"]) <= 1e-5 * max(1, g["prop_loss"])
    assert abs(O.loss_psnr(img_loss).item() - g["psnr"]) <= 1e-4


def test_g24_config0_train_step_with_adam(golden):
    """BASELINE configs[0] at its own shape (200 x 200 image, 256 rays, 32 + 64 samples): ONE iteration of train.py:151-218 run by the real
    reference (golden G24) -- the oracle restates it end to end: pixel rays, stratified depths, both networks, sampling, compositing, losses,
    torch autograd gradients and the Adam step at the scheduler's iteration-0 learning rate (the CPU plumbing leg of the config)."""
    g = golden("g24_config0_train_step")
    N = 256
    pose, focal = g["pose"], tuple(g["focal"].tolist())
    img = g["img"]
    pix, coords = O.pixel_table(img, (1.0, 1.0))                      # utils.py:47-69
    d = O.ray_dirs_pixels(coords[g["idx"]], pose, focal)
    rays = torch.cat((pose[:, -1].expand(N, -1), d), -1)
    assert torch.equal(rays, g["rays"]) and torch.equal(pix[g["idx"]], g["rgb_tgt"])
    z_c = O.stratified_train(NEAR, FAR, 32, g["u_strat"])
    assert torch.equal(z_c, g["z_coarse"])
    prop = {k: v.clone().requires_grad_(True) for k, v in W.proposal_state("small").items()}
    mip = {k: v.clone().requires_grad_(True) for k, v in W.mip_state("small").items()}
    lr = 5e-4 * N / 512
    opt = torch.optim.Adam(list(mip.values()) + list(prop.values()), lr=lr, betas=(0.9, 0.999))
    dens = F.softplus(O.proposal_forward(prop, pose[:, -1] + d[:, None, :] * z_c[:, :, None]))
    pw = O.max_blur(O.sigma_to_weights(dens, z_c, d), 0.01)
    z_f, below = O.inverse_sample(pw, z_c, g["u_inv"], sort=True)
    z_f = z_f[..., :-1]
    rend, wts, _ = O.composite(O.mip_forward(mip, O.length2pts(rays, z_f)), z_f, d)
    img_loss = torch.mean((rend - g["rgb_tgt"]) ** 2)
    p_loss = O.proposal_loss(O.get_bounds(pw, below), wts.detach())
    (p_loss + img_loss).backward()
    assert max_abs(z_f.detach(), g["z_fine"]) <= 2e-6 and (below != g["below"]).float().mean().item() <= 0.01
    assert max_abs(rend.detach(), g["rendered"]) <= 2e-6 and max_abs(wts.detach(), g["weights"]) <= 2e-6
    assert abs(img_loss.item() - g["img_loss"]) <= 1e-6 and abs(p_loss.item() - g["prop_loss"]) <= 1e-5 * max(1, g["prop_loss"])
    rel = lambda got, want: max_abs(got, want) / max(want.abs().max().item(), 1e-30)
    assert rel(mip["rgb_layer.2.weight"].grad, g["g_mip_rgb"]) <= 1e-4 and rel(prop["layers.8.weight"].grad, g["g_prop_head"]) <= 1e-4
    assert rel(mip["opacity_head.0.weight"].grad, g["g_mip_sigma"]) <= 5e-3
    assert rel(mip["lin_block2.0.weight"].grad[:8], g["g_mip_skip"]) <= 5e-2 and rel(mip["lin_block1.0.weight"].grad[:8], g["g_mip_l1"]) <= 5e-2
    lr0 = lr * (0.01 * (1.0 - 0.0) + 0.0)                            # DecayLrScheduler(0.01, 0.1, 100000, lr, 500) at train_cnt 0 (nerf_base.py:115-134)
    assert lr0 == g["lr"]
    for gr in opt.param_groups:
        gr["lr"] = lr0
    opt.step()
    # the first Adam step moves every parameter by lr0 * sign(grad) (up to eps): compare the parameters after the step
    for key, name in (("p_mip_rgb_after", "rgb_layer.2.weight"), ("p_mip_l1_after", "lin_block1.0.weight")):
        got = mip[name].detach()
        got = got[:8] if key.endswith("l1_after") else got
        assert max_abs(got, g[key]) <= 2.5 * lr0, key                # a sign flip of a cancelling gradient entry moves a parameter by 2 lr0
        assert (got - g[key]).abs().gt(1e-9).float().mean().item() <= (0.0 if "rgb" in key else 0.3), key
    assert max_abs(prop["layers.8.weight"].detach(), g["p_prop_head_after"]) <= 1e-9


def test_g15_lr_schedule(golden):
    g = golden("g15_lr")
    min_r, decay_r, step, lr, warm = 0.01, 0.1, 100000, 3e-4, 500             # nerf_base.py:115-134
    for s, want in zip(g["steps"].tolist(), g["lr"].tolist()):
        if s < warm:
            r = s / warm
            got = lr * (min_r * (1.0 - r) + r)
        else:
            got = lr * max(decay_r ** ((s - warm) / step), min_r)
        assert abs(got - want) <= 1e-12


def test_g13_ide_and_refnerf(golden):
    g = golden("g13_refnerf")
    assert torch.equal(O.ide_encode(g["ide_dirs"], g["ide_rho"], 4), g["ide"])
    for tag in ("small", "he"):
        with torch.no_grad():
            rgbo, normal = O.ref_forward(W.ref_state(tag), g[tag + "_pts"])
        assert max_abs(rgbo, g[tag + "_rgbo"]) <= 2e-6 * max(1.0, g[tag + "_rgbo"].abs().max().item())
        assert max_abs(normal, g[tag + "_normal"]) <= 2e-6


def test_g19_refnerf_use_srgb(golden):
    """RefNeRF(use_srgb=True) (ref_model.py:100-102, nerf_helper.py:50-56): forward and parameter gradients of the oracle restatement
    against the real reference's."""
    g = golden("g19_refnerf_srgb")
    for tag in ("small", "he"):
        sd = {k: v.clone().requires_grad_(True) for k, v in W.ref_state(tag).items()}
        rgbo, normal = O.ref_forward(sd, g["pts"], use_srgb=True)
        scale = max(1.0, g[tag + "_rgbo"].abs().max().item())
        assert max_abs(rgbo.detach(), g[tag + "_rgbo"]) <= 2e-6 * scale
        assert max_abs(normal.detach(), g[tag + "_normal"]) <= 2e-6
        with torch.no_grad():                                               # it is not the use_srgb=False output
            assert max_abs(O.ref_forward(W.ref_state(tag), g["pts"])[0][..., :3], g[tag + "_rgbo"][..., :3]) > 1e-2
        ((rgbo * g["g_rgbo"]).sum() + (normal * g["g_normal"]).sum()).backward()
        for key, name, rows in (("g_spec", "spec_rgb_head.0.weight", None), ("g_nct", "norm_col_tint_head.weight", None),
                                ("g_nct_bias", "norm_col_tint_head.bias", None), ("g_rho_tau", "rho_tau_head.weight", None),
                                ("g_dir2_6", "dir_block2.6.weight", 8), ("g_spa2_6", "spa_block2.6.weight", 8), ("g_spa0", "spa_block1.0.weight", 8)):
            want = g[tag + "_" + key]
            got = sd[name].grad if rows is None else sd[name].grad[:rows]
            assert max_abs(got, want) <= 2e-4 * max(1e-6, want.abs().max().item()), (tag, key)


def test_g13_render_image_refnerf(golden):
    """Reference render_image with a RefNeRF (coarse+fine merge, softplus(sigma+.5), normals), one 50x50 tile."""
    g = golden("g13_refnerf")
    pose = g["pose"]
    f = tuple(g["img_focal"].tolist())
    dirs = O.ray_dirs_image(pose, 50, 50, f).reshape(-1, 3)
    rays = torch.cat((pose[:, -1].expand(2500, -1), dirs), -1)
    torch.manual_seed(4321)
    u1 = torch.rand((50, 50, 64)).view(-1, 64)
    u2 = torch.rand((2500, 65))
    with torch.no_grad():
        rgb, _, ex = O.render_rays_ref(W.proposal_state("small"), W.ref_state("small"), rays, u1, u2, NEAR, FAR, 64, white_bkg=True,
                                       cam_z=pose[:, -2])
    assert max_abs(rgb.view(50, 50, 3).permute(2, 0, 1), g["img_rgb"]) <= 2e-6
    assert max_abs(ex["depth_img"].view(50, 50), g["img_depth"]) <= 2e-6
    assert max_abs(ex["normal_img"].view(50, 50), g["img_normal"]) <= 2e-6


def test_g17_ref_train_step(golden):
    """Ref-NeRF training step with prop_normal (train.py:164-199): the oracle restatement reproduces the reference's
    intermediates, losses and parameter gradients from the same inputs / recorded bottle-neck noise."""
    g = golden("g17_ref_train_step")
    prop = {k: v.clone().requires_grad_(True) for k, v in W.proposal_state("small").items()}
    ref = {k: v.clone().requires_grad_(True) for k, v in W.ref_state("small").items()}
    out = O.ref_train_step(prop, ref, g["rays"], g["z_coarse"], g["u_inv"], g["noise"], g["rgb_tgt"], 32)
    assert torch.equal(out["sort_ids"], g["sort_ids"]) and torch.equal(out["below_merged"], g["below_merged"])
    for k in ("z_fine", "z_merged", "rgbo_raw", "pred_normal", "weights", "rendered"):
        assert max_abs(out[k], g[k]) <= 2e-6, k
    for k in ("density_grad", "coarse_grad"):
        assert max_abs(out[k], g[k]) <= 2e-4, k                                   # unit vectors of tiny gradients (norm clamp 1e-5)
    for k in ("normal_loss", "bf_loss", "coarse_normal_loss", "img_loss", "prop_loss", "loss"):
        assert abs(float(out[k]) - float(g[k])) <= 2e-5 * max(1.0, abs(float(g[k]))), k
    out["loss"].backward()
    for key, name, rows in (("g_spa0", "spa_block1.0.weight", 8), ("g_rho_tau", "rho_tau_head.weight", None), ("g_nct", "norm_col_tint_head.weight", None),
                            ("g_bottle", "bottle_neck.weight", 8), ("g_dir0", "dir_block1.0.weight", 8), ("g_spec", "spec_rgb_head.0.weight", None)):
        got = ref[name].grad if rows is None else ref[name].grad[:rows]
        assert max_abs(got, g[key]) <= 2e-5 * max(1.0, g[key].abs().max().item()), key
    assert max_abs(prop["layers.0.weight"].grad[:8], g["g_prop_l0"]) <= 5e-4 * max(1.0, g["g_prop_l0"].abs().max().item())
    assert max_abs(prop["layers.8.weight"].grad, g["g_prop_head"]) <= 2e-5 * max(1.0, g["g_prop_head"].abs().max().item())


def test_contraction_definition():
    """contract() (Mip-NeRF 360 eq. 10; the build's own definition, parity unpinned): identity inside the unit ball, continuous at
    the boundary, image inside radius 2, direction preserved, monotone in |x|."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4000, 3, generator=g) * torch.logspace(-2, 4, 4000)[:, None]
    c = O.contract(x)
    n, cn = x.norm(dim=-1), c.norm(dim=-1)
    inside = n <= 1.0
    assert torch.equal(c[inside], x[inside])
    assert (cn[~inside] < 2.0).all() and (cn[~inside] >= 1.0 - 1e-6).all()
    assert torch.allclose(cn[~inside], 2.0 - 1.0 / n[~inside], rtol=1e-5, atol=1e-6)
    assert torch.allclose(c / cn[:, None], x / n[:, None], atol=1e-5)
    edge = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0000001, 0.0]])
    assert torch.allclose(O.contract(edge), edge, atol=1e-6)


SHALLOW = ((6, True, 256), (10, False, 256), (4, False, 96))


def shallow_states(L, cat, width):
    """the states tests/golden/make_golden.py:write_shallow_encodings loaded into the reference's modules (deterministic)"""
    seed = 100 * L + width + int(cat)
    return (O.init_linear_params(O.mip_shapes(L, 4, width, cat), seed, std=0.08, bias_std=0.05),
            O.init_linear_params(O.proposal_shapes(L, width, cat), seed + 1, std=0.08, bias_std=0.05),
            O.init_linear_params(O.ref_shapes(L, 4, width, 128, width, cat), seed + 2, std=0.08, bias_std=0.05))


@pytest.mark.parametrize("L,cat,width", SHALLOW)
def test_g21_shallow_encodings_and_cat_origin(golden, L, cat, width):
    """Fewer encoding octaves and cat_origin=False (constructor arguments of the three networks, mip_model.py:15-18, addtional.py:61,
    ref_model.py:17-24): the oracle's `Lp` / `cat_origin` parameters against the REAL modules' forward values and parameter gradients."""
    g = golden("g21_shallow_encodings")
    tag = "L%d_%s_w%d" % (L, "cat" if cat else "nocat", width)
    req = lambda sd: {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    msd, psd, rsd = (req(s) for s in shallow_states(L, cat, width))
    pts = g["pts"]
    y = O.mip_forward(msd, pts, Lp=L, cat_origin=cat)
    (y * g["G4"]).sum().backward()
    assert max_abs(y, g[tag + "_mip"]) <= 1e-6
    for key, name in (("_mip_g0", "lin_block1.0.weight"), ("_mip_gskip", "lin_block2.0.weight"), ("_mip_grgb", "rgb_layer.0.weight")):
        want = g[tag + key]
        assert tuple(msd[name].grad.shape[1:]) == tuple(want.shape[1:])
        assert max_abs(msd[name].grad[:8], want) <= 2e-5 * max(1.0, want.abs().max().item()), key
    d = O.proposal_forward(psd, pts[..., :3], L=L, cat_origin=cat)
    (d * g["G1"]).sum().backward()
    assert max_abs(d, g[tag + "_prop"]) <= 1e-6 * max(1.0, g[tag + "_prop"].abs().max().item())
    assert max_abs(psd["layers.0.weight"].grad[:8], g[tag + "_prop_g0"]) <= 2e-5 * max(1.0, g[tag + "_prop_g0"].abs().max().item())
    rgbo, nrm = O.ref_forward(rsd, pts, Lp=L, cat_origin=cat)
    ((rgbo * g["G4"]).sum() + (nrm * g["G3"]).sum()).backward()
    assert max_abs(rgbo, g[tag + "_ref_rgbo"]) <= 2e-6 * max(1.0, g[tag + "_ref_rgbo"].abs().max().item()) and max_abs(nrm, g[tag + "_ref_normal"]) <= 2e-6
    for key, name in (("_ref_g0", "spa_block1.0.weight"), ("_ref_gskip", "spa_block2.0.weight")):
        want = g[tag + key]
        assert max_abs(rsd[name].grad[:8], want) <= 1e-4 * max(1.0, want.abs().max().item()), key


GENERIC_REF_CASES = ((10, 5, 256, False), (10, 4, 320, False), (11, 3, 288, True))


def generic_ref_state(L, deg, width):
    return O.init_linear_params(O.ref_shapes(L, deg, width, 128, width, True), 3000 + 10 * L + deg + width, std=0.07, bias_std=0.05)


@pytest.mark.parametrize("L,deg,width,srgb", GENERIC_REF_CASES)
def test_g22_refnerf_outside_the_compiled_shapes(golden, L, deg, width, srgb):
    """`--ide_level 5`, hidden width 320, 11 octaves + use_srgb (the shapes the generic Ref-NeRF path of the product exists for): the oracle's
    `deg` / width / `Lp` parameters against the REAL RefNeRF's forward values, RefNeRF.get_grad and parameter gradients (golden G22)."""
    g = golden("g22_generic_refnerf")
    tag = "L%d_d%d_w%d" % (L, deg, width)
    sd = {k: v.clone().requires_grad_(True) for k, v in generic_ref_state(L, deg, width).items()}
    pos = g["pos"].clone().requires_grad_(True)
    rgbo, nrm = O.ref_forward(sd, pos, g["dirs"], Lp=L, deg=deg, use_srgb=srgb)
    grad, = torch.autograd.grad(rgbo[..., -1], pos, torch.ones_like(rgbo[..., -1]), retain_graph=True)
    gn = grad.norm(dim=-1, keepdim=True)
    grad = grad / torch.maximum(torch.full_like(gn, 1e-5), gn)
    ((rgbo * g["G4"]).sum() + (nrm * g["G3"]).sum()).backward()
    sc = lambda t: max(1.0, t.abs().max().item())
    tol = 2e-6 if L <= 10 else 2e-4          # (octave 10 multiplies the position by 1024 before sin / cos: fp32 argument rounding)
    assert max_abs(rgbo, g[tag + "_rgbo"]) <= tol * sc(g[tag + "_rgbo"]) and max_abs(nrm, g[tag + "_normal"]) <= tol
    assert max_abs(grad, g[tag + "_density_grad"]) <= 50 * tol
    for key, name in (("_g_spa0", "spa_block1.0.weight"), ("_g_dir0", "dir_block1.0.weight"), ("_g_dirskip", "dir_block2.0.weight"),
                      ("_g_heads", "norm_col_tint_head.weight"), ("_g_rho_tau", "rho_tau_head.weight"), ("_g_bottle", "bottle_neck.weight")):
        want = g[tag + key]
        got = sd[name].grad[: want.shape[0]]
        assert tuple(got.shape) == tuple(want.shape), key
        assert max_abs(got, want) <= 1e-4 * sc(want), (key, max_abs(got, want), sc(want))


def test_philox_normal_is_normal():
    """The in-kernel bottle-neck perturbation (oracle twin of device_common.h philox_normal8; the GPU test compares the kernel with it):
    zero mean, unit variance, Kolmogorov-Smirnov against N(0, 1), no correlation between features / neighbouring samples, the 4.85 sigma
    bound of 16-bit Box-Muller uniforms, and the documented feature -> (block, element) map."""
    import numpy as np
    from scipy import stats
    from oracle import nerf_oracle as O
    z = O.philox_normal(20260930, 4000, 1.0, 123).numpy().astype(np.float64)
    assert z.shape == (4000, 128) and abs(z.mean()) < 5e-3 and abs(z.std() - 1.0) < 5e-3 and np.abs(z).max() < 4.86
    assert stats.kstest(z.ravel(), "norm").pvalue > 1e-3
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.06 and abs(np.corrcoef(z[:-1].ravel(), z[1:].ravel())[0, 1]) < 5e-3
    assert abs(stats.kurtosis(z.ravel())) < 0.03
    half = O.philox_normal(20260930, 2000, 0.5, 123 + 2000).numpy()                      # std scales, the sample offset shifts the rows
    assert np.allclose(half, 0.5 * z[2000:], atol=1e-6)
