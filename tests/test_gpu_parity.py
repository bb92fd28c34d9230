"""GPU parity tests: the HIP path (through the C-ABI of libnerf_amd.so, via the host-side mirror of the
reference interface) against the CPU oracle on identical inputs, and against the golden vectors produced by
the real reference.  Run with ``-m gpu`` on an MI355X.

Tolerances (north_star: <= 1e-4 abs fp32 on RGB / depth / weights):
  * sampling / compositing kernels: 1e-6 .. 1e-5 (same arithmetic, different summation order);
  * fp32-MFMA MLPs: 1e-5 x output scale;   end-to-end fp32 render: 1e-4 abs;
  * bf16-MFMA MLPs: compared with the oracle run in bf16-operand emulation (operands rounded to bf16, fp32
    accumulate), 5e-3 x output scale; bf16 end-to-end is judged on image error / PSNR, not on 1e-4.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import weights as W
from conftest import max_abs, gate
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
NEAR, FAR = 2.0, 6.0


@pytest.fixture(scope="module")
def A():
    """The product package, imported lazily so that collection works without a GPU."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import nerf_amd
    from nerf_amd import ops, procedures, utils, nerf_base, mip_model, addtional, mip_methods, nerf_helper

    class NS:
        pass
    ns = NS()
    ns.pkg, ns.ops, ns.procedures, ns.utils, ns.nerf_base = nerf_amd, ops, procedures, utils, nerf_base
    ns.mip_model, ns.addtional, ns.mip_methods, ns.nerf_helper = mip_model, addtional, mip_methods, nerf_helper
    return ns


def dev(t):
    return t.cuda()


def build_nets(A, tag):
    prop = A.addtional.ProposalNetwork(10, 256)
    mip = A.mip_model.MipNeRF(10, 4, 256)
    prop.load_state_dict(W.proposal_state(tag))
    mip.load_state_dict(W.mip_state(tag))
    return prop.cuda().eval(), mip.cuda().eval()


def test_loaded_library_is_native(A):
    n_cu, is950 = A.ops.C.c_int(0), A.ops.C.c_int(0)
    assert A.ops.lib.nerf_amd_device_info(A.ops.C.byref(n_cu), A.ops.C.byref(is950)) == 0
    assert n_cu.value >= 64 and is950.value == 1


# ------------------------------------------------------------------------------------------------ rows 1-3
def test_raygen(A, golden):
    g = golden("g01_raygen")
    pose = g["pose"]
    for size, focal, key in (((100, 100), tuple(g["focal_sq"].tolist()), "ray_raw_sq"), ((100, 100), 138.5, "ray_raw_scalar"),
                             ((100, 150), tuple(g["focal_tuple_img"].tolist()), "ray_raw_tuple")):
        fx, fy = (focal[1], focal[0]) if isinstance(focal, tuple) else (focal, focal)
        rays = A.ops.generate_rays(pose, size[0], size[1], fx, fy, "cuda").cpu()
        assert max_abs(rays[:, 3:].view(size[0], size[1], 3), g[key]) <= 1e-6
        assert torch.equal(rays[:, :3], pose[:, -1].expand(size[0] * size[1], -1))
    # sub-range == slice of the full table
    full = A.ops.generate_rays(pose, 100, 100, 138.5, 138.5, "cuda")
    part = A.ops.generate_rays(pose, 100, 100, 138.5, 138.5, "cuda", ray_offset=1234, n=777)
    assert torch.equal(full[1234:1234 + 777], part)


def test_valid_sampler_matches_reference_rng(A, golden):
    g1, g = golden("g01_raygen"), golden("g02_stratified")
    torch.manual_seed(5)
    pts, z, rgb, rays = A.utils.validSampler(dev(g1["pix"]), dev(g1["coords"]), dev(g1["pose"]), 8, 32,
                                             tuple(g1["focal_tuple"].tolist()), NEAR, FAR, True, rng="reference")
    assert torch.equal(z.cpu(), g["z_train"]) and torch.equal(rgb.cpu(), g["rgb_train"])
    assert max_abs(rays.cpu(), g["rays_train"]) <= 1e-6
    assert max_abs(pts.cpu(), g["pts_train"]) <= 1e-5
    torch.manual_seed(11)
    rgb2, rays2 = A.utils.validSampler(dev(g1["pix"]), dev(g1["coords"]), dev(g1["pose"]), 16, 32,
                                       tuple(g1["focal_tuple"].tolist()), NEAR, FAR, False, rng="reference")
    assert torch.equal(rgb2.cpu(), g1["sampler_rgb"]) and max_abs(rays2.cpu(), g1["sampler_rays"]) <= 1e-6


def test_valid_sampler_in_kernel_rng(A, golden):
    """The default validSampler draws pixel indices and depth jitter inside ONE kernel (nerf_amd_sample_training_rays, Philox):
    every output is consistent with the reference's definitions for the pixels it drew (utils.py:78-90) -- colour = that pixel's,
    ray = the ray through it (same kernel arithmetic as the reference-stream path), depths stratified in their bins, pts = o + d z --
    it replays under torch.manual_seed, and the pixel draw is uniform over the table."""
    g1 = golden("g01_raygen")
    pix, coords, pose = dev(g1["pix"]), dev(g1["coords"]), dev(g1["pose"])
    focal = tuple(g1["focal_tuple"].tolist())
    N, C = 3001, 48
    torch.manual_seed(9)
    pts, z, rgb, rays = A.utils.validSampler(pix, coords, pose, N, C, focal, NEAR, FAR, True)
    torch.manual_seed(9)
    pts2, z2, rgb2, rays2 = A.utils.validSampler(pix, coords, pose, N, C, focal, NEAR, FAR, True)
    assert all(torch.equal(a, b) for a, b in ((pts, pts2), (z, z2), (rgb, rgb2), (rays, rays2)))
    rgb3, rays3 = A.utils.validSampler(pix, coords, pose, N, C, focal, NEAR, FAR, False)
    assert rgb3.shape == (N, 3) and rays3.shape == (N, 6) and not torch.equal(rays3, rays)          # (the next seed of the CPU stream)
    # which pixel did ray n take?  invert the ray: the table's rays are distinct
    table = A.ops.pixel_rays(coords, pose, float(focal[1]), float(focal[0]))
    d2 = torch.cdist(rays[:, 3:].double(), table[:, 3:].double())
    idx = d2.argmin(dim=1)
    assert float(d2.gather(1, idx[:, None]).max()) == 0.0 and torch.equal(rays, table[idx]) and torch.equal(rgb, pix[idx])
    res = (FAR - NEAR) / C
    base = NEAR + res * torch.arange(C, device="cuda", dtype=torch.float32)
    assert bool((z >= base - 1e-6).all()) and bool((z < base + res + 1e-6).all())
    assert max_abs(pts, rays[:, None, :3] + rays[:, None, 3:] * z[:, :, None]) <= 1e-6
    # uniformity of the pixel draw and of the jitter (chi-square over 16 index bins, KS on the jitter)
    from scipy import stats
    counts = torch.bincount((idx * 16 // coords.shape[0]).cpu(), minlength=16).double().numpy()
    assert stats.chisquare(counts).pvalue > 1e-3
    assert stats.kstest(((z - base) / res).clamp(0, 1).cpu().double().numpy().ravel(), "uniform").pvalue > 1e-3


def test_positional_encoding(A, golden):
    g = golden("g03_pe")
    assert max_abs(A.nerf_helper.positional_encoding(dev(g["x"]), 10).cpu(), g["pe10"]) <= 3e-7
    assert max_abs(A.nerf_helper.positional_encoding(dev(g["x"]), 4).cpu(), g["pe4"]) <= 3e-7
    assert max_abs(A.nerf_helper.positional_encoding(dev(g["x2d"]), 4).cpu(), g["pe4_2d"]) <= 3e-7
    x = (torch.rand(4096, 3, generator=torch.Generator().manual_seed(1)) - 0.5) * 20
    assert max_abs(A.nerf_helper.positional_encoding(dev(x), 10).cpu(), O.positional_encoding(x, 10)) <= 3e-7
    assert A.nerf_helper.positional_encoding(torch.empty(0, 3, device="cuda"), 10).shape == (0, 60)


# ------------------------------------------------------------------------------------------------ rows 5-8, 10, 11
def test_sigma_to_weights(A, golden):
    g = golden("g05_weights")
    s, z, d = dev(g["sigma"]), dev(g["z"]), dev(g["dirs"])
    assert max_abs(A.addtional.ProposalNetwork.get_weights(s, z, d).cpu(), g["w_prop"]) <= 1e-6
    assert max_abs(A.addtional.ProposalNetwork.get_weights(s, z, None).cpu(), g["w_prop_nodir"]) <= 1e-6
    assert max_abs(A.nerf_base.NeRF.getNormedWeight(s, z).cpu(), g["w_nerf"]) <= 1e-6
    assert max_abs(A.nerf_base.NeRF.getNormedWeight(s, z, lambda t: t.abs()).cpu(), g["w_nerf_id"]) <= 1e-6
    # ragged sample counts, incl. > 64 (multi-chunk scan carry) and 1
    gen = torch.Generator().manual_seed(3)
    for S in (1, 2, 63, 64, 65, 128, 129, 192, 257):
        sig = torch.randn(7, S, generator=gen) * 2
        zz, _ = torch.sort(NEAR + (FAR - NEAR) * torch.rand(7, S, generator=gen), dim=-1)
        assert max_abs(A.nerf_base.NeRF.getNormedWeight(dev(sig), dev(zz)).cpu(), O.sigma_to_weights(sig, zz)) <= 1e-6, S


def test_max_blur(A, golden):
    g = golden("g06_maxblur")
    assert torch.equal(A.mip_methods.maxBlurFilter(dev(g["w"]), 0.01).cpu(), g["out"])
    assert torch.equal(A.mip_methods.maxBlurFilter(dev(g["w"]), 0.25).cpu(), g["out_a"])


def _below_mismatch(a, b):
    return (a != b).float().mean().item()


def _assert_below_explained(got_below, want_below, u, cdf_want, cdf_got, what=""):
    """`below` is index work: it must EQUAL the reference's except where the two sides searched CDFs that differ (upstream float
    differences: the GPU's expf / MLP bits) and the uniform fell between the two versions of one edge.  Every mismatch must be off
    by exactly one bin, and u must lie inside [min, max] of that edge's two CDF values, widened by one fp32 ulp.
    got/want_below (N,K) in the order of `u` (N,K) -- pass u sorted when the samples were sorted; cdf_* (N,B)."""
    got_below, want_below = got_below.cpu().long(), want_below.cpu().long()
    bad = got_below != want_below
    if not bool(bad.any()):
        return 0
    n_idx, k_idx = bad.nonzero(as_tuple=True)
    g, w = got_below[n_idx, k_idx], want_below[n_idx, k_idx]
    assert bool(((g - w).abs() == 1).all()), what + ": a `below` index is off by more than one bin"
    edge = torch.maximum(g, w)                                   # the CDF entry the two searches disagree about
    c1, c2 = cdf_want[n_idx, edge].double(), cdf_got[n_idx, edge].double()
    uu = u.cpu()[n_idx, k_idx].double()
    ulp = 2.0 ** -23
    lo, hi = torch.minimum(c1, c2) - ulp, torch.maximum(c1, c2) + ulp
    assert bool(((uu >= lo) & (uu <= hi)).all()), what + ": a `below` mismatch is not explained by the two CDFs' difference at that edge"
    return int(bad.sum())


def test_inverse_sampling(A, golden):
    """Row 7 on identical inputs: the kernels sum the pdf normaliser in torch's CPU order (O.cascade_row_sum), accumulate the CDF in
    fp64 like torch's cumsum and search it like searchsorted(right=True) -- so the INDICES equal the reference's exactly (index
    work: bit-exact) and the depths agree to the last bits of the same fp32 expression."""
    g = golden("g07_inverse")
    z, below = A.utils.inverseSample(dev(g["w"]), dev(g["z"]), 129, sort=True, u=g["u"])
    assert torch.equal(below.cpu(), g["below_sorted"])
    assert max_abs(z.cpu(), g["z_sorted"]) <= 1e-6
    assert bool((z[:, 1:] >= z[:, :-1]).all())
    zr = A.utils.inverseSample(dev(g["w"]), dev(g["z"]), 129, sort=False, u=g["u"])
    assert max_abs(zr.cpu(), g["z_raw"]) <= 1e-6
    mids = 0.5 * (g["z"][..., 1:] + g["z"][..., :-1])
    s, b, a = A.utils.sample_pdf(dev(mids), dev(g["w"][..., 1:-1].contiguous()), 33, u=g["u_pdf"])
    assert max_abs(s.cpu(), g["s_pdf"]) <= 1e-6
    assert torch.equal(b.cpu(), g["below_pdf"]) and torch.equal(a.cpu(), g["above_pdf"])
    # the reference's own RNG protocol: a seeded CPU draw inside inverseSample
    torch.manual_seed(21)
    z2, _ = A.utils.inverseSample(dev(g["w"]), dev(g["z"]), 129, sort=True)
    assert torch.equal(z2, z)


@pytest.mark.parametrize("C", [3, 5, 9, 10, 17, 34, 64, 130, 256])
def test_inverse_sampling_indices_are_exact_for_every_row_length(A, C):
    """The pdf-normaliser order has three regimes (rows < 8: scalar accumulators; whole vectors + tail; >= 4 vectors: 4-way
    interleave); `below` must equal torch's searchsorted on the oracle's CDF in all of them."""
    gen = torch.Generator().manual_seed(100 + C)
    N, K = 53, 97
    w = torch.rand(N, C, generator=gen) ** 3 + 0.01
    z = torch.sort(NEAR + (FAR - NEAR) * torch.rand(N, C, generator=gen), dim=-1)[0]
    u = torch.rand(N, K, generator=gen)
    want_z, want_b = O.inverse_sample(w, z, u, sort=True)
    got_z, got_b = A.utils.inverseSample(dev(w), dev(z), K, sort=True, u=u)
    assert torch.equal(got_b.cpu(), want_b)
    assert max_abs(got_z.cpu(), want_z) <= 2e-6


def test_assembly(A, golden):
    g = golden("g08_assembly")
    assert max_abs(A.nerf_base.NeRF.length2pts(dev(g["rays"]), dev(g["zf"])).cpu(), g["l2p"]) == 0.0
    s, z = A.nerf_base.NeRF.coarseFineMerge(dev(g["rays"]), dev(g["zc"]), dev(g["zf"]))
    assert torch.equal(s.cpu(), g["m2_samples"]) and torch.equal(z.cpu(), g["m2_z"])
    s, z, inds, order = A.nerf_base.NeRF.coarseFineMerge(dev(g["rays"]), dev(g["zc"]), dev(g["zf"]), dev(g["finds"]))
    assert torch.equal(s.cpu(), g["m4_samples"]) and torch.equal(z.cpu(), g["m4_z"])
    assert torch.equal(inds.cpu(), g["m4_inds"])


def test_composite(A, golden):
    g = golden("g10_composite")
    rgbo, z, d, nrm, cz = dev(g["rgbo"]), dev(g["z"]), dev(g["dirs"]), dev(g["normal"]), dev(g["cam_z"])
    for wb in (False, True):
        for mn in (False, True):
            rgb, w, ex = A.nerf_base.NeRF.render(rgbo, z, d, mul_norm=mn, white_bkg=wb, render_depth=(NEAR, FAR),
                                                 normal_info=(nrm, cz))
            k = "wb%d_mn%d_" % (wb, mn)
            assert max_abs(rgb.cpu(), g[k + "rgb"]) <= 2e-6 and max_abs(w.cpu(), g[k + "w"]) <= 1e-6
            assert max_abs(ex["depth_img"].cpu(), g[k + "depth"]) <= 2e-6
            assert max_abs(ex["normal_img"].cpu(), g[k + "normal"]) <= 2e-6
    rgb, w, _ = A.nerf_base.NeRF.render(rgbo, z, d, density_act=F.softplus)
    assert max_abs(rgb.cpu(), g["softplus_rgb"]) <= 2e-6 and max_abs(w.cpu(), g["softplus_w"]) <= 1e-6


def test_get_bounds(A, golden):
    g = golden("g14_train_step")
    gen = torch.Generator().manual_seed(4)
    w = torch.rand(32, 32, generator=gen)
    got = A.addtional.getBounds(dev(w), dev(g["below"])).cpu()
    assert max_abs(got, O.get_bounds(w, g["below"])) <= 1e-5


# ------------------------------------------------------------------------------------------------ rows 4 / 9: MLPs
@pytest.mark.parametrize("tag", ["small", "he"])
def test_mlps_fp32_vs_reference_golden(A, golden, tag):
    g = golden("g04_g09_mlp")
    prop, mip = build_nets(A, tag)
    A.pkg.set_precision("fp32")
    with torch.no_grad():
        dens = prop.forward(dev(g[tag + "_pts_c"])).cpu()
        rgbo = mip.forward(dev(g[tag + "_pts_f"])).cpu()
    sd = max(1.0, g[tag + "_density"].abs().max().item())
    so = max(1.0, g[tag + "_rgbo"].abs().max().item())
    assert max_abs(dens, g[tag + "_density"]) <= 1e-5 * sd
    assert max_abs(rgbo, g[tag + "_rgbo"]) <= 1e-5 * so


@pytest.mark.parametrize("tag", ["small", "he"])
@pytest.mark.parametrize("M", [1, 31, 32, 33, 127, 128, 129, 255, 256, 257, 1000, 70001])
def test_mlps_ragged_sizes(A, tag, M):
    """Tail handling: sample counts around the 128 (fp32) / 256 (bf16) tile sizes, persistent multi-tile grids."""
    if tag == "he" and M not in (33, 257, 70001):
        pytest.skip("one weight flavour is enough for the small sizes")
    prop, mip = build_nets(A, tag)
    gen = torch.Generator().manual_seed(M)
    pts = torch.cat(((torch.rand(M, 1, 3, generator=gen) - 0.5) * 8, torch.randn(M, 1, 3, generator=gen)), -1)
    with torch.no_grad():
        want_d = O.proposal_forward(W.proposal_state(tag), pts[..., :3])
        want_o = O.mip_forward(W.mip_state(tag), pts)
        want_d16 = O.proposal_forward(W.proposal_state(tag), pts[..., :3], emulate_bf16=True)
        want_o16 = O.mip_forward(W.mip_state(tag), pts, emulate_bf16=True)
        A.pkg.set_precision("fp32")
        got_d, got_o = prop.forward(dev(pts[..., :3].contiguous())).cpu(), mip.forward(dev(pts)).cpu()
        A.pkg.set_precision("bf16")
        got_d16, got_o16 = prop.forward(dev(pts[..., :3].contiguous())).cpu(), mip.forward(dev(pts)).cpu()
        A.pkg.set_precision("fp32")
    sd, so = max(1.0, want_d.abs().max().item()), max(1.0, want_o.abs().max().item())
    assert max_abs(got_d, want_d) <= 1e-5 * sd and max_abs(got_o, want_o) <= 1e-5 * so
    # bf16 mode vs the bf16-operand emulation: differences are single bf16-ulp (2^-8) rounding flips of activations
    # that sit on a rounding boundary (fp32 sum order differs), amplified by the O(1) 'he' weights
    assert max_abs(got_d16, want_d16) <= 1e-2 * sd and max_abs(got_o16, want_o16) <= 1e-2 * so


def test_mlp_empty_and_autocast(A):
    prop, mip = build_nets(A, "small")
    A.pkg.set_precision(None)
    with torch.no_grad():
        assert prop.forward(torch.empty(0, 64, 3, device="cuda")).shape == (0, 64)
        assert mip.forward(torch.empty(0, 128, 6, device="cuda")).shape == (0, 128, 4)
        pts = torch.randn(5, 7, 6, device="cuda")
        a = mip.forward(pts)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            b = mip.forward(pts)
        A.pkg.set_precision("bf16")
        c = mip.forward(pts)
    A.pkg.set_precision("fp32")
    assert torch.equal(b, c) and not torch.equal(a, b)


def test_repack_after_inplace_update(A):
    prop, _ = build_nets(A, "small")
    pts = torch.randn(3, 64, 3, device="cuda")
    with torch.no_grad():
        a = prop.forward(pts)
        prop.layers[8].bias.add_(1.0)
        b = prop.forward(pts)
    assert max_abs(b.cpu(), a.cpu() + 1.0) <= 5e-6          # the bias seeds the fp32 accumulator, so the rounding path differs


def test_product_rejects_cpu_tensors(A):
    prop, mip = build_nets(A, "small")
    with pytest.raises(RuntimeError):
        A.nerf_helper.positional_encoding(torch.zeros(4, 3), 10)
    with pytest.raises(RuntimeError):
        A.procedures.render_image(mip, prop, torch.eye(4)[:3], 50, 100.0, NEAR, FAR)


# ------------------------------------------------------------------------------------------------ end to end
def _rays_and_u(n, n_fine, seed):
    gen = torch.Generator().manual_seed(seed)
    pose = O.pose_spherical(-63.0, -30.0, 4.0)[:3]
    dirs = O.ray_dirs_image(pose, 80, 80, O.fov2focal(0.6911112070083618, (80, 80))).reshape(-1, 3)
    pick = torch.randperm(dirs.shape[0], generator=gen)[:n]
    rays = torch.cat((pose[:, -1].expand(n, -1), dirs[pick]), -1).contiguous()
    return rays, torch.rand(n, 64, generator=gen), torch.rand(n, n_fine + 1, generator=gen)


@pytest.mark.parametrize("tag,n,n_fine", [("small", 256, 64), ("small", 300, 128), ("he", 300, 128),
                                          ("small", 200, 40), ("small", 150, 200), ("small", 100, 300)])
def test_render_rays_fp32_parity(A, tag, n, n_fine):
    """The north-star gate: RGB / depth / weights vs the CPU path on identical rays and uniforms, <= 1e-4 abs.  Sample counts beside the
    headline ones: 40 (the stratified jitter exceeds the bin spacing, so the coarse depths are out of order -- the reference does not
    sort them either), 200 (beyond the resampling kernel's register-prefetch shape, four-chunk compositing) and 300 (generic paths)."""
    prop, mip = build_nets(A, tag)
    rays, u1, u2 = _rays_and_u(n, n_fine, 17)
    stages = {}
    with torch.no_grad():
        want_rgb, want_w, want_depth = O.render_rays(W.proposal_state(tag), W.mip_state(tag), rays, u1, u2, NEAR, FAR, n_fine,
                                                     white_bkg=True, stages=stages)
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    rgb, depth, w, _ = A.ops.render_rays(prop.packed(A.ops.F32), mip.packed(A.ops.F32), A.ops.F32, dev(rays), z_base, dev(u1), dev(u2),
                                         n_fine, NEAR, FAR, True, want_depth=True, want_weights=True)
    tol = 1e-4 if tag == "small" else 2e-4           # 'he': the fp32 reference's own noise floor is ~1e-4 (test_he_weights_conditioning)
    assert max_abs(rgb.cpu(), want_rgb) <= tol
    assert max_abs(depth.cpu(), want_depth) <= tol
    assert max_abs(w.cpu(), want_w) <= tol
    # stage by stage through the individual entry points
    jit = (FAR - NEAR) / n_fine
    dens = A.ops.proposal_forward_samples(prop.packed(A.ops.F32), A.ops.F32,
                                          A.ops.samples_rays(dev(rays), 64, z_base=z_base, u=dev(u1), z_jitter=jit), (n, 64), "cuda")
    assert max_abs(dens.cpu(), stages["density"]) <= 1e-5 * max(1.0, stages["density"].abs().max().item())
    z_f, below, w_prop, z_c = A.ops.resample(dens, None, z_base, dev(u1), jit, dev(rays), dev(u2), n_fine + 1, want_below=True,
                                             want_w=True, want_zc=True)
    assert torch.equal(z_c.cpu(), stages["z_coarse"])
    assert max_abs(w_prop.cpu(), stages["w_prop"]) <= 1e-5
    assert max_abs(z_f[:, :-1].cpu(), stages["z_fine"]) <= 2e-5
    # `below` is index work.  (i) On IDENTICAL inputs -- the oracle's sampler fed the GPU's own blurred weights and coarse depths -- the
    # sorted indices and the sort are equal, bit for bit, for every sample count.
    _, below_same = O.inverse_sample(w_prop.cpu(), z_c.cpu(), u2, sort=True)
    assert torch.equal(below.cpu(), below_same)
    # (ii) Against the oracle's own chain the GPU's w_prop differs in its last bits (expf / MLP), which can move a CDF edge across a
    # uniform: every mismatch must be EXPLAINED by that (off by one bin, u between the two versions of the edge), and they are rare.
    cdf_want, cdf_got = O.pdf_cdf(stages["w_prop"][:, 1:-1]), O.pdf_cdf(w_prop.cpu()[:, 1:-1])
    if n_fine >= 63:                                          # (ascending bins: the sorted samples are in the order of their uniforms)
        n_bad = _assert_below_explained(below, stages["below"], torch.sort(u2, dim=-1)[0], cdf_want, cdf_got, "resample")
    else:                                                     # (out-of-order coarse depths: compare in the order of the uniforms, before the sort)
        mids = lambda z: 0.5 * (z[..., 1:] + z[..., :-1])
        _, b_got, _ = A.ops.sample_pdf(mids(z_c), w_prop[:, 1:-1].contiguous(), dev(u2))
        _, b_want, _ = O.sample_pdf(mids(stages["z_coarse"]), stages["w_prop"][:, 1:-1], u2)
        n_bad = _assert_below_explained(b_got, b_want, u2, cdf_want, cdf_got, "sample_pdf")
    assert n_bad <= 0.002 * below.numel()


BF16_RENDER_DB, BF16_DEPTH_TOL = 90.0, 1e-4             # measured on MI355X (round 3): 102.4 dB, max |rgb err| 2.5e-5, max |depth err| 2.1e-5


def test_render_rays_bf16_close(A):
    prop, mip = build_nets(A, "small")
    rays, u1, u2 = _rays_and_u(512, 128, 5)
    with torch.no_grad():
        want_rgb, _, want_depth = O.render_rays(W.proposal_state("small"), W.mip_state("small"), rays, u1, u2, NEAR, FAR, 128,
                                                white_bkg=True)
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    rgb, depth, _, _ = A.ops.render_rays(prop.packed(A.ops.BF16), mip.packed(A.ops.BF16), A.ops.BF16, dev(rays), z_base, dev(u1), dev(u2),
                                         128, NEAR, FAR, True)
    mse = torch.mean((rgb.cpu() - want_rgb) ** 2).item()
    print("\nbf16 render vs fp32 oracle: %.1f dB, max |rgb err| %.2e, max |depth err| %.2e" % (-10 * math.log10(max(mse, 1e-12)), max_abs(rgb.cpu(), want_rgb),
                                                                                                 max_abs(depth.cpu(), want_depth)))
    # gates a few times above the measured error of this configuration (reference-style weights), not at "roughly an image"
    assert -10 * math.log10(max(mse, 1e-12)) >= BF16_RENDER_DB
    assert max_abs(depth.cpu(), want_depth) <= BF16_DEPTH_TOL


@pytest.mark.parametrize("tag,size,sn", [("small_50", 50, 128), ("he_50", 50, 128), ("small_100", 100, 64), ("small_200", 200, 64)])
def test_render_image_vs_reference(A, golden, tag, size, sn):
    """Drop-in surface: same seed -> same image as the REAL reference's render_image (its tile order and
    CPU RNG draw order reproduced), 1e-4 abs.  small_200 is BASELINE config 1's image size."""
    g = golden("g11_render_image")
    prop, mip = build_nets(A, tag.split("_")[0])
    A.pkg.set_precision("fp32")
    torch.manual_seed(1234)
    with torch.no_grad():
        res = A.procedures.render_image(mip, prop, dev(g["pose"]), size, tuple(g[tag + "_focal"].tolist()), NEAR, FAR, sn,
                                        white_bkg=True, render_depth=True, rng="reference")
    assert list(res.keys()) == ["rgb", "depth_img"]
    assert res["rgb"].shape == (3, size, size) and res["depth_img"].shape == (3, size, size)
    # 'he' weights (O(1) activations through 10 PE octaves) are an error-amplification stress on which the fp32 reference itself is
    # ~1e-4 from the exact value of its own expressions (test_he_weights_conditioning): 2e-4.  Reference-style weights ('small') are
    # two orders of magnitude below the 1e-4 north-star gate.
    tol = 1e-4 if tag.startswith("small") else 2e-4
    assert max_abs(res["rgb"].cpu(), g[tag + "_rgb"]) <= tol
    assert max_abs(res["depth_img"][0].cpu(), g[tag + "_depth"]) <= tol
    assert torch.equal(res["depth_img"][0], res["depth_img"][2])


def test_render_through_the_nerf_shim_package(A, golden):
    """the drop-in is importable as `nerf.*`.  With compat/ on sys.path every name the reference's entry scripts import
    (golden G23, train.py:12-20 / ddp_train.py:17-25 / model_average.py:16-27) comes from the shim, and a render through
    `nerf.procedures.render_image` with modules built from `nerf.mip_model` / `nerf.addtional` reproduces the REAL reference's image
    (golden G11, BASELINE config 1's 200 x 200) to 1e-4 -- what `train.py -r` would run."""
    import importlib
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compat")
    sys.path.insert(0, compat)
    try:
        for k in [k for k in sys.modules if k == "nerf" or k.startswith("nerf.")]:
            del sys.modules[k]
        want = golden("g23_entry_imports")
        for script, imp in want.items():
            for m, names in imp.items():
                mod = importlib.import_module("nerf." + m)
                assert mod.__file__.startswith(compat), mod.__file__
                for name in names:
                    assert hasattr(mod, name), (script, m, name)
        from nerf.addtional import ProposalNetwork
        from nerf.mip_model import MipNeRF
        from nerf.procedures import render_image
        from nerf.timer import Timer
        g = golden("g11_render_image")
        prop, mip = ProposalNetwork(10, 256), MipNeRF(10, 4, 256)
        prop.load_state_dict(W.proposal_state("small"))
        mip.load_state_dict(W.mip_state("small"))
        prop, mip = prop.cuda().eval(), mip.cuda().eval()
        A.pkg.set_precision("fp32")
        torch.manual_seed(1234)
        t = Timer(3)
        t.tic()
        with torch.no_grad():
            res = render_image(mip, prop, dev(g["pose"]), 200, tuple(g["small_200_focal"].tolist()), NEAR, FAR, 64, white_bkg=True, render_depth=True,
                               rng="reference")
        assert t.toc() > 0 and isinstance(t.remaining_time(3), str)
        assert list(res.keys()) == ["rgb", "depth_img"]
        assert max_abs(res["rgb"].cpu(), g["small_200_rgb"]) <= 1e-4 and max_abs(res["depth_img"][0].cpu(), g["small_200_depth"]) <= 1e-4
    finally:
        sys.path.remove(compat)
        for k in [k for k in sys.modules if k == "nerf" or k.startswith("nerf.")]:
            del sys.modules[k]


@pytest.mark.parametrize("prec,tag", [("fp32", "small"), ("fp32", "he"), ("bf16", "he")])
def test_full_size_properties(A, prec, tag):
    """BASELINE config 2 size (800x800, 64+128, all 640 000 rays in both precisions): properties that need no oracle at this scale + an
    oracle spot check on a random subset (fp32: 2 048 rays; reference-style `small` weights at the north star's 1e-4, the O(1)-activation
    `he` stress weights at 2.5e-4 -- on those the fp32 REFERENCE itself sits ~1e-4 from the exact value of its own expressions,
    test_he_weights_conditioning)."""
    prop, mip = build_nets(A, tag)
    P = A.ops.F32 if prec == "fp32" else A.ops.BF16
    H = Wd = 800
    pose = O.pose_spherical(20.0, -30.0, 4.0)[:3]
    f = O.fov2focal(0.6911112070083618, (H, Wd))
    n = H * Wd                                               # the whole image in both precisions (round 6: fp32 used to stop at a 200 000-ray slab)
    rays = A.ops.generate_rays(pose, H, Wd, f[1], f[0], "cuda", 0, n)
    gen = torch.Generator(device="cuda").manual_seed(9)
    u1 = torch.rand(n, 64, device="cuda", generator=gen)
    u2 = torch.rand(n, 129, device="cuda", generator=gen)
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    pk_p, pk_m = prop.packed(P), mip.packed(P)
    rgb_w, depth, w, ws = A.ops.render_rays(pk_p, pk_m, P, rays, z_base, u1, u2, 128, NEAR, FAR, True, want_depth=True, want_weights=True)
    rgb_b, _, _, ws = A.ops.render_rays(pk_p, pk_m, P, rays, z_base, u1, u2, 128, NEAR, FAR, False, workspace=ws)
    acc = w.sum(-1)
    assert bool(torch.isfinite(rgb_w).all()) and bool(torch.isfinite(depth).all())
    assert float(acc.max()) <= 1.0 + 1e-4 and float(w.min()) >= 0.0
    assert max_abs(rgb_w - rgb_b, (1.0 - acc)[:, None].expand(-1, 3)) <= 2e-6          # white background is affine in (1 - sum w)
    assert float(rgb_b.min()) >= 0.0 and float(rgb_b.max()) <= 1.0 + 1e-5
    # batch-composition independence: any sub-batch reproduces the same bits
    lo, cnt = 123_457, 4_099
    r2, d2, w2, _ = A.ops.render_rays(pk_p, pk_m, P, rays[lo:lo + cnt].contiguous(), z_base, u1[lo:lo + cnt].contiguous(),
                                      u2[lo:lo + cnt].contiguous(), 128, NEAR, FAR, True, want_depth=True, want_weights=True)
    assert torch.equal(r2, rgb_w[lo:lo + cnt]) and torch.equal(d2, depth[lo:lo + cnt]) and torch.equal(w2, w[lo:lo + cnt])
    # oracle spot check on a random subset
    pick = torch.randperm(n, generator=torch.Generator().manual_seed(2))[:2048 if prec == "fp32" else 192]
    with torch.no_grad():
        want_rgb, want_w, want_depth = O.render_rays(W.proposal_state(tag), W.mip_state(tag), rays[pick].cpu(), u1[pick].cpu(),
                                                     u2[pick].cpu(), NEAR, FAR, 128, white_bkg=True, emulate_bf16=False)
    if prec == "fp32":
        tol = 1e-4 if tag == "small" else 2.5e-4
        gate("full-size fp32 [%s] rgb vs oracle, 2048 rays of 640000" % tag, max_abs(rgb_w[pick].cpu(), want_rgb), tol)
        gate("full-size fp32 [%s] depth vs oracle, 2048 rays of 640000" % tag, max_abs(depth[pick].cpu(), want_depth), tol)
        gate("full-size fp32 [%s] weights vs oracle, 2048 rays of 640000" % tag, max_abs(w[pick].cpu(), want_w), tol)
    else:
        mse16 = torch.mean((rgb_w[pick].cpu() - want_rgb) ** 2).item()
        print("\nfull-size bf16 spot check ('he' weights): MSE %.2e = %.1f dB" % (mse16, -10 * math.log10(max(mse16, 1e-12))))
        assert mse16 <= BF16_HE_SPOT_MSE


BF16_HE_SPOT_MSE = 1.5e-4                                 # measured on MI355X (round 3): 4.3e-5 (43.6 dB) on the O(1)-activation 'he' weights


# per-stage limits of the full-size bf16 check below: (density, rgb channels of rgbo, sigma channel of rgbo -- each relative to the stage's
# largest |value| -- , final rgb abs, depth abs).  Set from the values the first run on MI355X recorded (profiles/r06_measured_gates.log).
# first run (round 6, one MI355X): small 8.1e-4 / 1.7e-5 / 2.1e-3 / 1.1e-5 / 6.2e-6; he 3.1e-3 / 8.9e-3 / 9.4e-3 (end to end, `he` is judged by MSE only:
# on O(1)-activation weights one inverse-CDF bucket that tips the other way changes a ray's colour by O(1) -- measured max 0.81 on 1 of 2 048 rays)
BF16_STAGE_LIMITS = {"small": (2.5e-3, 1e-4, 6e-3, 1e-4, 1e-4), "he": (1e-2, 3e-2, 3e-2, None, None)}


@pytest.mark.parametrize("tag", ["small", "he"])
def test_full_size_bf16_render_stage_by_stage_against_the_bf16_oracle(A, tag):
    """VERDICT r5 item 8 -- the headline dtype at the headline size.  ONE 800 x 800, 64 + 128 bf16 render through nerf_amd_render_rays;
    on 2 048 random rays of it every stage is compared with the oracle fed the HIP path's OWN previous stage (teacher-forced, so that one
    flipped inverse-CDF bucket does not drown the comparison):
      density  = proposal MLP on the stratified points    vs the oracle's proposal_forward in bf16-operand emulation
      z_fine   = weights -> blur -> inverse-CDF sampling  vs the oracle's fp32 rows 5-7 on the HIP density (fp32 stage: tight)
      rgbo     = fine MLP at the HIP z_fine               vs the oracle's mip_forward in bf16-operand emulation
      rgb / depth / weights = compositing of the HIP rgbo vs the oracle's composite (fp32 stage: tight)
    and the end-to-end image against the oracle's bf16-emulated render of the same rays.  Every figure goes through conftest.gate (value
    on record next to the limit)."""
    prop, mip = build_nets(A, tag)
    P = A.ops.BF16
    H = Wd = 800
    pose = O.pose_spherical(20.0, -30.0, 4.0)[:3]
    f = O.fov2focal(0.6911112070083618, (H, Wd))
    n = H * Wd
    rays = A.ops.generate_rays(pose, H, Wd, f[1], f[0], "cuda", 0, n)
    gen = torch.Generator(device="cuda").manual_seed(19)
    u1 = torch.rand(n, 64, device="cuda", generator=gen)
    u2 = torch.rand(n, 129, device="cuda", generator=gen)
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    rgb, depth, w, ws = A.ops.render_rays(prop.packed(P), mip.packed(P), P, rays, z_base, u1, u2, 128, NEAR, FAR, True, want_depth=True, want_weights=True)
    torch.cuda.synchronize()
    # the intermediates of that launch sequence, in the entry point's workspace: density (N,64) | z_fine (N,129) | rgbo (N,128,4), 256-byte aligned
    base = (ws.data_ptr() + 255) // 256 * 256 - ws.data_ptr()
    al = lambda b: (b + 255) // 256 * 256
    o_d, o_z = base, base + al(n * 64 * 4)
    o_r = o_z + al(n * 129 * 4)
    dens = ws[o_d: o_d + n * 64 * 4].view(torch.float32).view(n, 64)
    z_fine = ws[o_z: o_z + n * 129 * 4].view(torch.float32).view(n, 129)
    rgbo = ws[o_r: o_r + n * 128 * 16].view(torch.float32).view(n, 128, 4)
    pick = torch.randperm(n, generator=torch.Generator().manual_seed(5))[:2048]
    r, a1, a2 = rays[pick].cpu(), u1[pick].cpu(), u2[pick].cpu()
    d_hip, z_hip, c_hip = dens[pick].cpu(), z_fine[pick].cpu(), rgbo[pick].cpu()
    psd, msd = W.proposal_state(tag), W.mip_state(tag)
    lim = BF16_STAGE_LIMITS[tag]
    with torch.no_grad():
        z_c = O.stratified_render(NEAR, FAR, 128, a1)
        pts_c = r[:, None, :3] + z_c[..., None] * r[:, None, 3:]
        d_want = O.proposal_forward(psd, pts_c, emulate_bf16=True)
        gate("full-size bf16 [%s] density vs bf16-emulated oracle, rel. to max |density|" % tag, max_abs(d_hip, d_want) / d_want.abs().max().item(), lim[0])
        z_want, _ = O.inverse_sample(O.max_blur(O.sigma_to_weights(d_hip, z_c, r[:, 3:]), 0.01), z_c, a2, sort=True)
        gate("full-size bf16 [%s] z_fine from the HIP density vs oracle rows 5-7 (fp32)" % tag, max_abs(z_hip, z_want), 2e-5)
        c_want = O.mip_forward(msd, O.length2pts(r, z_hip[:, :-1]), emulate_bf16=True)
        gate("full-size bf16 [%s] rgb of rgbo vs bf16-emulated oracle" % tag, max_abs(c_hip[..., :3], c_want[..., :3]), lim[1])
        gate("full-size bf16 [%s] sigma of rgbo vs bf16-emulated oracle, rel. to max |sigma|" % tag,
             max_abs(c_hip[..., 3], c_want[..., 3]) / c_want[..., 3].abs().max().item(), lim[2])
        rgb_c, w_c, ex = O.composite(c_hip, z_hip[:, :-1], r[:, 3:], white_bkg=True, render_depth=(NEAR, FAR))
        gate("full-size bf16 [%s] rgb from the HIP rgbo vs oracle composite (fp32)" % tag, max_abs(rgb[pick].cpu(), rgb_c), 2e-5)
        gate("full-size bf16 [%s] weights from the HIP rgbo vs oracle composite (fp32)" % tag, max_abs(w[pick].cpu(), w_c), 2e-5)
        gate("full-size bf16 [%s] depth from the HIP rgbo vs oracle composite (fp32)" % tag, max_abs(depth[pick].cpu(), ex["depth_img"]), 1e-4)
        e_rgb, e_w, e_depth = O.render_rays(psd, msd, r, a1, a2, NEAR, FAR, 128, white_bkg=True, emulate_bf16=True)
        if lim[3] is not None:
            gate("full-size bf16 [%s] end-to-end rgb vs the oracle's bf16-emulated render" % tag, max_abs(rgb[pick].cpu(), e_rgb), lim[3])
            gate("full-size bf16 [%s] end-to-end depth vs the oracle's bf16-emulated render" % tag, max_abs(depth[pick].cpu(), e_depth), lim[4])
        mse = torch.mean((rgb[pick].cpu() - e_rgb) ** 2).item()
        # (`he`: 6.8e-4 measured, almost all of it from ~1 ray in 2 048 whose inverse-CDF draw fell into the neighbouring bucket -- an O(1) colour
        #  change on these weights; the teacher-forced stages above are the statement, this figure is on record only)
        gate("full-size bf16 [%s] end-to-end image MSE vs the oracle's bf16-emulated render (dB = -10 log10)" % tag, mse, 1e-6 if tag == "small" else 3e-3)
        off = ((rgb[pick].cpu() - e_rgb).abs().amax(dim=-1) > 0.05).float().mean().item()
        gate("full-size bf16 [%s] fraction of rays further than 0.05 from the oracle's bf16-emulated render" % tag, off, 1e-4 if tag == "small" else 1e-2)


# ------------------------------------------------------------------------------------------------ row 13: Ref-NeRF
def build_ref(A, tag):
    from nerf_amd.ref_model import RefNeRF
    net = RefNeRF(10, 4)
    net.load_state_dict(W.ref_state(tag))
    return net.cuda().eval()


@pytest.mark.parametrize("tag", ["small", "he"])
def test_refnerf_forward_vs_reference_golden(A, golden, tag):
    g = golden("g13_refnerf")
    net = build_ref(A, tag)
    A.pkg.set_precision("fp32")
    with torch.no_grad():
        rgbo, normal = net.forward(dev(g[tag + "_pts"]))
    scale = max(1.0, g[tag + "_rgbo"].abs().max().item())
    assert max_abs(rgbo.cpu(), g[tag + "_rgbo"]) <= 2e-5 * scale
    assert max_abs(normal.cpu(), g[tag + "_normal"]) <= 2e-5
    # split position / direction call form (ref_model.py:68,89)
    with torch.no_grad():
        rgbo2, _ = net.forward(dev(g[tag + "_pts"][..., :3].contiguous()), dev(g[tag + "_pts"][..., 3:].contiguous()))
    assert torch.equal(rgbo, rgbo2)
    # bf16 mode against the bf16-operand emulation of the oracle
    A.pkg.set_precision("bf16")
    with torch.no_grad():
        rgbo16, _ = net.forward(dev(g[tag + "_pts"]))
        want16, _ = O.ref_forward(W.ref_state(tag), g[tag + "_pts"], emulate_bf16=True)
    A.pkg.set_precision("fp32")
    assert max_abs(rgbo16.cpu(), want16) <= 2e-2 * scale


@pytest.mark.parametrize("tag", ["small", "he"])
def test_refnerf_use_srgb_forward_and_gradients(A, golden, tag):
    """RefNeRF(use_srgb=True) (ref_model.py:100-102: sigmoid(diffuse - log 3), linear_to_srgb): the kernel's output transform and its
    backward (NERF_AMD_REF_SRGB) against the real reference's forward and parameter gradients (golden G19); the render path passes the
    flag too."""
    from nerf_amd.ref_model import RefNeRF
    g = golden("g19_refnerf_srgb")
    net = RefNeRF(10, 4, use_srgb=True)
    net.load_state_dict(W.ref_state(tag))
    net = net.cuda().eval()
    A.pkg.set_precision("fp32")
    scale = max(1.0, g[tag + "_rgbo"].abs().max().item())
    with torch.no_grad():
        rgbo, normal = net.forward(dev(g["pts"]))
    assert max_abs(rgbo.cpu(), g[tag + "_rgbo"]) <= 2e-5 * scale
    assert max_abs(normal.cpu(), g[tag + "_normal"]) <= 2e-5
    # gradients: forward with the activation dump, nerf_amd_ref_backward with the flag
    rgbo_t, normal_t = net.forward(dev(g["pts"]))
    assert max_abs(rgbo_t.detach().cpu(), g[tag + "_rgbo"]) <= 2e-5 * scale
    ((rgbo_t * dev(g["g_rgbo"])).sum() + (normal_t * dev(g["g_normal"])).sum()).backward()
    got = dict(net.named_parameters())
    for key, name, rows in (("g_spec", "spec_rgb_head.0.weight", None), ("g_nct", "norm_col_tint_head.weight", None),
                            ("g_nct_bias", "norm_col_tint_head.bias", None), ("g_rho_tau", "rho_tau_head.weight", None),
                            ("g_dir2_6", "dir_block2.6.weight", 8), ("g_spa2_6", "spa_block2.6.weight", 8), ("g_spa0", "spa_block1.0.weight", 8)):
        want = g[tag + "_" + key]
        have = got[name].grad.cpu() if rows is None else got[name].grad[:rows].cpu()
        assert max_abs(have, want) <= 3e-3 * max(1e-6, want.abs().max().item()), (tag, key, max_abs(have, want), want.abs().max().item())
    # bf16 mode against the oracle's bf16-operand emulation
    A.pkg.set_precision("bf16")
    with torch.no_grad():
        rgbo16, _ = net.forward(dev(g["pts"]))
        want16, _ = O.ref_forward(W.ref_state(tag), g["pts"], emulate_bf16=True, use_srgb=True)
    A.pkg.set_precision("fp32")
    assert max_abs(rgbo16.cpu(), want16) <= 2e-2 * scale
    if tag == "small":                                       # the whole-tile render entry (nerf_amd_render_rays_ref) takes the same flag
        prop, _ = build_nets(A, "small")
        rays, u1, u2 = _rays_and_u(300, 64, 19)
        z_base = torch.linspace(NEAR, FAR, 64).cuda()
        with torch.no_grad():
            want_rgb, _, _ = O.render_rays_ref(W.proposal_state("small"), W.ref_state("small"), rays, u1, u2, NEAR, FAR, 64, white_bkg=True,
                                               use_srgb=True)
        rgb, _, _, _ = A.ops.render_rays_ref(prop.packed(A.ops.F32), net.packed(A.ops.F32), A.ops.F32, dev(rays), z_base, dev(u1), dev(u2), 64,
                                             NEAR, FAR, True, flags=net.kernel_flags)
        assert max_abs(rgb.cpu(), want_rgb) <= 1e-4


@pytest.mark.parametrize("M", [1, 33, 129, 1000, 40001])
def test_refnerf_ragged_sizes(A, M):
    net = build_ref(A, "small")
    gen = torch.Generator().manual_seed(M)
    pts = torch.cat(((torch.rand(M, 1, 3, generator=gen) - 0.5) * 8, torch.randn(M, 1, 3, generator=gen)), -1)
    with torch.no_grad():
        want, wn = O.ref_forward(W.ref_state("small"), pts)
        A.pkg.set_precision("fp32")
        got, gn = net.forward(dev(pts))
    assert max_abs(got.cpu(), want) <= 2e-5 * max(1.0, want.abs().max().item()) and max_abs(gn.cpu(), wn) <= 2e-5


@pytest.mark.parametrize("M", [1, 33, 300, 1000])
def test_refnerf_backward_ragged_sizes(A, M):
    """The fused backward chains work on 256-sample tiles of 32-sample subtiles: sample counts that end inside a subtile / a tile.  Every
    parameter gradient and RefNeRF.get_grad (the density-gradient chain) of the fp32 kernels against fp64 autograd of the oracle's forward
    (ref_model.py:68-125) on the same points."""
    from nerf_amd.ref_model import RefNeRF
    net = build_ref(A, "small")                          # (eval mode: no bottle-neck noise; the training forward runs because gradients are asked for)
    gen = torch.Generator().manual_seed(500 + M)
    pts = torch.cat(((torch.rand(M, 1, 3, generator=gen) - 0.5) * 6, torch.nn.functional.normalize(torch.randn(M, 1, 3, generator=gen), dim=-1)), -1)
    g_rgbo, g_nrm = torch.randn(M, 1, 4, generator=gen), torch.randn(M, 1, 3, generator=gen)
    A.pkg.set_precision("fp32")
    pos = dev(pts[..., :3]).contiguous().requires_grad_(True)
    rgbo, nrm = net.forward(pos, dev(pts[..., 3:]).contiguous())
    dgrad = RefNeRF.get_grad(rgbo[..., -1], pos)
    ((rgbo * dev(g_rgbo)).sum() + (nrm * dev(g_nrm)).sum()).backward()
    sd = {k: v.double().requires_grad_(True) for k, v in W.ref_state("small").items()}
    p64 = pts[..., :3].double().requires_grad_(True)
    want, wn = O.ref_forward(sd, torch.cat((p64, pts[..., 3:].double()), -1))
    dg64, = torch.autograd.grad(want[..., -1].sum(), p64, retain_graph=True)
    dg64 = dg64 / torch.clamp(dg64.norm(dim=-1, keepdim=True), min=1e-5)
    ((want * g_rgbo.double()).sum() + (wn * g_nrm.double()).sum()).backward()
    assert max_abs(rgbo.detach().cpu().double(), want.detach()) <= 2e-5 * max(1.0, want.abs().max().item())
    assert max_abs(dgrad.cpu().double(), dg64) <= 2e-3
    worst = {}
    for name, p in net.named_parameters():
        ex = sd[name].grad
        top = max(ex.abs().max().item(), 1e-12)
        worst[name] = (p.grad.detach().cpu().double() - ex).abs().max().item() / top
        assert worst[name] <= 2e-3, (M, name, worst[name], top)
    print("\nRef-NeRF backward, M = %d: worst parameter-gradient error relative to the fp64 value %.1e" % (M, max(worst.values())))


def test_render_image_refnerf_vs_reference(A, golden):
    """Drop-in surface with a RefNeRF: the reference's own image (coarse+fine merge, softplus(sigma+.5), normal map)."""
    g = golden("g13_refnerf")
    prop, _ = build_nets(A, "small")
    net = build_ref(A, "small")
    A.pkg.set_precision("fp32")
    torch.manual_seed(4321)
    with torch.no_grad():
        res = A.procedures.render_image(net, prop, dev(g["pose"]), 50, tuple(g["img_focal"].tolist()), NEAR, FAR, 64, white_bkg=True, rng="reference",
                                        render_depth=True, render_normal=True)
    assert list(res.keys()) == ["rgb", "depth_img", "normal_img"]
    assert max_abs(res["rgb"].cpu(), g["img_rgb"]) <= 1e-4
    assert max_abs(res["depth_img"][0].cpu(), g["img_depth"]) <= 1e-4
    assert max_abs(res["normal_img"][0].cpu(), g["img_normal"]) <= 1e-4


# ------------------------------------------------------------------------------------------------ training step (autograd bridge)
def test_train_step_gradients(A, golden):
    """train.py:164-199 (non-ref) through the nerf_amd surface: HIP forward, device-side torch VJP backward.  Losses and
    parameter gradients against the REAL reference's (golden G14)."""
    g = golden("g14_train_step")
    prop, mip = build_nets(A, "small")
    prop.train(); mip.train()
    A.pkg.set_precision("fp32")
    rays, z_c, tgt = dev(g["rays"]), dev(g["z_coarse"]), dev(g["rgb_tgt"])
    pts = (rays[:, None, :3] + rays[:, None, 3:] * z_c[:, :, None]).contiguous()
    density = F.softplus(prop.forward(pts))
    pw = A.mip_methods.maxBlurFilter(A.addtional.ProposalNetwork.get_weights(density, z_c, rays[:, 3:]), 0.01)
    fl, below = A.utils.inverseSample(pw, z_c, 65, sort=True, u=g["u_inv"])
    fl = fl[..., :-1].contiguous()
    rgbo = mip.forward(A.nerf_base.NeRF.length2pts(rays, fl))
    rend, wts, _ = A.nerf_base.NeRF.render(rgbo, fl, rays[:, 3:])
    bounds = A.addtional.getBounds(pw, below)
    img_loss = torch.nn.MSELoss()(rend, tgt)
    p_loss = A.addtional.ProposalLoss()(bounds, wts.detach())
    (p_loss + img_loss).backward()
    assert max_abs(rend.detach().cpu(), g["rendered"]) <= 1e-5 and max_abs(wts.detach().cpu(), g["weights"]) <= 1e-5
    assert abs(img_loss.item() - g["img_loss"]) <= 1e-6 and abs(p_loss.item() - g["prop_loss"]) <= 1e-4 * max(1.0, g["prop_loss"])

    def rel(got, want):
        return max_abs(got.cpu(), want) / max(want.abs().max().item(), 1e-30)
    errs = {"mip_l1": rel(mip.lin_block1[0].weight.grad[:8], g["g_mip_l1"]), "mip_rgb": rel(mip.rgb_layer[2].weight.grad, g["g_mip_rgb"]),
            "mip_sigma": rel(mip.opacity_head[0].weight.grad, g["g_mip_sigma"]), "prop_l0": rel(prop.layers[0].weight.grad[:8], g["g_prop_l0"]),
            "prop_head": rel(prop.layers[8].weight.grad, g["g_prop_head"])}
    # first-layer gradients are sums of ~2000 sign-alternating terms of size ~1e-9 (they nearly cancel): their fp32
    # summation-order noise is a few % of the largest entry; the head gradients agree to 1e-6
    tol = {"mip_l1": 5e-2, "prop_l0": 5e-2, "mip_sigma": 5e-3, "mip_rgb": 1e-4, "prop_head": 1e-4}
    assert all(errs[k] <= tol[k] for k in errs), errs
    # ... and that statement is measured, not assumed: the same step evaluated in fp64 (oracle + torch.autograd on the CPU, same fine
    # depths and bin indices) is the exact value; the REFERENCE's fp32 gradients (the golden) sit a few % from it on the cancelling
    # tensors, and the HIP gradients are no further from it than the reference is (factor 2).
    d64 = lambda sd: {k: v.double().requires_grad_(True) for k, v in sd.items()}
    p64, m64 = d64(W.proposal_state("small")), d64(W.mip_state("small"))
    r64, z64, fl64 = g["rays"].double(), g["z_coarse"].double(), fl.detach().cpu().double()
    dens64 = F.softplus(O.proposal_forward(p64, r64[:, None, :3] + r64[:, None, 3:] * z64[:, :, None]))
    pw64 = O.max_blur(O.sigma_to_weights(dens64, z64, r64[:, 3:]), 0.01)
    rgbo64 = O.mip_forward(m64, O.length2pts(r64, fl64))
    rend64, wts64, _ = O.composite(rgbo64, fl64, r64[:, 3:])
    loss64 = torch.mean((rend64 - g["rgb_tgt"].double()) ** 2) + O.proposal_loss(O.get_bounds(pw64, below.cpu()), wts64.detach())
    loss64.backward()
    exact = {"mip_l1": m64["lin_block1.0.weight"].grad[:8], "mip_rgb": m64["rgb_layer.2.weight"].grad, "mip_sigma": m64["opacity_head.0.weight"].grad,
             "prop_l0": p64["layers.0.weight"].grad[:8], "prop_head": p64["layers.8.weight"].grad}
    have = {"mip_l1": mip.lin_block1[0].weight.grad[:8], "mip_rgb": mip.rgb_layer[2].weight.grad, "mip_sigma": mip.opacity_head[0].weight.grad,
            "prop_l0": prop.layers[0].weight.grad[:8], "prop_head": prop.layers[8].weight.grad}
    gold = {"mip_l1": g["g_mip_l1"], "mip_rgb": g["g_mip_rgb"], "mip_sigma": g["g_mip_sigma"], "prop_l0": g["g_prop_l0"], "prop_head": g["g_prop_head"]}
    for k in exact:
        top = exact[k].abs().max().item()
        ref_exact = (gold[k].double() - exact[k]).abs().max().item() / top
        hip_exact = (have[k].detach().cpu().double() - exact[k]).abs().max().item() / top
        assert hip_exact <= max(2.0 * ref_exact, 2e-5), (k, hip_exact, ref_exact)
    print("\nG14 gradients, max error relative to the fp64 value:", {k: "ref %.1e hip %.1e" % (
        (gold[k].double() - exact[k]).abs().max().item() / exact[k].abs().max().item(),
        (have[k].detach().cpu().double() - exact[k]).abs().max().item() / exact[k].abs().max().item()) for k in exact})
    # an optimiser step invalidates the packed weights; the next forward must see the new parameters
    opt = torch.optim.Adam(list(mip.parameters()) + list(prop.parameters()), lr=1e-3)
    before = mip.forward(dev(g["rays"][:4, None, :].repeat(1, 3, 1))).detach().clone()
    opt.step()
    after = mip.forward(dev(g["rays"][:4, None, :].repeat(1, 3, 1))).detach()
    assert not torch.equal(before, after)


@pytest.mark.parametrize("n_fine,n", [(128, 777), (64, 1000), (32, 513)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_fused_compositing_epilogue_equals_two_launches(A, n_fine, n, prec):
    """nerf_amd_mip_forward_composite (fine MLP + in-kernel wave-prefix-product compositing by the last wavefront of each
    ray) against MipNeRF.forward followed by NeRF.render; ragged ray counts exercise the tile tail."""
    _, mip = build_nets(A, "he")
    P = A.ops.F32 if prec == "fp32" else A.ops.BF16
    gen = torch.Generator().manual_seed(n)
    pose = O.pose_spherical(10.0, -30.0, 4.0)[:3]
    dirs = O.ray_dirs_image(pose, 40, 40, O.fov2focal(0.6911112070083618, (40, 40))).reshape(-1, 3)
    rays = torch.cat((pose[:, -1].expand(n, -1), dirs[torch.randint(0, 1600, (n,), generator=gen)]), -1).contiguous().cuda()
    z = torch.sort(NEAR + (FAR - NEAR) * torch.rand(n, n_fine + 1, generator=gen), dim=-1)[0].cuda()
    for wb in (False, True):
        rgb, depth, w = A.ops.mip_forward_composite(mip.packed(P), P, rays, z, n_fine, wb, NEAR, FAR, want_depth=True, want_weights=True)
        rgbo = A.ops.mip_forward_samples(mip.packed(P), P, A.ops.samples_rays(rays, n_fine, z=z), (n, n_fine), "cuda")
        rgb2, w2, depth2, _ = A.ops.composite(rgbo, z, rays, True, wb, A.ops.ACT_RELU, (NEAR, FAR))
        assert max_abs(rgb, rgb2) <= 2e-6 and max_abs(depth, depth2) <= 2e-6 and max_abs(w, w2) <= 1e-6


@pytest.mark.parametrize("K,kind", [(129, "uniform"), (193, "uniform"), (65, "uniform"), (129, "clustered"), (129, "constant"), (300, "uniform")])
def test_inverse_sampling_sort_paths(A, K, kind):
    """The bucket-rank sort (uniform u) and its O(K^2) fallback (clustered / constant u overflow a bucket) against torch.sort."""
    gen = torch.Generator().manual_seed(K)
    N, C = 37, 64
    w = torch.rand(N, C, generator=gen) ** 3 + 0.01
    z = torch.sort(NEAR + (FAR - NEAR) * torch.rand(N, C, generator=gen), dim=-1)[0]
    if kind == "uniform":
        u = torch.rand(N, K, generator=gen)
    elif kind == "clustered":
        u = 0.5 + 1e-4 * torch.rand(N, K, generator=gen)
    else:
        u = torch.full((N, K), 0.25)
    want_z, want_b = O.inverse_sample(w, z, u, sort=True)
    got_z, got_b = A.utils.inverseSample(dev(w), dev(z), K, sort=True, u=u)
    assert bool((got_z[:, 1:] >= got_z[:, :-1]).all())
    assert max_abs(got_z.cpu(), want_z) <= 2e-6
    if kind != "constant":
        assert torch.equal(got_b.cpu(), want_b)
    else:                                                     # all samples tie: any permutation of equal depths is a valid sort
        assert torch.equal(torch.sort(got_b.cpu(), -1)[0], torch.sort(want_b, -1)[0])


def test_inverse_sampling_with_unsorted_depths_sorts_the_values(A):
    """inverseSample(sort=True) on depths that are NOT ascending (the reference's torch.sort does not care): the order of the samples
    is then not the order of their uniforms, the kernel must detect it and sort the values."""
    gen = torch.Generator().manual_seed(77)
    N, C, K = 41, 64, 129
    w = torch.rand(N, C, generator=gen) + 0.05
    z = NEAR + (FAR - NEAR) * torch.rand(N, C, generator=gen)                 # unsorted bin edges
    u = torch.rand(N, K, generator=gen)
    want_z, _ = O.inverse_sample(w, z, u, sort=True)
    got_z, _ = A.utils.inverseSample(dev(w), dev(z), K, sort=True, u=u)
    assert bool((got_z[:, 1:] >= got_z[:, :-1]).all())
    assert max_abs(got_z.cpu(), want_z) <= 2e-6


# ------------------------------------------------------------------------------------------------ robustness
def test_non_contiguous_inputs_and_side_stream(A):
    """Views / strided tensors are accepted (made contiguous by the host layer) and the kernels run on the caller's
    current HIP stream."""
    prop, mip = build_nets(A, "small")
    A.pkg.set_precision("fp32")
    gen = torch.Generator().manual_seed(5)
    big = torch.randn(64, 20, 12, generator=gen).cuda()
    pts_view = big[::2, :, 3:9]                                      # non-contiguous (32, 20, 6)
    with torch.no_grad():
        a = mip.forward(pts_view)
        b = mip.forward(pts_view.contiguous())
        assert torch.equal(a, b)
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            c = mip.forward(pts_view)
            d = A.nerf_helper.positional_encoding(big[..., :3], 4)
        st.synchronize()
        assert torch.equal(a, c) and d.shape == (64, 20, 24)
    z = torch.sort(torch.rand(9, 70, generator=gen) * 4 + 2, dim=-1)[0].cuda()
    rgbo = torch.rand(9, 70, 4, generator=gen).cuda()
    dirs6 = torch.randn(9, 6, generator=gen).cuda()
    r1, w1, _ = A.nerf_base.NeRF.render(rgbo, z, dirs6[:, 3:])      # strided direction view
    r2, w2, _ = A.nerf_base.NeRF.render(rgbo, z, dirs6[:, 3:].contiguous())
    assert torch.equal(r1, r2) and torch.equal(w1, w2)


def test_kernels_launch_on_torchs_current_stream(A):
    """ops._stream() takes the raw handle of torch's CURRENT stream from the C extension (round 6: torch.cuda.current_stream() cost ~10 us per
    call, thirteen calls per small training step): it must follow `torch.cuda.stream(...)` contexts and graph capture exactly like the
    Stream object did -- and a kernel launched inside a side-stream context must be ordered on that stream."""
    assert A.ops._stream() == torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert A.ops._stream() == side.cuda_stream == torch.cuda.current_stream().cuda_stream
        x = torch.rand(1000, 3, device="cuda")
        y = A.ops.positional_encoding(x, 4)
        ev = torch.cuda.Event()
        ev.record(side)
    ev.synchronize()
    assert A.ops._stream() == torch.cuda.current_stream().cuda_stream != side.cuda_stream
    assert max_abs(y.cpu(), O.positional_encoding(x.cpu(), 4)) <= 2e-6


def test_error_reporting(A):
    from nerf_amd._lib import NerfAmdError
    with pytest.raises(NerfAmdError):                                  # C < 3 is rejected by the C-ABI with a message
        A.ops.inverse_sample(torch.rand(4, 2).cuda(), torch.rand(4, 2).cuda(), torch.rand(4, 8).cuda(), True)
    with pytest.raises(NerfAmdError):
        A.ops.mip_forward_composite(build_nets(A, "small")[1].packed(A.ops.F32), A.ops.F32, torch.rand(4, 6).cuda(),
                                    torch.rand(4, 101).cuda(), 100, False, 2.0, 6.0)       # S not in {32, 64, 128}
    with torch.no_grad():                                              # integrated PE WITH contraction layer by layer (round 6; used to raise)
        assert A.mip_model.MipNeRF(10, 4, 512).cuda().eval().forward_rays(torch.rand(2, 6).cuda(), torch.rand(2, 5).cuda().sort(-1)[0], 4, ipe_radius=1e-3,
                                                                            contract=True).shape == (2, 4, 4)
    assert A.addtional.ProposalNetwork(10, 512).cuda().eval().forward(torch.rand(2, 3, 3).cuda() * 5, contract=True).shape == (2, 3)   # (round 5)
    assert A.addtional.ProposalNetwork(10, 512).cuda().eval().forward(torch.rand(2, 3, 3).cuda()).shape == (2, 3)   # wider than compiled: generic path
    assert A.addtional.ProposalNetwork(10).cuda().eval().forward(torch.rand(2, 3, 3).cuda()).shape == (2, 3)   # class default 128: zero-padded


def test_render_only_cli(A, tmp_path):
    """procedures.render_only (procedures.py:99-164): checkpoint files -> test-set poses -> PNGs (rgb | ground truth)."""
    import json
    import numpy as np
    from PIL import Image
    from nerf_amd.nerf_helper import saveModel
    from nerf_amd.procedures import get_parser, render_only
    root = str(tmp_path)
    scene = os.path.join(root, "data", "toy")
    os.makedirs(os.path.join(scene, "test"))
    frames = []
    for k in range(2):
        Image.fromarray(np.full((50, 50, 3), 255, np.uint8)).save(os.path.join(scene, "test", "r_%d.png" % k))
        pose = A.utils.pose_spherical(40.0 * k, -30.0, 4.0)
        frames.append({"file_path": "./test/r_%d" % k, "transform_matrix": pose.tolist()})
    json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, open(os.path.join(scene, "transforms_test.json"), "w"))
    prop, mip = build_nets(A, "small")
    os.makedirs(os.path.join(root, "model"))
    saveModel(mip, os.path.join(root, "model", "toy_mip.pth"))
    saveModel(prop, os.path.join(root, "model", "toy_prop.pth"))
    args = get_parser().parse_args(["--name", "toy", "--dataset_name", "toy", "--img_scale", "1.0", "--opt_mode", "none", "-e", "-w"])
    torch.manual_seed(3)
    render_only(args, os.path.join(root, "model") + "/", "O1", dataset_root=os.path.join(root, "data") + "/", output_root=os.path.join(root, "out") + "/")
    for k in range(2):
        im = np.asarray(Image.open(os.path.join(root, "out", "given", "result_%03d.png" % k)))
        assert im.shape == (54, 2 * 52 + 2, 3)                               # rgb | gt with 2-pixel padding
        assert (im[2:52, 54:104] == 255).all()                               # the ground-truth panel
    # the rgb panel equals a direct render_image call with the same RNG state
    torch.manual_seed(3)
    with torch.no_grad():
        focal = A.utils.fov2Focal(0.6911112070083618, (50, 50))
        direct = A.procedures.render_image(mip.eval(), prop.eval(), torch.tensor(frames[0]["transform_matrix"])[:3].cuda(), (50, 50), focal, 2.0, 6.0,
                                           128, white_bkg=True)["rgb"]
    first = torch.from_numpy(np.asarray(Image.open(os.path.join(root, "out", "given", "result_000.png")))[2:52, 2:52].astype(np.float32) / 255.0)
    assert (first.permute(2, 0, 1) - direct.cpu().clamp(0, 1)).abs().max() <= 0.5 / 255 + 1e-4


# ------------------------------------------------------------------------------------------------ HIP backward
def _vjp(fn, grad, *args):
    leaves = [a.detach().clone().requires_grad_(True) if (isinstance(a, torch.Tensor) and a.is_floating_point()) else a for a in args]
    with torch.enable_grad():
        y = fn(*leaves)
    return torch.autograd.grad(y, [l for l in leaves if isinstance(l, torch.Tensor) and l.requires_grad], grad, allow_unused=True)


@pytest.mark.parametrize("S,act", [(64, 0), (128, 0), (129, 2), (200, 1), (7, 0), (256, 0), (257, 2), (320, 2), (513, 0), (1024, 2)])
def test_weights_and_composite_backward_vs_autograd(A, S, act):
    # (rows above 256 samples: 8 / 16 register chunks since round 5 -- `-t --fine_sample_pnum 256` merges 320 samples per ray)
    import torch_spec as ab
    gen = torch.Generator().manual_seed(S)
    N = 37
    sigma = (torch.randn(N, S, generator=gen) * 2.0).cuda()
    sigma[0, :5] = 0.0                                                             # relu'(0) = 0
    if act == 1:
        sigma = sigma.abs()                                                        # no activation: exp(-sigma * 1e10) must stay finite
    z = torch.sort(torch.rand(N, S, generator=gen) * 4 + 2, dim=-1)[0].cuda()
    g = torch.randn(N, S, generator=gen).cuda()
    want, = _vjp(lambda s: ab.weights_expr(s, z, act), g, sigma)
    got = A.ops.sigma_to_weights_backward(sigma, z, None, act, g)
    assert max_abs(got.cpu(), want.cpu()) <= 2e-5 * max(1.0, want.abs().max().item())
    # compositing: d(rgb)/d(rgbo) with white background and |d| scaling, Ref-NeRF style shift for the softplus case
    rgbo = torch.cat((torch.rand(N, S, 3, generator=gen), torch.randn(N, S, 1, generator=gen) * 2), -1).cuda()
    if act == 1:
        rgbo = rgbo.abs()
    dirs = (torch.randn(N, 3, generator=gen) * 0.3 + torch.tensor([0.0, 0.0, -1.0])).cuda()
    d_rgb = torch.randn(N, 3, generator=gen).cuda()
    shift = 0.5 if act == 2 else 0.0

    def expr(r):
        zz = z * dirs.norm(dim=-1, keepdim=True)
        w_ = ab.weights_expr(r[..., 3] + shift, zz, act)
        c = torch.sum(w_[:, :, None] * r[..., :3], dim=-2)
        return c + (1.0 - torch.sum(w_, -1)[..., None])
    want, = _vjp(expr, d_rgb, rgbo)
    got = A.ops.composite_backward(rgbo, z, dirs, True, True, act, None, d_rgb, None, None, sigma_shift=shift)
    assert max_abs(got.cpu(), want.cpu()) <= 2e-5 * max(1.0, want.abs().max().item())
    # all three upstream gradients at once (weights and depth too)
    d_w, d_dep = torch.randn(N, S, generator=gen).cuda(), torch.randn(N, generator=gen).cuda()

    def expr3(r):
        zz = z * dirs.norm(dim=-1, keepdim=True)
        w_ = ab.weights_expr(r[..., 3] + shift, zz, act)
        c = torch.sum(w_[:, :, None] * r[..., :3], dim=-2)
        dep = (torch.sum(w_ * zz, -1) - 2.0) / (6.0 - 2.0)
        return (c * d_rgb).sum() + (w_ * d_w).sum() + (dep * d_dep).sum()
    want, = _vjp(expr3, torch.ones((), device="cuda"), rgbo)
    got = A.ops.composite_backward(rgbo, z, dirs, True, False, act, (2.0, 6.0), d_rgb, d_w, d_dep, sigma_shift=shift)
    assert max_abs(got.cpu(), want.cpu()) <= 3e-5 * max(1.0, want.abs().max().item())


def test_blur_and_bounds_backward_vs_autograd(A):
    import torch_spec as ab
    gen = torch.Generator().manual_seed(11)
    w = torch.rand(53, 64, generator=gen).cuda()
    w[0, 3] = w[0, 4]                                                               # a tie: the gradient splits in halves
    g = torch.randn(53, 64, generator=gen).cuda()
    want, = _vjp(lambda x: ab.max_blur_expr(x, 0.01), g, w)
    assert max_abs(A.ops.max_blur_backward(w, g).cpu(), want.cpu()) <= 1e-6
    below = torch.sort(torch.randint(0, 62, (53, 129), generator=gen), dim=-1)[0].cuda()
    gb = torch.randn(53, 128, generator=gen).cuda()
    want, = _vjp(lambda x: ab.bounds_expr(x, below), gb, w)
    assert max_abs(A.ops.get_bounds_backward(below, gb, 64).cpu(), want.cpu()) <= 2e-5 * max(1.0, want.abs().max().item())
    unsorted = torch.randint(0, 62, (53, 129), generator=gen).cuda()               # the reference never sorts `below` itself
    want, = _vjp(lambda x: ab.bounds_expr(x, unsorted), gb, w)
    assert max_abs(A.ops.get_bounds_backward(unsorted, gb, 64).cpu(), want.cpu()) <= 2e-5 * max(1.0, want.abs().max().item())


def test_integration_md_ctypes_stub_runs(A):
    """The ctypes stub printed in INTEGRATION.md (section B) is executed as is against the built library and must agree with
    the package's own bindings -- the document cannot drift from the C-ABI."""
    import re
    from nerf_amd._lib import LIB_PATH
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    block = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "ctypes stub" in b][0]
    ns = {}
    exec(block.replace('C.CDLL("libnerf_amd.so")', "C.CDLL(%r)" % LIB_PATH), ns)
    gen = torch.Generator().manual_seed(21)
    N, S = 19, 64
    rgbo = torch.cat((torch.rand(N, S, 3, generator=gen), torch.randn(N, S, 1, generator=gen)), -1).cuda()
    z = torch.sort(torch.rand(N, S, generator=gen) * 4 + 2, dim=-1)[0].cuda()
    d = torch.randn(N, 3, generator=gen).cuda()
    rgb, w, extras = ns["render"](rgbo, z, d, True, True, (2.0, 6.0))
    rgb2, w2, dep2, _ = A.ops.composite(rgbo, z, d, True, True, A.ops.ACT_RELU, (2.0, 6.0))
    assert torch.equal(rgb, rgb2) and torch.equal(w, w2) and torch.equal(extras["depth_img"], dep2)
    g = torch.randn(N, 3, generator=gen).cuda()
    assert torch.equal(ns["render_backward"](rgbo, z, d, g, True, True), A.ops.composite_backward(rgbo, z, d, True, True, A.ops.ACT_RELU, None, g, None, None))
    wts, u = torch.rand(N, S, generator=gen).cuda(), torch.rand(N, 33, generator=gen).cuda()
    zs, below = ns["inverse_sample"](wts, z, u, True)
    zs2, below2 = A.ops.inverse_sample(wts, z, u, True)
    assert torch.equal(zs, zs2) and torch.equal(below, below2)
    _, mip = build_nets(A, "small")
    pts = torch.randn(5, 40, 6, generator=gen).cuda()
    blob = ns["pack_mip"](mip, A.ops.F32)
    with torch.no_grad():
        A.pkg.set_precision("fp32")
        assert torch.equal(ns["mip_forward"](blob, A.ops.F32, pts), mip.forward(pts))
        # second block of the document: training forward + hand-written backward through the C-ABI == the package's own path
        from nerf_amd import mlp_backward
        exec([b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "def mip_train_forward" in b][0], ns)
        out, dump = ns["mip_train_forward"](blob, A.ops.F32, pts)
        assert torch.equal(out, mip.forward(pts))
        layers = mip._linear_layers()
        ws, bs = [l.weight.detach().contiguous() for l in layers], [l.bias.detach().contiguous() for l in layers]
        g = torch.randn(5, 40, 4, generator=gen).cuda()
        gw, gb = ns["mip_backward"](mip, A.ops.F32, g, out, dump, ws, bs)
        gw2, gb2 = mlp_backward.mip_backward(g, out, pts, dump, A.ops.F32, ws, bs)
        assert all(torch.equal(a, b) for a, b in zip(gw + gb, gw2 + gb2))


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_mlp_training_forward_and_backward(A, prec):
    """nerf_amd_*_forward_train == the inference kernels; the activation dump read back as rows == the torch layer outputs; the
    GEMM-chain backward == torch.autograd of the reference expressions (fp32 tight; bf16 within operand rounding)."""
    from nerf_amd import mlp_backward
    import torch_spec as ab
    prop, mip = build_nets(A, "he" if prec == "fp32" else "small")     # stress weights where the comparison is exact arithmetic
    A.pkg.set_precision(prec)
    P = A.ops.current_precision()
    gen = torch.Generator().manual_seed(31)
    M = 1000                                                                       # not a multiple of the 256-sample tile
    pts3 = (torch.rand(M, 3, generator=gen) * 2 - 1).cuda()
    pts6 = torch.cat((pts3, torch.randn(M, 3, generator=gen).cuda()), -1).contiguous()
    tol = 2e-4 if prec == "fp32" else 4e-2
    with torch.no_grad():
        dens, dump_p = A.ops.proposal_forward_train(prop.packed(P), P, pts3)
        assert torch.equal(dens, A.ops.proposal_forward(prop.packed(P), P, pts3))
        rgbo, dump_m = A.ops.mip_forward_train(mip.packed(P), P, pts6)
        assert torch.equal(rgbo, A.ops.mip_forward(mip.packed(P), P, pts6))
        # hidden activations of the proposal net, layer by layer, against torch
        wl = [l.weight for l in prop._linear_layers()]
        bl = [l.bias for l in prop._linear_layers()]
        h = torch.cat((pts3, ab._pe(pts3, 10)), -1)
        for l in range(4):
            h = F.relu(F.linear(h, wl[l], bl[l]))
            got = A.ops.train_dump_rows(dump_p, A.ops.NET_PROPOSAL, P, M, l, 256).float()
            assert max_abs(got.cpu(), h.cpu()) <= tol * max(1.0, h.abs().max().item()), l
    # parameter gradients through the module surface (HipOp + mlp_backward) vs torch.autograd on the expressions
    g1 = torch.randn(M, generator=gen).cuda()
    g4 = torch.randn(M, 4, generator=gen).cuda()
    for net, pts, g, expr, n in ((prop, pts3, g1, ab.proposal_expr, 5), (mip, pts6, g4, ab.mip_expr, 11)):
        layers = net._linear_layers()
        params = [l.weight for l in layers] + [l.bias for l in layers]
        for p_ in params:
            p_.grad = None
        net.train()
        out = net.forward(pts.view(M // 8, 8, -1))
        out.backward(g.view(out.shape))
        got = [p_.grad.clone() for p_ in params]
        leaves = [p_.detach().clone().requires_grad_(True) for p_ in params]
        y = expr(pts, leaves[:n], leaves[n:])
        want = torch.autograd.grad(y, leaves, g.view(y.shape))
        for k, (a_, b_) in enumerate(zip(got, want)):
            scale = max(1e-6, b_.abs().max().item())
            # (fp32: reduction order of the M-long wgrad sums; the first layer's gradient is the difference of large terms)
            if prec == "fp32":
                assert max_abs(a_.cpu(), b_.cpu()) <= 5e-3 * scale, (type(net).__name__, k, max_abs(a_.cpu(), b_.cpu()), scale)
            else:
                # bf16 vs an fp32 forward: ReLU masks near zero flip (a forward-precision effect), which alone moves bias sums by >10 %:
                # only the direction of every gradient tensor is checked here; the arithmetic of the chain itself is checked below
                cos = F.cosine_similarity(a_.reshape(1, -1), b_.reshape(1, -1)).item()
                assert cos >= 0.97, (type(net).__name__, k, cos)
    if prec == "bf16":
        # the bf16 GEMM chain against the same chain in fp32 ON THE SAME DUMP (identical masks): operand rounding only
        wl = [l.weight.detach() for l in prop._linear_layers()]
        acts = [A.ops.train_dump_rows(dump_p, A.ops.NET_PROPOSAL, P, M, l, 256).float() for l in range(4)]
        delta = (g1[:, None] * wl[4]) * (acts[3] > 0)
        want_w = [None] * 4
        for l in (3, 2, 1):
            want_w[l] = delta.t() @ acts[l - 1]
            delta = (delta @ wl[l]) * (acts[l - 1] > 0)
        want_w[0] = delta.t() @ torch.cat((pts3, ab._pe(pts3, 10)), -1)
        gW, _ = mlp_backward.proposal_backward(g1, pts3, dump_p, P, wl)
        for l in range(4):
            rel = ((gW[l] - want_w[l]).norm() / want_w[l].norm()).item()
            assert rel <= 2e-2, (l, rel)
    A.pkg.set_precision("fp32")


def test_get_grad_of_proposal_density_then_parameter_backward(A):
    """train.py:165-168 with prop_normal: positions require grad, RefNeRF.get_grad(density, positions) is taken with
    retain_graph, and the SAME graph is backpropagated to the parameters afterwards."""
    import torch_spec as ab
    prop, _ = build_nets(A, "small")
    A.pkg.set_precision("fp32")
    prop.train()
    gen = torch.Generator().manual_seed(41)
    pts = (torch.rand(12, 16, 3, generator=gen) * 2 - 1).cuda().requires_grad_(True)
    dens = prop.forward(pts)
    from nerf_amd.ref_model import RefNeRF
    normals = RefNeRF.get_grad(dens, pts)
    layers = prop._linear_layers()
    params = [l.weight for l in layers] + [l.bias for l in layers]
    for p_ in params:
        p_.grad = None
    F.softplus(dens).sum().backward()
    leaves = [p_.detach().clone().requires_grad_(True) for p_ in params]
    x = pts.detach().clone().requires_grad_(True)
    y = ab.proposal_expr(x, leaves[:5], leaves[5:])
    g, = torch.autograd.grad(y, x, torch.ones_like(y), retain_graph=True)
    want_n = g / torch.maximum(torch.full_like(g[..., :1], 1e-5), g.norm(dim=-1, keepdim=True))
    assert max_abs(normals.cpu(), want_n.cpu()) <= 1e-4
    want = torch.autograd.grad(F.softplus(y).sum(), leaves)
    for a_, b_ in zip([p_.grad for p_ in params], want):
        assert max_abs(a_.cpu(), b_.cpu()) <= 1e-4 * max(1.0, b_.abs().max().item())


def _normals_close(name, got, want, x, g64, worst=5e-2):
    """unit normals against the fp64 value: every sample within 2e-3 except isolated ones where a ReLU whose pre-activation is within fp32
    rounding of zero falls on the other side in the kernel (the gradient w.r.t. ONE sample's position has no sum over samples to hide it
    in; see G17_FP64_GATE) -- at most 2 % of the samples, none beyond `worst`; the offenders are printed."""
    err = (got - want).abs().amax(dim=-1).reshape(-1)
    bad = torch.nonzero(err > 2e-3).reshape(-1)
    r = x.reshape(-1, 3).norm(dim=-1)
    info = [(int(i), round(float(r[i]), 3), float(err[i]), float(g64.reshape(-1, 3)[i].norm())) for i in bad[:12]]
    if len(bad):
        print("contracted normals (%s): %d of %d samples beyond 2e-3 (index, |x|, error, |g64|): %s" % (name, len(bad), err.numel(), info))
    assert len(bad) <= 0.02 * err.numel(), (name, len(bad), info)
    gate("contracted density-gradient normals (%s): worst sample vs fp64" % name, float(err.max()), worst)


def test_scene_contraction_on_the_layer_by_layer_route(A):
    """Round 5: scene contraction used to raise for every network outside the compiled shapes.  It is a stage in front of the encoder there
    (nerf_amd_contract_positions; RefNeRF.get_grad pulls the encoding's gradient back through the contraction's Jacobian).  Width-320
    networks (generic path) with contract=True: the stage itself and its pull-back against oracle.contract / fp64 autograd; proposal
    density, its density-gradient normals and the MipNeRF output against the fp64 oracle on contracted positions; and the networks' whole
    render_image(contract=True) against oracle.render_rays(contracted=True) on the Philox uniforms."""
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.mip_model import MipNeRF
    from nerf_amd.procedures import render_image
    from nerf_amd.ref_model import RefNeRF
    A.pkg.set_precision("fp32")
    gen = torch.Generator().manual_seed(91)
    M = 400
    x = torch.randn(M, 3, generator=gen) * torch.linspace(0.2, 8.0, M)[:, None]
    g = torch.randn(M, 3, generator=gen)
    x64 = x.double().requires_grad_(True)
    c64 = O.contract(x64)
    (c64 * g.double()).sum().backward()
    assert max_abs(A.ops.contract_positions(dev(x)).cpu(), c64.detach()) <= 2e-6
    assert max_abs(A.ops.contract_positions(dev(x), grad=dev(g)).cpu(), x64.grad) <= 2e-6 * max(1.0, x64.grad.abs().max().item())
    torch.manual_seed(17)
    prop, mip = ProposalNetwork(10, 320), MipNeRF(10, 4, 320)
    prop, mip = prop.cuda().eval(), mip.cuda().eval()
    assert prop._generic() and mip._generic()
    psd = {k: v.detach().cpu() for k, v in prop.state_dict().items()}
    msd = {k: v.detach().cpu() for k, v in mip.state_dict().items()}
    unit = lambda v: v / torch.clamp(v.norm(dim=-1, keepdim=True), min=1e-5)
    # density-gradient normals: on a copy with 4x the weights (the reference's 0.02-sigma initialisation gives 1e-6-sized gradients whose
    # direction is ill-conditioned; the renders below keep the initialisation, whose smooth density keeps the resampling away from bin flips)
    prop4 = ProposalNetwork(10, 320)
    prop4.load_state_dict(psd)
    with torch.no_grad():
        for m_ in prop4.modules():
            if isinstance(m_, torch.nn.Linear):
                m_.weight.mul_(4.0); m_.bias.normal_(0.0, 0.05)
    prop4 = prop4.cuda().eval()
    psd4 = {k: v.detach().cpu() for k, v in prop4.state_dict().items()}
    p_ = dev(x[:, None, :]).requires_grad_(True)
    dens = prop4.forward(p_, contract=True)
    got_n = RefNeRF.get_grad(dens, p_)
    x64 = x[:, None, :].double().requires_grad_(True)
    y = O.proposal_forward({k: v.double() for k, v in psd4.items()}, O.contract(x64))
    g64, = torch.autograd.grad(y.sum(), x64)
    gate("generic proposal density (4x weights, contracted positions) vs fp64", max_abs(dens.detach().cpu().double(), y.detach()) / max(1.0, y.abs().max().item()), 1e-4)
    # (worst sample: with the 4x weights one flipped unit of the last hidden layer carries up to ~ 1/sqrt(320) * 4 of the gradient's direction;
    #  measured 398 of 400 samples within 2e-3, the two others 0.020 and 0.069)
    _normals_close("generic proposal", got_n.cpu().double(), unit(g64), x, g64, worst=0.2)
    d = F.normalize(torch.randn(M, 3, generator=gen), dim=-1)
    with torch.no_grad():
        rgbo = mip.forward(dev(torch.cat((x, d), -1)[:, None, :]), contract=True)
        want = O.mip_forward({k: v.double() for k, v in msd.items()}, torch.cat((O.contract(x.double()), d.double()), -1)[:, None, :])
    assert max_abs(rgbo.cpu().double(), want) <= 2e-5 * max(1.0, want.abs().max().item())
    # the whole render: 40 x 40, unbounded depths, Philox uniforms re-derived for the oracle
    pose = O.pose_spherical(20.0, -30.0, 4.0)
    focal = O.fov2focal(0.6911112070083618, (40, 40))
    near, far, n_f, seed = 0.2, 30.0, 64, 777
    with torch.no_grad():
        res = render_image(mip, prop, pose.cuda(), 40, focal, near, far, n_f, white_bkg=True, render_depth=True, contract=True, seed=seed)
    fx, fy = (float(focal[1]), float(focal[0])) if isinstance(focal, (tuple, list)) else (float(focal), float(focal))
    rays = A.ops.generate_rays(pose[:3].cuda(), 40, 40, fx, fy, torch.device("cuda", 0)).cpu()          # 40 % 40 == 0: ONE tile = raster order
    u1, u2 = O.philox_uniforms(seed, 1600, 0, 64, n_f + 1)
    with torch.no_grad():
        w_rgb, _, w_depth = O.render_rays(psd, msd, rays, u1, u2, near, far, n_f, white_bkg=True, contracted=True)
    gate("generic-route contracted render_image: rgb vs oracle", max_abs(res["rgb"].cpu(), w_rgb.view(40, 40, 3).permute(2, 0, 1)), 1e-4)
    gate("generic-route contracted render_image: depth vs oracle", max_abs(res["depth_img"][0].cpu(), w_depth.view(40, 40)), 1e-4)
    # ... and the integrated PE (mip_methods.py:15-58) on the same route: the stand-alone encoder feeds [frustum mean | feature] to the layers
    near, far = 2.0, 6.0
    radius = 2.0 / (12.0 ** 0.5) / fx
    with torch.no_grad():
        res = render_image(mip, prop, pose.cuda(), 40, focal, near, far, n_f, white_bkg=True, render_depth=True, ipe=True, seed=seed)
        w_rgb, _, w_depth = O.render_rays(psd, msd, rays, u1, u2, near, far, n_f, white_bkg=True, ipe_radius=radius)
    gate("generic-route integrated-PE render_image: rgb vs oracle", max_abs(res["rgb"].cpu(), w_rgb.view(40, 40, 3).permute(2, 0, 1)), 1e-4)
    gate("generic-route integrated-PE render_image: depth vs oracle", max_abs(res["depth_img"][0].cpu(), w_depth.view(40, 40)), 1e-4)
    # ... and BOTH together (round 6: nerf_amd_ipe_feature_contracted; the combination used to be refused on this route): first the encoder
    # itself -- frustum means contracted, covariances metric -- against oracle.ipe_feature(contracted=True) on depths out to z = 30, then
    # the whole unbounded render against oracle.render_rays(contracted=True, ipe_radius=...)
    near, far = 0.2, 30.0
    zt = torch.sort(torch.rand(1600, n_f + 1, generator=gen) * (far - near) + near, dim=-1)[0]
    dn = A.ops.dirs_norm(dev(rays))
    feat, mu, mu_t = A.ops.ipe_feature(dev(zt), dev(rays), 10, radius, dn, contract=True)
    with torch.no_grad():
        w_feat, w_mu, w_mut = O.ipe_feature(zt, rays, 10, radius, dn.cpu(), contracted=True)
    assert float((w_mu.norm(dim=-1) > 1.0).float().mean()) > 0.5                                      # most frusta lie outside the unit ball
    gate("contracted integrated-PE encoder: contracted mean vs oracle", max_abs(mu.cpu(), w_mu), 2e-6)
    # (the highest octave's argument is 512 x the contracted mean: the mean's 3.6e-7 -- one or two ulps, the contraction's norm / divide are not
    #  the oracle's operation order -- may show up as 1.8e-4 in sin / cos; measured 3.9e-5)
    gate("contracted integrated-PE encoder: feature vs oracle", max_abs(feat.cpu(), w_feat), 2e-4)
    assert max_abs(mu_t.cpu(), w_mut) <= 1e-5
    plain_feat, plain_mu, _ = A.ops.ipe_feature(dev(zt), dev(rays), 10, radius, dn)
    assert not torch.equal(plain_mu, mu) and max_abs(A.ops.contract_positions(plain_mu.view(-1, 3)).view_as(mu), mu) <= 2e-6
    with torch.no_grad():
        res = render_image(mip, prop, pose.cuda(), 40, focal, near, far, n_f, white_bkg=True, render_depth=True, ipe=True, contract=True, seed=seed)
        w_rgb, _, w_depth = O.render_rays(psd, msd, rays, u1, u2, near, far, n_f, white_bkg=True, contracted=True, ipe_radius=radius)
    gate("generic-route contracted integrated-PE render_image: rgb vs oracle", max_abs(res["rgb"].cpu(), w_rgb.view(40, 40, 3).permute(2, 0, 1)), 1e-4)
    gate("generic-route contracted integrated-PE render_image: depth vs oracle", max_abs(res["depth_img"][0].cpu(), w_depth.view(40, 40)), 1e-4)


def test_density_gradient_normals_through_the_scene_contraction(A):
    """RefNeRF.get_grad of contracted positions used to raise.  d density / d x with the Mip-NeRF 360 contraction in the
    sample fetch = the encoding's derivative at contract(x) pulled back through the contraction's Jacobian (pe_grad_contract_kernel), for
    the proposal network (train.py:165-168) and Ref-NeRF's spatial network (train.py:178), against fp64 autograd of
    oracle.contract -> oracle forward on positions inside AND outside the unit ball; the parameter backward over the same graph still runs."""
    from nerf_amd.ref_model import RefNeRF
    A.pkg.set_precision("fp32")
    gen = torch.Generator().manual_seed(77)
    M = 300
    x = torch.randn(M, 1, 3, generator=gen) * torch.linspace(0.2, 8.0, M)[:, None, None]          # |x| from 0.1 to ~20
    assert int((x.norm(dim=-1) > 1).sum()) > 50 and int((x.norm(dim=-1) < 1).sum()) > 20
    unit = lambda g: g / torch.clamp(g.norm(dim=-1, keepdim=True), min=1e-5)
    # proposal network
    prop, _ = build_nets(A, "small")
    prop.train()
    p_ = dev(x).requires_grad_(True)
    dens = prop.forward(p_, contract=True)
    got = RefNeRF.get_grad(dens, p_)
    dens.sum().backward()                                                                           # (the dump is still there for the parameters)
    assert float(prop._linear_layers()[0].weight.grad.abs().max()) > 0
    sd = {k: v.double() for k, v in W.proposal_state("small").items()}
    x64 = x.double().requires_grad_(True)
    y = O.proposal_forward(sd, O.contract(x64))
    g64, = torch.autograd.grad(y.sum(), x64)
    assert max_abs(dens.detach().cpu().double(), y.detach()) <= 2e-5 * max(1.0, y.abs().max().item())
    _normals_close("proposal", got.cpu().double(), unit(g64), x, g64)
    # Ref-NeRF
    net = build_ref(A, "small")
    d = F.normalize(torch.randn(M, 1, 3, generator=gen), dim=-1)
    pos = dev(x).requires_grad_(True)
    rgbo, nrm = net.forward(pos, dev(d), contract=True)
    got_r = RefNeRF.get_grad(rgbo[..., -1], pos)
    (rgbo.sum() + nrm.sum()).backward()
    sdr = {k: v.double() for k, v in W.ref_state("small").items()}
    x64 = x.double().requires_grad_(True)
    want, _ = O.ref_forward(sdr, torch.cat((O.contract(x64), d.double()), -1))
    g64, = torch.autograd.grad(want[..., -1].sum(), x64)
    assert max_abs(rgbo.detach().cpu().double(), want.detach()) <= 2e-5 * max(1.0, want.abs().max().item())
    _normals_close("Ref-NeRF", got_r.cpu().double(), unit(g64), x, g64)
    with torch.no_grad():                                                                           # eval path takes the flag too
        e_rgbo, _ = net.forward(dev(x), dev(d), contract=True)
    assert max_abs(e_rgbo.cpu().double(), want.detach()) <= 2e-5 * max(1.0, want.abs().max().item())


# HIP fp32 gradient vs the fp64 value, relative to the tensor's largest entry (set from the measured values the test prints; a ReLU whose
# pre-activation is within rounding of zero can fall on the other side in the kernel, which moves isolated entries by ~1e-3)
G17_FP64_GATE = {k: 3e-4 for k in ("g_rho_tau", "g_rho_tau_bias", "g_spa2_6", "g_nct", "g_spec", "g_prop_head", "g_bottle", "g_dir0", "g_spa0", "g_prop_l0")}
# measured on MI355X (round 3): reference fp32 <= 6.5e-5, HIP fp32 <= 7.8e-5 (g_bottle) of each tensor's largest entry


def test_refnerf_train_step_vs_reference_golden(A, golden):
    """G17: the Ref-NeRF branch of the training step with prop_normal (train.py:164-199) written against the nerf_amd surface --
    train-mode forward with the recorded bottle-neck noise, RefNeRF.get_grad of the density w.r.t. the positions, normal /
    back-face / coarse-normal losses, backward to every parameter -- against the REAL reference's numbers."""
    from nerf_amd.addtional import ProposalLoss, ProposalNetwork, getBounds
    from nerf_amd.mip_methods import maxBlurFilter
    from nerf_amd.nerf_base import NeRF
    from nerf_amd.ref_model import BackFaceLoss, RefNeRF, WeightedNormalLoss
    from nerf_amd.utils import inverseSample
    g = golden("g17_ref_train_step")
    A.pkg.set_precision("fp32")
    prop = ProposalNetwork(10, 256)
    prop.load_state_dict(W.proposal_state("small"))
    net = RefNeRF(10, 4)
    net.load_state_dict(W.ref_state("small"))
    prop, net = prop.cuda().train(), net.cuda().train()
    net.noise_rng = "torch"                              # the recorded perturbation enters through torch.normal (default: in-kernel Philox)
    rays, zc, tgt = dev(g["rays"]), dev(g["z_coarse"]), dev(g["rgb_tgt"])
    noise = dev(g["noise"])
    C17 = zc.shape[-1]
    real_normal = torch.normal
    torch.normal = lambda *a, **k: noise
    try:
        pts = (rays[:, None, :3] + rays[:, None, 3:] * zc[:, :, None]).contiguous().requires_grad_(True)
        dens = prop.forward(pts)
        coarse_grad = -RefNeRF.get_grad(dens, pts)
        dens = F.softplus(dens)
        pw = maxBlurFilter(ProposalNetwork.get_weights(dens, zc, rays[:, 3:]), 0.01)
        fl, below = inverseSample(pw, zc, g["u_inv"].shape[-1], sort=True, u=g["u_inv"])
        samples, fl, below, sort_ids = NeRF.coarseFineMerge(rays, zc, fl, below)
        pos, d = samples.split((3, 3), dim=-1)
        pos = pos.contiguous().requires_grad_(True)
        rgbo, nrm = net.forward(pos, d.contiguous())
        dgrad = -RefNeRF.get_grad(rgbo[..., -1], pos)
        rgbo_raw = rgbo.detach().clone()
        rgbo[..., -1] = F.softplus(rgbo[..., -1] + 0.5)
        rend, wts, _ = NeRF.render(rgbo, fl, rays[:, 3:], net.density_act)        # the reference's positional quirk (train.py:182)
        nl = WeightedNormalLoss()(wts, dgrad, nrm)
        bf = BackFaceLoss()(wts, nrm, d)
        cnl = WeightedNormalLoss()(pw, RefNeRF.coarse_grad_select(dgrad, sort_ids, C17).detach(), coarse_grad)
        img = torch.mean((rend - tgt) ** 2)
        pl = ProposalLoss()(getBounds(pw, below), wts.detach())
        loss = pl + img + 4e-4 * (nl + 0.1 * cnl) + 0.1 * bf
        loss.backward()
    finally:
        torch.normal = real_normal
    # the same train-mode forward through the bf16 kernels (noise path included): close to the reference's fp32 numbers
    A.pkg.set_precision("bf16")
    torch.normal = lambda *a, **k: noise
    try:
        with torch.no_grad():
            rgbo16, nrm16 = net.forward(pos.detach(), d.contiguous())
    finally:
        torch.normal = real_normal
        A.pkg.set_precision("fp32")
    assert max_abs(rgbo16.cpu(), g["rgbo_raw"]) <= 3e-2 * max(1.0, g["rgbo_raw"].abs().max().item()) and max_abs(nrm16.cpu(), g["pred_normal"]) <= 5e-2
    assert torch.equal(sort_ids.cpu(), g["sort_ids"]) and torch.equal(below.cpu(), g["below_merged"])
    assert max_abs(fl.cpu(), g["z_merged"]) <= 2e-5 and max_abs(rgbo_raw.cpu(), g["rgbo_raw"]) <= 2e-5 and max_abs(nrm.detach().cpu(), g["pred_normal"]) <= 2e-5
    assert max_abs(wts.detach().cpu(), g["weights"]) <= 2e-5 and max_abs(rend.detach().cpu(), g["rendered"]) <= 2e-5
    assert max_abs(dgrad.cpu(), g["density_grad"]) <= 2e-3 and max_abs(coarse_grad.cpu(), g["coarse_grad"]) <= 2e-3
    for name, val in (("normal_loss", nl), ("bf_loss", bf), ("coarse_normal_loss", cnl), ("img_loss", img), ("prop_loss", pl), ("loss", loss)):
        assert abs(val.item() - float(g[name])) <= 2e-4 * max(1.0, abs(float(g[name]))), name
    # parameter gradients RELATIVE to each tensor's own size (they range from 1e-8 to 1e-3 with these weights, so an absolute gate
    # would check nothing).  rho_tau_head and the last spa_block2 layer are where the gradient THROUGH THE WEIGHTS lands (normal /
    # back-face losses on the un-detached weights, train.py:183-184): they fail if render's weights output is not differentiable.
    # ... and anchored: the same step in fp64 (oracle.ref_train_step + torch.autograd on the CPU) is the exact value; the REFERENCE's fp32
    # gradients (the golden) sit within ~6e-5 of it on these tensors, and the HIP gradients' distance from it is printed and gated below
    d64 = lambda sd: {k: v.double().requires_grad_(True) for k, v in sd.items()}
    p64, r64 = d64(W.proposal_state("small")), d64(W.ref_state("small"))
    o64 = O.ref_train_step(p64, r64, g["rays"].double(), g["z_coarse"].double(), g["u_inv"].double(), g["noise"].double(), g["rgb_tgt"].double(),
                           g["u_inv"].shape[-1] - 1)
    o64["loss"].backward()
    assert torch.equal(o64["sort_ids"], g["sort_ids"]) and torch.equal(o64["below_merged"], g["below_merged"])
    exact = {"g_rho_tau": r64["rho_tau_head.weight"].grad, "g_rho_tau_bias": r64["rho_tau_head.bias"].grad, "g_spa2_6": r64["spa_block2.6.weight"].grad[:8],
             "g_nct": r64["norm_col_tint_head.weight"].grad, "g_spec": r64["spec_rgb_head.0.weight"].grad, "g_prop_head": p64["layers.8.weight"].grad,
             "g_bottle": r64["bottle_neck.weight"].grad[:8], "g_dir0": r64["dir_block1.0.weight"].grad[:8], "g_spa0": r64["spa_block1.0.weight"].grad[:8],
             "g_prop_l0": p64["layers.0.weight"].grad[:8]}
    have = {"g_rho_tau": net.rho_tau_head.weight.grad, "g_rho_tau_bias": net.rho_tau_head.bias.grad, "g_spa2_6": net.spa_block2[6].weight.grad[:8],
            "g_nct": net.norm_col_tint_head.weight.grad, "g_spec": net.spec_rgb_head[0].weight.grad, "g_prop_head": prop.layers[8].weight.grad,
            "g_bottle": net.bottle_neck.weight.grad[:8], "g_dir0": net.dir_block1[0].weight.grad[:8], "g_spa0": net.spa_block1[0].weight.grad[:8],
            "g_prop_l0": prop.layers[0].weight.grad[:8]}
    rep = {}
    for k in exact:
        top = exact[k].abs().max().item()
        rep[k] = ((g[k].double() - exact[k]).abs().max().item() / top, (have[k].detach().cpu().double() - exact[k]).abs().max().item() / top)
    print("\nG17 gradients, max error relative to the fp64 value (reference fp32 | HIP fp32):", {k: "%.1e | %.1e" % v for k, v in rep.items()})
    for k, (ref_e, hip_e) in rep.items():
        assert hip_e <= max(2.0 * ref_e, G17_FP64_GATE[k]), (k, ref_e, hip_e)
    for key, got, rel in (("g_rho_tau", net.rho_tau_head.weight.grad, 5e-4), ("g_rho_tau_bias", net.rho_tau_head.bias.grad, 5e-4),
                          ("g_spa2_6", net.spa_block2[6].weight.grad[:8], 5e-4), ("g_nct", net.norm_col_tint_head.weight.grad, 5e-4),
                          ("g_spec", net.spec_rgb_head[0].weight.grad, 5e-4), ("g_prop_head", prop.layers[8].weight.grad, 5e-4),
                          ("g_bottle", net.bottle_neck.weight.grad[:8], 5e-4), ("g_dir0", net.dir_block1[0].weight.grad[:8], 5e-4),
                          ("g_spa0", net.spa_block1[0].weight.grad[:8], 5e-4), ("g_prop_l0", prop.layers[0].weight.grad[:8], 5e-4)):
        err, size = max_abs(got.cpu(), g[key]), g[key].abs().max().item()
        assert err <= rel * size, "%s: |err| %.3e vs max|g| %.3e (rel %.2e > %.0e)" % (key, err, size, err / size, rel)


def test_backward_kernels_full_size_linearity(A):
    """BASELINE size (640 000 rays x 128 samples): the backward kernels are linear in the upstream gradient and agree with the
    forward through a directional derivative -- size-independent properties, no oracle needed."""
    N, S = 640000, 128
    gen = torch.Generator(device="cuda").manual_seed(9)
    rgbo = torch.cat((torch.rand(N, S, 3, device="cuda", generator=gen), torch.randn(N, S, 1, device="cuda", generator=gen)), -1)
    z = torch.sort(torch.rand(N, S, device="cuda", generator=gen) * 4 + 2, dim=-1)[0]
    dirs = torch.randn(N, 3, device="cuda", generator=gen)
    g1, g2 = torch.randn(N, 3, device="cuda", generator=gen), torch.randn(N, 3, device="cuda", generator=gen)
    bwd = lambda g: A.ops.composite_backward(rgbo, z, dirs, True, True, A.ops.ACT_RELU, None, g, None, None)
    b1, b2, b12 = bwd(g1), bwd(g2), bwd(2.0 * g1 - 0.5 * g2)
    lin = 2.0 * b1 - 0.5 * b2
    assert (b12 - lin).abs().max().item() <= 1e-4 * max(1.0, lin.abs().max().item())
    # <g, J v> = <J^T g, v> with a finite-difference J v of the forward (colour channels only: the map is linear in them)
    v = torch.zeros_like(rgbo)
    v[..., :3] = torch.randn(N, S, 3, device="cuda", generator=gen)
    f0 = A.ops.composite(rgbo, z, dirs, True, True, A.ops.ACT_RELU, None, want_weights=False)[0]
    f1 = A.ops.composite(rgbo + v, z, dirs, True, True, A.ops.ACT_RELU, None, want_weights=False)[0]
    lhs = ((f1 - f0) * g1).sum(dtype=torch.float64).item()
    rhs = (b1 * v).sum(dtype=torch.float64).item()
    assert abs(lhs - rhs) <= 2e-4 * max(1.0, abs(rhs))


def test_scene_contraction_flag(A):
    """`contract` in the samples descriptor (Mip-NeRF 360 eq. 10; BASELINE config 5, parity unpinned -- own definition in
    oracle.contract): MLP kernels on contracted positions and the whole render path, against the oracle; off by default."""
    prop, mip = build_nets(A, "small")
    A.pkg.set_precision("fp32")
    gen = torch.Generator().manual_seed(77)
    pts = torch.cat((torch.randn(300, 3, generator=gen) * torch.logspace(-1, 2, 300)[:, None], torch.randn(300, 3, generator=gen)), -1)
    with torch.no_grad():
        got = A.ops.mip_forward(mip.packed(A.ops.F32), A.ops.F32, pts.cuda(), contract=True).cpu()
        want = O.mip_forward(W.mip_state("small"), torch.cat((O.contract(pts[:, :3]), pts[:, 3:]), -1))
        assert max_abs(got, want) <= 2e-5
        assert max_abs(A.ops.proposal_forward(prop.packed(A.ops.F32), A.ops.F32, pts[:, :3].contiguous().cuda(), contract=True).cpu(),
                       O.proposal_forward(W.proposal_state("small"), O.contract(pts[:, :3]))) <= 2e-5
        assert not torch.equal(got, A.ops.mip_forward(mip.packed(A.ops.F32), A.ops.F32, pts.cuda()).cpu())
        # render path: unbounded-style rays (far = 40) through both networks
        N = 96
        o = torch.tensor([0.0, 0.0, 0.5]).expand(N, 3)
        d = F.normalize(torch.randn(N, 3, generator=gen), dim=-1)
        rays = torch.cat((o, d), -1).contiguous()
        u1, u2 = torch.rand(N, 64, generator=gen), torch.rand(N, 129, generator=gen)
        near, far = 0.1, 40.0
        rgb, depth, w, _ = A.ops.render_rays(prop.packed(A.ops.F32), mip.packed(A.ops.F32), A.ops.F32, rays.cuda(), torch.linspace(near, far, 64).cuda(),
                                             u1.cuda(), u2.cuda(), 128, near, far, True, want_depth=True, want_weights=True, contract=True)
        r_rgb, r_w, r_depth = O.render_rays(W.proposal_state("small"), W.mip_state("small"), rays, u1, u2, near, far, 128, white_bkg=True, contracted=True)
        assert max_abs(rgb.cpu(), r_rgb) <= 1e-4 and max_abs(w.cpu(), r_w) <= 1e-4
        gate("contracted render, depths 0.1 .. 40: depth vs oracle", max_abs(depth.cpu(), r_depth), 1e-4)          # (measured 4.8e-7; was 1e-3)


def test_render_image_config5_shape_untiled_contracted(A):
    """BASELINE config 5's shape class: a non-square image whose width no reference tile size divides (the reference raises there;
    this build renders un-tiled), unbounded near/far with scene contraction, tuple focal -- render_image vs the oracle on the same
    CPU-generator uniforms."""
    prop, mip = build_nets(A, "small")
    A.pkg.set_precision("fp32")
    H, Wd, near, far, n_f = 41, 67, 0.2, 30.0, 64
    pose = O.pose_spherical(25.0, -20.0, 1.5)[:3]
    focal = (55.0, 48.0)                                                            # (fy, fx) tuple form (procedures.py:45-47)
    torch.manual_seed(2024)
    with torch.no_grad():
        res = A.procedures.render_image(mip.eval(), prop.eval(), pose.cuda(), (H, Wd), focal, near, far, n_f, white_bkg=True, render_depth=True,
                                        contract=True, rng="reference")
    torch.manual_seed(2024)
    u1, u2 = torch.rand((H * Wd, 64)), torch.rand((H * Wd, n_f + 1))
    dirs = O.ray_dirs_image(pose, H, Wd, focal).reshape(-1, 3)
    rays = torch.cat((pose[:, -1].expand(H * Wd, -1), dirs), -1)
    rgb, _, depth = O.render_rays(W.proposal_state("small"), W.mip_state("small"), rays, u1, u2, near, far, n_f, white_bkg=True, contracted=True)
    assert res["rgb"].shape == (3, H, Wd) and res["depth_img"].shape == (3, H, Wd)
    assert max_abs(res["rgb"].cpu(), rgb.view(H, Wd, 3).permute(2, 0, 1)) <= 1e-4
    gate("config5-shape untiled contracted render_image: depth vs oracle", max_abs(res["depth_img"][0].cpu(), depth.view(H, Wd)), 1e-4)   # (measured 7.2e-7)


@pytest.mark.parametrize("K,C", [(129, 64), (65, 64), (7, 3), (1, 1), (300, 200)])
def test_merge_depths_equals_sort_of_concatenation(K, C):
    """Render-path sort of coarseFineMerge (nerf_base.py:59-73, procedures.py:72) as a merge: bit-identical values, ties included."""
    from nerf_amd import ops
    g = torch.Generator().manual_seed(K * 1000 + C)
    N = 777
    a = torch.sort(torch.rand(N, K, generator=g) * 4 + 2, dim=-1)[0]
    b = torch.sort(torch.rand(N, C, generator=g) * 4 + 2, dim=-1)[0]
    if K > 2 and C > 2:                                   # exact ties within and across the two sets
        b[:, 1] = a[:, 2]
        a[::2, 1] = a[::2, 2]
        b[::3, C - 1] = b[::3, C - 2]
    a, b = torch.sort(a, dim=-1)[0].cuda(), torch.sort(b, dim=-1)[0].cuda()
    want = torch.sort(torch.cat((a, b), dim=-1), dim=-1)[0][:, :-1]
    got = ops.merge_depths(a, b)
    assert got.shape == want.shape and torch.equal(got, want)
    assert ops.merge_depths(a[:0], b[:0]).shape == (0, K + C - 1)
    # stratified depths whose jitter exceeds the bin spacing (procedures.py:59 with fewer than 63 fine samples) are NOT ascending:
    # rays with an out-of-order input take the sorting path and still equal the sort
    b2 = b.clone()
    if C > 3:
        b2[::4, 1], b2[::4, 2] = b[::4, 2].clone(), b[::4, 1].clone()
        b2[1::7] = b[1::7].flip(-1)
    a2 = a.clone()
    if K > 3:
        a2[2::5, 0], a2[2::5, K - 1] = a[2::5, K - 1].clone(), a[2::5, 0].clone()
    want2 = torch.sort(torch.cat((a2, b2), dim=-1), dim=-1)[0][:, :-1]
    assert torch.equal(ops.merge_depths(a2, b2), want2)


@pytest.mark.parametrize("K,C", [(129, 64), (65, 64), (7, 3), (1, 1), (300, 200)])
def test_coarse_fine_merge_with_order_equals_the_stable_sort(K, C):
    """coarseFineMerge as the training loop calls it (nerf_base.py:59-73 with f_inds, train.py:176): sorted depths, the sort order and
    the gathered bin indices from ONE merge kernel equal torch.sort(stable=True) + arange / cat / gather exactly -- ties and
    out-of-order inputs included (integer work: torch.equal)."""
    from nerf_amd import ops
    from nerf_amd.nerf_base import NeRF
    g = torch.Generator().manual_seed(K * 1000 + C + 1)
    N = 515
    a = torch.sort(torch.rand(N, K, generator=g) * 4 + 2, dim=-1)[0]
    b = torch.sort(torch.rand(N, C, generator=g) * 4 + 2, dim=-1)[0]
    if K > 2 and C > 2:
        b[:, 1] = a[:, 2]
        a[::2, 1] = a[::2, 2]
        b[::3, C - 1] = b[::3, C - 2]
        a, b = torch.sort(a, dim=-1)[0], torch.sort(b, dim=-1)[0]
    if C > 3:                                              # some rays out of order (the rank-sort path keeps the indices with the values)
        b[::4, 1], b[::4, 2] = b[::4, 2].clone(), b[::4, 1].clone()
        b[1::7] = b[1::7].flip(-1)
    if K > 3:
        a[2::5, 0], a[2::5, K - 1] = a[2::5, K - 1].clone(), a[2::5, 0].clone()
    f_inds = torch.randint(0, C, (N, K), generator=g)
    a, b, f_inds = a.cuda(), b.cuda(), f_inds.cuda()
    wz, wo = torch.sort(torch.cat((a, b), dim=-1), dim=-1, stable=True)
    wi = torch.gather(torch.cat((f_inds, torch.arange(C, device="cuda").expand(N, -1)), dim=-1), -1, wo)
    z, order, all_inds = ops.merge_depths_order(a, b, f_inds)
    assert torch.equal(z, wz[:, :-1]) and torch.equal(order, wo) and torch.equal(all_inds, wi)
    z2, order2, none = ops.merge_depths_order(a, b)
    assert none is None and torch.equal(z2, z) and torch.equal(order2, order)
    rays = torch.cat((torch.zeros(N, 3), torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)), -1).cuda()
    samples, zz, ai, so = NeRF.coarseFineMerge(rays, b, a, f_inds)
    assert torch.equal(zz, wz[:, :-1]) and torch.equal(ai, wi) and torch.equal(so, wo[:, :-1])
    assert torch.equal(samples, ops.length2pts(rays, wz[:, :-1].contiguous()))
    assert len(NeRF.coarseFineMerge(rays, b, a)) == 2


def test_coarse_grad_select_kernel_equals_the_references_mask():
    """RefNeRF.coarse_grad_select (ref_model.py:108-117) on the device, against the oracle's boolean-mask form on real merge orders, and
    the rows a boolean mask cannot express (fewer flagged positions than c_pnum: the stable-sort definition)."""
    from nerf_amd import ops
    from nerf_amd.ref_model import RefNeRF
    g = torch.Generator().manual_seed(12)
    N, K, C = 301, 129, 64
    a = torch.sort(torch.rand(N, K, generator=g) * 4 + 2, dim=-1)[0]
    b = torch.sort(torch.rand(N, C, generator=g) * 4 + 2, dim=-1)[0]
    b[:, -1] = 6.5                                          # the last coarse depth is the largest, as in the training loop: it is the one dropped
    T = K + C - 1
    order = torch.sort(torch.cat((a, b), dim=-1), dim=-1, stable=True)[1][:, :-1]
    grads = torch.randn(N, T, 3, generator=g)
    want = O.coarse_grad_select(grads, order, C)
    got = RefNeRF.coarse_grad_select(grads.cuda(), order.cuda(), C)
    assert got.shape == (N, C, 3) and torch.equal(got.cpu(), want)
    # fewer flagged positions than asked for: the flagged ones first, then the rest in order
    si = torch.arange(T).expand(4, -1).clone()
    si[1] = si[1].flip(-1)
    si[2, :] = 0
    si[3, ::2] = T
    sel = (si >= T - C).to(torch.int8)
    pos = torch.sort(sel, dim=-1, descending=True, stable=True)[1][:, :C]
    g4 = torch.randn(4, T, 5, generator=g)
    want4 = torch.gather(g4, 1, pos[:, :, None].expand(-1, -1, 5))
    assert torch.equal(ops.coarse_grad_select(g4.cuda(), si.cuda(), C).cpu(), want4)
    assert ops.coarse_grad_select(g4.cuda()[:0], si.cuda()[:0], C).shape == (0, C, 5)


def test_mfma_stream_measurement_aid_runs():
    """nerf_amd_mfma_stream (bench.py roofline.mfma_stream_ref): every operand mode launches, completes, and sustains a rate in the range
    of a matrix-core stream (sanity bounds only: the number is a measurement, not a contract); operands that toggle must not come out
    FASTER than constant ones by more than the box's noise (the constant-operand stream is the optimistic ceiling)."""
    from nerf_amd import ops
    dev = torch.device("cuda")
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    rates = {}
    for mode in (0, 1, 2, 3):
        ops.mfma_stream(200, n_cu, dev, mode)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.mfma_stream(5000, n_cu, dev, mode)
        e.record()
        torch.cuda.synchronize()
        rates[mode] = n_cu * 4 * 5000 * 64 * 32768.0 / (s.elapsed_time(e) * 1e-3) / 1e12
    print("\nMFMA-only streams on %d CUs, TFLOP/s: constant %.0f | random %.0f | weights x post-ReLU %.0f | + A through LDS %.0f"
          % (n_cu, rates[0], rates[1], rates[2], rates[3]))
    for tf in rates.values():
        assert 500.0 < tf < 2600.0
    assert rates[1] <= rates[0] * 1.05
    with pytest.raises(RuntimeError):
        ops.mfma_stream(10, 0, dev)
    with pytest.raises(RuntimeError):
        ops.mfma_stream(10, n_cu, dev, 7)


def test_refnerf_normal_losses_kernels_equal_the_torch_expressions():
    """WeightedNormalLoss / BackFaceLoss (ref_model.py:127-143) on the device (nerf_amd_weighted_dot_loss[_backward]): value and the three
    gradients against the reference's torch expressions evaluated in fp64 (the kernels sum in double, fixed order)."""
    from nerf_amd.ref_model import BackFaceLoss, WeightedNormalLoss
    g = torch.Generator().manual_seed(3)
    for shape in ((1, 1), (37, 5), (512, 192)):
        w = torch.rand(*shape, generator=g)
        a = torch.nn.functional.normalize(torch.randn(*shape, 3, generator=g), dim=-1)
        both = torch.randn(*shape, 6, generator=g)                     # (b is a strided view, like fine_dir in train.py:177)
        for mod, expr in ((WeightedNormalLoss(), lambda w_, a_, b_: torch.sum(w_ * (1.0 - torch.sum(a_ * b_, dim=-1)))),
                          (WeightedNormalLoss(size_average=True), lambda w_, a_, b_: torch.mean(w_ * (1.0 - torch.sum(a_ * b_, dim=-1)))),
                          (BackFaceLoss(), lambda w_, a_, b_: torch.mean(w_ * F.relu(torch.sum(a_ * b_, dim=-1))))):
            W64, A64, B64 = (t.double().requires_grad_(True) for t in (w, a, both[..., 3:]))
            want = expr(W64, A64, B64)
            (want * 1.7).backward()
            Wd, Ad = w.cuda().requires_grad_(True), a.cuda().requires_grad_(True)
            Bd = both.cuda().requires_grad_(True)
            got = mod(Wd, Ad, Bd[..., 3:])
            (got * 1.7).backward()
            assert got.shape == () and abs(got.item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
            for have, ex in ((Wd.grad, W64.grad), (Ad.grad, A64.grad), (Bd.grad[..., 3:], B64.grad)):
                assert max_abs(have.cpu().double(), ex) <= 2e-6 * max(1e-6, ex.abs().max().item())
            assert float(Bd.grad[..., :3].abs().max()) == 0.0
            with torch.no_grad():                                       # no graph: the plain kernel
                assert abs(mod(Wd, Ad, Bd[..., 3:]).item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
            # only the inputs that ask get a gradient (train.py:187: the coarse normals carry none)
            Wd2 = w.cuda().requires_grad_(True)
            mod(Wd2, a.cuda(), both.cuda()[..., 3:]).backward()
            assert max_abs(Wd2.grad.cpu().double() * 1.7, W64.grad) <= 4e-6 * max(1e-6, W64.grad.abs().max().item())


@pytest.mark.parametrize("L,normalize", [(10, False), (4, True)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_encode_rows_matches_positional_encoding(L, normalize, prec):
    """nerf_amd_encode_rows = [x | positional_encoding (nerf_helper.py:38-48) | 0 pad]: the wgrad operand of the first / skip layers."""
    from nerf_amd import ops
    g = torch.Generator().manual_seed(5 + L)
    pts = (torch.rand(3001, 6, generator=g) * 8 - 4).cuda()
    x = pts[:, 3:6] if normalize else pts[:, :3]                    # strided column slices of the (M,6) sample matrix
    precision = ops.BF16 if prec == "bf16" else ops.F32
    got = ops.encode_rows(x, L, precision, normalize)
    xr = (x / x.norm(dim=-1, keepdim=True) if normalize else x).cpu()
    pe = O.positional_encoding(xr, L)
    ncol = (3 + 6 * L + 7) // 8 * 8
    want = torch.zeros(x.shape[0], ncol)
    want[:, :3] = xr
    want[:, 3:3 + 6 * L] = pe
    assert got.shape == want.shape and got.dtype == (torch.bfloat16 if prec == "bf16" else torch.float32)
    if prec == "fp32":
        assert max_abs(got.cpu(), want) <= (2e-6 * 2 ** L if normalize else 2e-6)   # a 1-ulp difference of the normalised direction scales with the octave
    else:
        assert max_abs(got.float().cpu(), want.bfloat16().float()) <= 2 ** -7   # one bf16 ulp at |v| <= 4
    assert ops.encode_rows(x[:0], L, precision, normalize).shape == (0, ncol)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_train_dump_rows_mask_equals_the_two_separate_passes(A, prec):
    """nerf_amd_train_dump_rows_mask == nerf_amd_train_dump_to_rows followed by nerf_amd_relu_mask_bias (rows and masked delta bit
    for bit, column sums to summation order), for 256- and 128-wide layers and a sample count that is no multiple of a tile."""
    prop, mip = build_nets(A, "he")
    A.pkg.set_precision(prec)
    P = A.ops.current_precision()
    dt = torch.bfloat16 if prec == "bf16" else torch.float32
    gen = torch.Generator().manual_seed(77)
    M = 4099
    pts6 = torch.cat((torch.rand(M, 3, generator=gen) * 2 - 1, torch.randn(M, 3, generator=gen)), -1).cuda().contiguous()
    with torch.no_grad():
        _, dump = A.ops.mip_forward_train(mip.packed(P), P, pts6)
        for layer, width in ((2, 256), (6, 256), (7, 128)):
            delta = torch.randn(M, width, generator=gen).cuda().to(dt)
            rows = A.ops.train_dump_rows(dump, A.ops.NET_MIP, P, M, layer, width)
            want_d, want_s = A.ops.relu_mask_bias_(delta.clone(), rows, P)
            act, got_d, got_s = A.ops.train_dump_rows_mask_(dump, A.ops.NET_MIP, P, layer, delta.clone())
            assert torch.equal(act, rows) and torch.equal(got_d, want_d), (layer, width)
            assert 0.05 < (rows > 0).float().mean().item() < 0.95                 # the mask is not trivial
            assert max_abs(got_s.cpu(), want_s.cpu()) <= 1e-3 * max(1.0, want_s.abs().max().item())
            assert max_abs(got_s.cpu(), want_d.float().sum(0).cpu()) <= 1e-3 * max(1.0, want_s.abs().max().item())
    A.pkg.set_precision("fp32")


@pytest.mark.parametrize("S", [1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300])
def test_composite_sample_count_sweep(A, S):
    """Compositing (nerf_base.py:75-113) across the kernel's paths: register fast path with 2 chunks (S <= 128) and 4 chunks (S <= 256),
    generic path beyond -- rgb, weights and depth against the oracle, with and without the |d| scaling / white background."""
    gen = torch.Generator().manual_seed(100 + S)
    N = 333
    rgbo = torch.randn(N, S, 4, generator=gen)
    z = torch.sort(torch.rand(N, S, generator=gen) * 4 + 2, dim=-1)[0]
    d = torch.randn(N, 3, generator=gen)
    for mn, wb in ((True, True), (False, False)):
        rgb, w, ex = A.nerf_base.NeRF.render(dev(rgbo), dev(z), dev(d), mul_norm=mn, white_bkg=wb, render_depth=(NEAR, FAR))
        want_rgb, want_w, want_ex = O.composite(rgbo, z, d, mul_norm=mn, white_bkg=wb, render_depth=(NEAR, FAR))
        # (the oracle's transmittance is an fp32 cumprod of S terms, the kernel's a double-precision scan: the gap grows with S)
        assert max_abs(rgb.cpu(), want_rgb) <= 1e-5 and max_abs(w.cpu(), want_w) <= 3e-6, (S, mn, wb)
        assert max_abs(ex["depth_img"].cpu(), want_ex["depth_img"]) <= 1e-5, (S, mn, wb)


def test_density_gradient_kernels_in_bf16_keep_the_fp32_direction(A):
    """RefNeRF.get_grad on the proposal density (train.py:165-168) runs the dgrad-only chain + encoding derivative
    (nerf_amd_density_grad) and the parameter backward runs the fused chain + MFMA weight gradients, in fp32 and in bf16 arithmetic:
    the bf16 density-gradient normals and parameter gradients keep the direction of the fp32 ones, get_grad leaves `.grad` alone, and
    no torch.autograd VJP is involved any more (the positions' gradient comes from the kernels)."""
    from nerf_amd import autograd_bridge as ab
    from nerf_amd.ref_model import RefNeRF
    prop, _ = build_nets(A, "small")
    prop.train()
    gen = torch.Generator().manual_seed(43)
    pts0 = (torch.rand(64, 64, 3, generator=gen) * 2 - 1).cuda()
    layers = prop._linear_layers()
    params = [l.weight for l in layers] + [l.bias for l in layers]
    res = {}
    for prec in ("fp32", "bf16"):
        A.pkg.set_precision(prec)
        for p_ in params:
            p_.grad = None
        pts = pts0.clone().requires_grad_(True)
        dens = prop.forward(pts)
        normals = RefNeRF.get_grad(dens, pts)
        assert all(p_.grad is None for p_ in params)                 # autograd.grad leaves .grad alone ...
        F.softplus(dens).sum().backward()
        res[prec] = (normals.detach(), [p_.grad.clone() for p_ in params])
    A.pkg.set_precision("fp32")
    cos_n = F.cosine_similarity(res["bf16"][0].reshape(-1, 3), res["fp32"][0].reshape(-1, 3), dim=-1)
    assert cos_n.median().item() >= 0.999 and (cos_n >= 0.9).float().mean().item() >= 0.97, (cos_n.median().item(), (cos_n >= 0.9).float().mean().item())
    for k, (a_, b_) in enumerate(zip(res["bf16"][1], res["fp32"][1])):
        cos = F.cosine_similarity(a_.reshape(1, -1), b_.reshape(1, -1)).item()
        assert cos >= 0.97, (k, cos)
    # the position gradient does not come from a torch re-evaluation: no autograd.grad call happens inside the op's backward
    seen = []
    real = torch.autograd.grad
    def spy(y, leaves, *a, **k):
        seen.append(len(leaves))
        return real(y, leaves, *a, **k)
    pts = pts0.clone().requires_grad_(True)
    dens = prop.forward(pts)
    with ab.inputs_only_grad():
        torch.autograd.grad = spy
        try:
            g, = real(dens, pts, torch.ones_like(dens), retain_graph=True)
        finally:
            torch.autograd.grad = real
    assert seen == [] and g.shape == pts.shape and bool(torch.isfinite(g).all())


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_render_rays_ref_equals_the_separate_entry_points(A, prec):
    """nerf_amd_render_rays_ref (the Ref-NeRF tile body of procedures.py:64-85 in one C-ABI call) == proposal / resample / merge /
    Ref-NeRF MLP / compositing called one by one -- bit for bit, with explicit rays and with rays generated from the camera descriptor
    (a sub-range of the image), with and without the normal image; plus its argument checks."""
    from nerf_amd._lib import Samples
    from nerf_amd.ref_model import RefNeRF
    ops = A.ops
    prop, _ = build_nets(A, "small")
    net = RefNeRF(10, 4)
    net.load_state_dict(W.ref_state("small"))
    net = net.cuda().eval()
    A.pkg.set_precision(prec)
    P = ops.current_precision()
    H, Wd, fx, n_fine = 24, 20, 30.0, 48
    gen = torch.Generator().manual_seed(3)
    pose = torch.eye(4)[:3].clone()
    pose[:, 3] = torch.tensor([0.1, -0.2, 4.0])
    pose = pose.cuda()
    rays_all = ops.generate_rays(pose, H, Wd, fx, fx, pose.device)
    lo, N = 37, 300                                                   # a sub-range of the raster, no multiple of anything
    rays = rays_all[lo:lo + N].contiguous()
    u1, u2 = torch.rand(N, 64, generator=gen).cuda(), torch.rand(N, n_fine + 1, generator=gen).cuda()
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    cam_dir = pose[:, 2].contiguous()
    with torch.no_grad():
        jitter = (FAR - NEAR) / n_fine
        sc = ops.samples_rays(rays, 64, z_base=z_base, u=u1, z_jitter=jitter)
        dens = ops.proposal_forward_samples(prop.packed(P), P, sc, (N, 64), rays.device)
        z_fine, _, _, z_c = ops.resample(dens, None, z_base, u1, jitter, rays, u2, n_fine + 1, want_zc=True)
        z_all = ops.merge_depths(z_fine, z_c)
        rgbo, normal = ops.ref_forward_samples(net.packed(P), P, ops.samples_rays(rays, n_fine + 64, z=z_all), (N, n_fine + 64), rays.device)
        want_rgb, _, want_depth, want_nimg = ops.composite(rgbo, z_all, rays, True, True, ops.ACT_SOFTPLUS, (NEAR, FAR), normal, cam_dir,
                                                           want_weights=False, sigma_shift=0.5)
        rgb, depth, nimg, ws = ops.render_rays_ref(prop.packed(P), net.packed(P), P, rays, z_base, u1, u2, n_fine, NEAR, FAR, True, cam_dir=cam_dir)
        assert torch.equal(rgb, want_rgb) and torch.equal(depth, want_depth) and torch.equal(nimg, want_nimg), (max_abs(rgb, want_rgb), max_abs(depth, want_depth), max_abs(nimg, want_nimg))
        rgb2, depth2, nimg2, _ = ops.render_rays_ref(prop.packed(P), net.packed(P), P, rays, z_base, u1, u2, n_fine, NEAR, FAR, True, workspace=ws)
        assert torch.equal(rgb2, want_rgb) and torch.equal(depth2, want_depth) and nimg2 is None
        cam = Samples()
        cam.H, cam.W, cam.fx, cam.fy = H, Wd, fx, fx
        for i, v in enumerate(pose.cpu().reshape(-1).tolist()):
            cam.pose[i] = v
        rgb3, depth3, nimg3, _ = ops.render_rays_ref(prop.packed(P), net.packed(P), P, None, z_base, u1, u2, n_fine, NEAR, FAR, True, cam_dir=cam_dir,
                                                     camera=cam, ray_offset=lo, n_rays=N)
        assert torch.equal(rgb3, want_rgb) and torch.equal(depth3, want_depth) and torch.equal(nimg3, want_nimg)
        with pytest.raises(RuntimeError, match="ray range"):
            ops.render_rays_ref(prop.packed(P), net.packed(P), P, None, z_base, u1, u2, n_fine, NEAR, FAR, True, camera=cam, ray_offset=H * Wd - 10, n_rays=N)
        cam.contract = 1                                        # (round 4: accepted -- test_refnerf_render_with_scene_contraction checks the values)
        rgb_c, _, _, _ = ops.render_rays_ref(prop.packed(P), net.packed(P), P, None, z_base, u1, u2, n_fine, NEAR, FAR, True, camera=cam, ray_offset=lo, n_rays=N)
        assert bool(rgb_c.isfinite().all())
    assert (rgb.isfinite().all() and 0.0 < rgb.std().item())
    A.pkg.set_precision("fp32")


@pytest.mark.parametrize("n_fine", [40, 64, 128])
def test_render_rays_ref_fp32_parity_vs_oracle(A, n_fine):
    """nerf_amd_render_rays_ref vs the CPU restatement of the reference's Ref-NeRF tile body (procedures.py:64-85) on identical rays and
    uniforms: rgb / depth / normal image <= 1e-4.  n_fine = 40: the stratified depths are out of order and the reference's sort of
    the merged depths really sorts."""
    from nerf_amd.ref_model import RefNeRF
    prop, _ = build_nets(A, "small")
    net = RefNeRF(10, 4)
    net.load_state_dict(W.ref_state("small"))
    net = net.cuda().eval()
    A.pkg.set_precision("fp32")
    n = 160
    rays, u1, u2 = _rays_and_u(n, n_fine, 23)
    cam_z = torch.tensor([0.2, -0.3, 0.9])
    with torch.no_grad():
        want_rgb, _, ex = O.render_rays_ref(W.proposal_state("small"), W.ref_state("small"), rays, u1, u2, NEAR, FAR, n_fine, white_bkg=True, cam_z=cam_z)
        rgb, depth, nimg, _ = A.ops.render_rays_ref(prop.packed(A.ops.F32), net.packed(A.ops.F32), A.ops.F32, dev(rays), torch.linspace(NEAR, FAR, 64).cuda(),
                                                    dev(u1), dev(u2), n_fine, NEAR, FAR, True, cam_dir=dev(cam_z))
    assert max_abs(rgb.cpu(), want_rgb) <= 1e-4
    assert max_abs(depth.cpu(), ex["depth_img"]) <= 1e-4
    assert max_abs(nimg.cpu(), ex["normal_img"]) <= 1e-4


# ------------------------------------------------------------------------------------------------ row 12: integrated PE
@pytest.mark.parametrize("name,L", [("g12_ipe", 6), ("g18_ipe_l10", 10)])
def test_ipe_feature_vs_reference_golden(A, golden, name, L):
    """nerf_amd_ipe_feature / nerf_amd_cone_parameters / nerf_amd_dirs_norm against the real reference's ipe_feature
    (mip_methods.py:15-58), <= 1e-6 (features live in [-1, 1]; means and moments relative to their size)."""
    g = golden(name)
    r = 0.0015 if name == "g12_ipe" else g["radius"]
    feat, mu, mu_t = A.mip_methods.ipe_feature(dev(g["z"]), dev(g["rays"]), L, r)
    assert feat.shape == g["feat"].shape and mu.shape == g["mu"].shape and mu_t.shape == g["mu_t"].shape
    assert max_abs(feat.cpu(), g["feat"]) <= 1e-6
    assert max_abs(mu.cpu(), g["mu"]) <= 1e-6 * max(1.0, g["mu"].abs().max().item())
    assert max_abs(mu_t.cpu(), g["mu_t"]) <= 1e-6 * max(1.0, g["mu_t"].abs().max().item())
    if name == "g18_ipe_l10":
        cp = A.mip_methods.coneParameters(dev(g["z"]), r)
        assert torch.equal(cp[0].cpu(), mu_t.cpu())
        assert max_abs(cp[1].cpu(), g["var_t"]) <= 1e-6 * g["var_t"].abs().max().item()
        assert max_abs(cp[2].cpu(), g["var_r"]) <= 1e-6 * g["var_r"].abs().max().item()
        assert abs(A.ops.dirs_norm(dev(g["rays"])).item() - g["dir_norm"]) <= 2e-7 * g["dir_norm"]


def test_ipe_feature_ragged_and_empty(A):
    gen = torch.Generator().manual_seed(4)
    for N, S in ((1, 1), (3, 7), (257, 5), (2, 300)):
        z = torch.sort(NEAR + (FAR - NEAR) * torch.rand(N, S + 1, generator=gen), -1)[0]
        rays = torch.cat((torch.randn(N, 3, generator=gen), torch.randn(N, 3, generator=gen)), -1)
        want = O.ipe_feature(z, rays, 10, 7e-4)
        got = A.mip_methods.ipe_feature(dev(z), dev(rays), 10, 7e-4)
        for a, b in zip(got, want):
            assert a.shape == b.shape and max_abs(a.cpu(), b) <= 1e-6 * max(1.0, b.abs().max().item())
    f, m, t = A.mip_methods.ipe_feature(torch.zeros(0, 9).cuda(), torch.zeros(0, 6).cuda(), 10, 1e-3)
    assert f.shape == (0, 8, 60) and m.shape == (0, 8, 3) and t.shape == (0, 8)


@pytest.mark.parametrize("tag,n,n_fine", [("small", 300, 128), ("he", 200, 128), ("small", 130, 64)])
def test_render_rays_ipe_fp32_parity(A, tag, n, n_fine):
    """BASELINE config 3 wiring (fine network on [mu | ipe_feature] of the frusta between consecutive fine depths; the build's own
    definition of the loop, O.render_rays(ipe_radius=...)): the fused in-register IPE of mip_kernel<.., IPE> against the oracle on
    identical rays and uniforms, <= 1e-4 abs on RGB / depth / weights."""
    prop, mip = build_nets(A, tag)
    rays, u1, u2 = _rays_and_u(n, n_fine, 23)
    radius = 2.0 / math.sqrt(12.0) / 1111.0
    with torch.no_grad():
        want_rgb, want_w, want_depth = O.render_rays(W.proposal_state(tag), W.mip_state(tag), rays, u1, u2, NEAR, FAR, n_fine,
                                                     white_bkg=True, ipe_radius=radius)
        plain_rgb, _, _ = O.render_rays(W.proposal_state(tag), W.mip_state(tag), rays, u1, u2, NEAR, FAR, n_fine, white_bkg=True)
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    rgb, depth, w, _ = A.ops.render_rays(prop.packed(A.ops.F32), mip.packed(A.ops.F32), A.ops.F32, dev(rays), z_base, dev(u1), dev(u2),
                                         n_fine, NEAR, FAR, True, want_depth=True, want_weights=True, ipe_radius=radius)
    tol = 1e-4 if tag == "small" else 2e-4                   # 'he' + IPE: the oracle itself is ~1e-4 from the exact value here
    assert max_abs(rgb.cpu(), want_rgb) <= tol and max_abs(depth.cpu(), want_depth) <= tol and max_abs(w.cpu(), want_w) <= tol
    if tag == "he":                                           # the encoding really changed the image (the 'small' networks are almost
        assert max_abs(want_rgb, plain_rgb) > 10 * tol        # insensitive to their input, so only the O(1)-activation set can show it)
    # the standalone entry points give the same fine-network input: mip_forward on the materialised [mu | ipe] is not available
    # (the network builds its own encoding), so compare the kernel's fused encoding through a bf16 run of the same call instead
    rgb16, _, _, _ = A.ops.render_rays(prop.packed(A.ops.BF16), mip.packed(A.ops.BF16), A.ops.BF16, dev(rays), z_base, dev(u1), dev(u2),
                                       n_fine, NEAR, FAR, True, ipe_radius=radius)
    assert torch.mean((rgb16.cpu() - want_rgb) ** 2).item() <= (1e-4 if tag == "small" else 2e-3)


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_full_size_properties_ipe(A, prec):
    """BASELINE config 3 at its size (Mip-NeRF with integrated PE, 800x800, 64+128; fp32 on a 200k-ray slab): scale-free properties
    + an oracle spot check that uses the SAME whole-batch direction norm (mip_methods.py:31) as the full launch."""
    prop, mip = build_nets(A, "he")
    P = A.ops.F32 if prec == "fp32" else A.ops.BF16
    H = Wd = 800
    pose = O.pose_spherical(20.0, -30.0, 4.0)[:3]
    f = O.fov2focal(0.6911112070083618, (H, Wd))
    radius = 2.0 / math.sqrt(12.0) / float(f[1])
    n = H * Wd if prec == "bf16" else 200_000
    rays = A.ops.generate_rays(pose, H, Wd, f[1], f[0], "cuda", 0, n)
    gen = torch.Generator(device="cuda").manual_seed(11)
    u1 = torch.rand(n, 64, device="cuda", generator=gen)
    u2 = torch.rand(n, 129, device="cuda", generator=gen)
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    pk_p, pk_m = prop.packed(P), mip.packed(P)
    rgb_w, depth, w, ws = A.ops.render_rays(pk_p, pk_m, P, rays, z_base, u1, u2, 128, NEAR, FAR, True, want_depth=True, want_weights=True,
                                            ipe_radius=radius)
    rgb_b, _, _, ws = A.ops.render_rays(pk_p, pk_m, P, rays, z_base, u1, u2, 128, NEAR, FAR, False, workspace=ws, ipe_radius=radius)
    rgb_pe, _, _, ws = A.ops.render_rays(pk_p, pk_m, P, rays, z_base, u1, u2, 128, NEAR, FAR, True, workspace=ws)
    acc = w.sum(-1)
    assert bool(torch.isfinite(rgb_w).all()) and bool(torch.isfinite(depth).all())
    assert float(acc.max()) <= 1.0 + 1e-4 and float(w.min()) >= 0.0
    assert max_abs(rgb_w - rgb_b, (1.0 - acc)[:, None].expand(-1, 3)) <= 2e-6
    assert float((rgb_w - rgb_pe).abs().max()) > 1e-3                                  # not the point-encoded image
    dn = A.ops.dirs_norm(rays)
    assert abs(dn.item() - rays[:, 3:].double().norm().item()) <= 1e-6 * dn.item()
    pick = torch.randperm(n, generator=torch.Generator().manual_seed(3))[:192]
    with torch.no_grad():
        want_rgb, want_w, want_depth = O.render_rays(W.proposal_state("he"), W.mip_state("he"), rays[pick].cpu(), u1[pick].cpu(),
                                                     u2[pick].cpu(), NEAR, FAR, 128, white_bkg=True, ipe_radius=radius, ipe_dir_norm=dn.cpu()[0])
    if prec == "fp32":                                        # ('he' + IPE: see test_he_weights_conditioning for what 1e-4 means on this set)
        assert max_abs(rgb_w[pick].cpu(), want_rgb) <= 2e-4 and max_abs(depth[pick].cpu(), want_depth) <= 2e-4
        assert max_abs(w[pick].cpu(), want_w) <= 2e-4
    else:
        assert torch.mean((rgb_w[pick].cpu() - want_rgb) ** 2).item() <= 1e-3


def test_render_image_ipe_flag(A):
    """Drop-in surface of config 3: render_image(..., ipe=True) = the reference's signature plus the flag; same tiling and RNG protocol."""
    prop, mip = build_nets(A, "he")
    A.pkg.set_precision("fp32")
    pose = dev(O.pose_spherical(40.0, -30.0, 4.0)[:3])
    f = O.fov2focal(0.6911112070083618, (100, 100))
    torch.manual_seed(5)
    with torch.no_grad():
        a = A.procedures.render_image(mip, prop, pose, 100, f, NEAR, FAR, 64, white_bkg=True, render_depth=True, ipe=True)
    torch.manual_seed(5)
    with torch.no_grad():
        b = A.procedures.render_image(mip, prop, pose, 100, f, NEAR, FAR, 64, white_bkg=True, render_depth=True)
    assert list(a.keys()) == ["rgb", "depth_img"] and a["rgb"].shape == (3, 100, 100)
    assert bool(torch.isfinite(a["rgb"]).all()) and float((a["rgb"] - b["rgb"]).abs().max()) > 1e-4
    assert bool(torch.isfinite(a["depth_img"]).all())


# ------------------------------------------------------------------------------------------------ weights are differentiable outputs of render
def test_render_weights_carry_gradient(A):
    """NeRF.render returns (rgb, weights, extras); the reference's Ref-NeRF step feeds the UN-detached weights into WeightedNormalLoss /
    BackFaceLoss (train.py:183-184), so d(loss)/d(rgbo) has a path through them.  HIP backward (nerf_amd_composite_backward with
    d_weights) against torch.autograd of the reference expression, for a loss that uses rgb only, weights only, and both."""
    gen = torch.Generator().manual_seed(31)
    N, S = 67, 96
    rgbo0 = torch.cat((torch.rand(N, S, 3, generator=gen), torch.randn(N, S, 1, generator=gen) * 2), -1)
    z = torch.sort(NEAR + (FAR - NEAR) * torch.rand(N, S, generator=gen), -1)[0]
    d = F.normalize(torch.randn(N, 3, generator=gen), dim=-1) * 1.3
    cw, cr = torch.rand(N, S, generator=gen), torch.rand(N, 3, generator=gen)
    for use_rgb, use_w in ((True, False), (False, True), (True, True)):
        ref = rgbo0.clone().requires_grad_(True)
        rgb_r, w_r, _ = O.composite(ref, z, d, white_bkg=True)
        ((rgb_r * cr).sum() * float(use_rgb) + (w_r * cw).sum() * float(use_w)).backward()
        x = dev(rgbo0).requires_grad_(True)
        rgb_g, w_g, _ = A.nerf_base.NeRF.render(x, dev(z), dev(d), white_bkg=True)
        assert w_g.requires_grad
        loss = 0.0
        if use_rgb:
            loss = loss + (rgb_g * dev(cr)).sum()
        if use_w:
            loss = loss + (w_g * dev(cw)).sum()
        loss.backward()
        scale = ref.grad.abs().max().item()
        assert max_abs(x.grad.cpu(), ref.grad) <= 2e-5 * max(1.0, scale), (use_rgb, use_w)


def test_config0_training_iteration_vs_reference_golden(A, golden):
    """BASELINE configs[0] at ITS shape on the HIP path: one whole iteration of train.py:151-218 -- randomFromOneImage on a 200 x 200 image,
    validSampler with the reference's CPU stream (256 rays, 32 coarse depths), both networks, inverseSample (65 draws), compositing, the
    two losses, backward through the hand-written kernels, DecayLrScheduler + the one-launch Adam -- against the REAL reference's run of
    the same iteration (golden G24; weak #3 of round 5's review: the 32-ray G14 was the only train-step golden)."""
    from nerf_amd.nerf_base import DecayLrScheduler
    from nerf_amd.optim import Adam
    g = golden("g24_config0_train_step")
    N = 256
    prop, mip = build_nets(A, "small")
    prop.train(); mip.train()
    A.pkg.set_precision("fp32")
    lr = 5e-4 * N / 512                                                                  # train.py:56
    opt = Adam(list(mip.parameters()) + list(prop.parameters()), lr=lr, betas=(0.9, 0.999))
    sch = DecayLrScheduler(0.01, 0.1, 100000, lr, 500)
    pix, coords = A.utils.randomFromOneImage(dev(g["img"]), (1.0, 1.0))
    torch.manual_seed(240)
    pts, z_c, tgt, rays = A.utils.validSampler(pix, coords, dev(g["pose"]), N, 32, tuple(g["focal"].tolist()), NEAR, FAR, True, rng="reference")
    assert torch.equal(z_c.cpu(), g["z_coarse"]) and torch.equal(tgt.cpu(), g["rgb_tgt"]) and max_abs(rays.cpu(), g["rays"]) <= 1e-6
    density = F.softplus(prop.forward(pts))
    pw = A.mip_methods.maxBlurFilter(A.addtional.ProposalNetwork.get_weights(density, z_c, rays[:, 3:]), 0.01)
    torch.manual_seed(241)
    fl, below = A.utils.inverseSample(pw, z_c, 65, sort=True)                             # the reference's own draw (utils.py:115) from the CPU generator
    fl = fl[..., :-1].contiguous()
    assert max_abs(fl.cpu(), g["z_fine"]) <= 2e-5 and (below.cpu() != g["below"]).float().mean().item() <= 0.01
    rgbo = mip.forward(A.nerf_base.NeRF.length2pts(rays, fl))
    rend, wts, _ = A.nerf_base.NeRF.render(rgbo, fl, rays[:, 3:])
    img_loss = torch.nn.MSELoss()(rend, tgt)
    p_loss = A.addtional.ProposalLoss()(A.addtional.getBounds(pw, below), wts.detach())
    opt.zero_grad()
    (p_loss + img_loss).backward()
    gate("config0 train step: rendered vs reference", max_abs(rend.detach().cpu(), g["rendered"]), 1e-5)
    gate("config0 train step: weights vs reference", max_abs(wts.detach().cpu(), g["weights"]), 1e-5)
    assert abs(img_loss.item() - g["img_loss"]) <= 1e-6 and abs(p_loss.item() - g["prop_loss"]) <= 1e-4 * max(1.0, g["prop_loss"])
    # Gradients.  The yardstick is the EXACT value of the same step (fp64 oracle + torch.autograd on the CPU, same fine depths and bins): tensors
    # whose entries are sums of thousands of cancelling terms carry fp32 summation-order noise in EVERY fp32 evaluation -- the reference's own
    # opacity-head gradient (max entry 2.3e-9) sits 8.6 % from the exact value at this batch -- so the HIP gradient must be within 1e-4 of the
    # reference's where that is conditioned (rgb / proposal heads) and no further from the exact value than 2x the reference is elsewhere.
    d64 = lambda sd: {k: v.double().requires_grad_(True) for k, v in sd.items()}
    p64, m64 = d64(W.proposal_state("small")), d64(W.mip_state("small"))
    r64, z64, fl64 = g["rays"].double(), g["z_coarse"].double(), fl.detach().cpu().double()
    dens64 = F.softplus(O.proposal_forward(p64, r64[:, None, :3] + r64[:, None, 3:] * z64[:, :, None]))
    pw64 = O.max_blur(O.sigma_to_weights(dens64, z64, r64[:, 3:]), 0.01)
    rend64, wts64, _ = O.composite(O.mip_forward(m64, O.length2pts(r64, fl64)), fl64, r64[:, 3:])
    (torch.mean((rend64 - g["rgb_tgt"].double()) ** 2) + O.proposal_loss(O.get_bounds(pw64, below.cpu()), wts64.detach())).backward()
    rel = lambda got, want: max_abs(got.cpu(), want) / max(want.abs().max().item(), 1e-30)
    gate("config0 train step: d rgb_layer.2 vs reference", rel(mip.rgb_layer[2].weight.grad, g["g_mip_rgb"]), 1e-4)
    gate("config0 train step: d proposal head vs reference", rel(prop.layers[8].weight.grad, g["g_prop_head"]), 1e-4)
    for name, have, gold, exact in (("opacity head", mip.opacity_head[0].weight.grad, g["g_mip_sigma"], m64["opacity_head.0.weight"].grad),
                                    ("skip layer rows 0-7", mip.lin_block2[0].weight.grad[:8], g["g_mip_skip"], m64["lin_block2.0.weight"].grad[:8]),
                                    ("first layer rows 0-7", mip.lin_block1[0].weight.grad[:8], g["g_mip_l1"], m64["lin_block1.0.weight"].grad[:8]),
                                    ("proposal first layer rows 0-7", prop.layers[0].weight.grad[:8], g["g_prop_l0"], p64["layers.0.weight"].grad[:8])):
        top = exact.abs().max().item()
        ref_exact = (gold.double() - exact).abs().max().item() / top
        hip_exact = (have.detach().cpu().double() - exact).abs().max().item() / top
        gate("config0 train step: d %s, HIP vs fp64 (the reference's fp32 is %.1e from it)" % (name, ref_exact), hip_exact, max(2.0 * ref_exact, 2e-5))
    _, lr0 = sch.update_opt_lr(0, opt)
    assert lr0 == g["lr"]
    opt.step()
    assert max_abs(mip.rgb_layer[2].weight.detach().cpu(), g["p_mip_rgb_after"]) <= 1e-8                 # |step| = lr0 sign(g): the same parameters
    assert max_abs(prop.layers[8].weight.detach().cpu(), g["p_prop_head_after"]) <= 1e-8
    assert max_abs(mip.lin_block1[0].weight.detach().cpu()[:8], g["p_mip_l1_after"]) <= 2.5 * lr0        # (a sign flip of a cancelling entry = 2 lr0)
    assert max_abs(prop.layers[0].weight.detach().cpu()[:8], g["p_prop_l0_after"]) <= 2.5 * lr0


# ------------------------------------------------------------------------------------------------ optimizer
def test_fused_adam_equals_torch_adam(A):
    """nerf_amd.optim.Adam (one HIP launch over all tensors, device-side step counter) against torch.optim.Adam on identical
    gradients: parameters and both moment tensors after several steps with a changing learning rate (nerf_base.DecayLrScheduler
    rewrites param_groups['lr'] every iteration, train.py:200-218), and state_dict interchange in both directions."""
    from nerf_amd.optim import Adam
    gen = torch.Generator().manual_seed(8)
    shapes = [(256, 63), (256,), (1, 256), (1,), (128, 283), (3,)]
    init = [torch.randn(s, generator=gen) * 0.1 for s in shapes]
    pa = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    pb = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    oa, ob = Adam(pa, lr=3e-4), torch.optim.Adam(pb, lr=3e-4)
    for it in range(7):
        lr = 3e-4 * (0.5 + 0.1 * it)
        for o in (oa, ob):
            for gr in o.param_groups:
                gr["lr"] = lr
        for x, y in zip(pa, pb):
            gq = (torch.randn(x.shape, generator=gen) * (10.0 ** ((it % 3) - 2))).cuda()
            x.grad, y.grad = gq.clone(), gq.clone()
        v0 = pa[0]._version
        oa.step(); ob.step()
        assert pa[0]._version > v0                              # the packed-weight caches see the update
        for x, y in zip(pa, pb):
            assert max_abs(x.detach().cpu(), y.detach().cpu()) <= 2e-6 * max(1.0, y.abs().max().item()), it
    for x, y in zip(pa, pb):
        assert max_abs(oa.state[x]["exp_avg"].cpu(), ob.state[y]["exp_avg"].cpu()) <= 1e-6 * max(1e-3, ob.state[y]["exp_avg"].abs().max().item())
        assert max_abs(oa.state[x]["exp_avg_sq"].cpu(), ob.state[y]["exp_avg_sq"].cpu()) <= 1e-6 * max(1e-6, ob.state[y]["exp_avg_sq"].abs().max().item())
        assert float(oa.state[x]["step"]) == float(ob.state[y]["step"]) == 7.0
    # checkpoints travel both ways (nerf_helper.saveModel stores opt.state_dict(), nerf_base.loadFromFile restores it): a state written
    # by either implementation, loaded into both, continues identically from identical parameters
    for src_opt, src_params in ((ob, pb), (oa, pa)):
        pc = [torch.nn.Parameter(t.detach().clone()) for t in src_params]
        pd = [torch.nn.Parameter(t.detach().clone()) for t in src_params]
        oc, od = Adam(pc, lr=1e-4), torch.optim.Adam(pd, lr=1e-4)
        import copy                                             # (load_state_dict keeps device tensors by reference: give each its own copy)
        oc.load_state_dict(copy.deepcopy(src_opt.state_dict())); od.load_state_dict(copy.deepcopy(src_opt.state_dict()))
        for x, y in zip(pc, pd):
            gq = torch.randn(x.shape, generator=gen).cuda() * 0.01
            x.grad, y.grad = gq.clone(), gq.clone()
        oc.step(); od.step()
        for x, y in zip(pc, pd):
            assert max_abs(x.detach().cpu(), y.detach().cpu()) <= 2e-6 * max(1.0, y.abs().max().item())
        assert float(oc.state[pc[0]]["step"]) == float(od.state[pd[0]]["step"]) == 8.0


def test_decay_phase_learning_rate_and_update_equal_the_oracles_optimizer(A):
    """VERDICT r5 item 1, first check: in the DECAY phase of the schedule (nerf_base.py:115-134, train.py:200-218) the learning rate the
    HIP Adam kernel multiplies with is the python double the reference's scheduler computes, to the last bit -- the device scalar of
    `Adam(lr_on_device=True)` is a float64 -- and the update it applies equals the one of torch.optim.Adam ON THE CPU (single-tensor
    path: the optimizer the oracle trains with) from a late-run state (step count 6 000+, bias corrections ~1, small gradients), eager
    and replayed from a hipGraph.  Parameters agree to one fp32 ulp of their magnitude after 40 steps; a schedule rounded to fp32 on its
    way to the device would fail the first assertion."""
    import copy
    from nerf_amd.nerf_base import DecayLrScheduler
    from nerf_amd.optim import Adam
    gen = torch.Generator().manual_seed(21)
    shapes = [(256, 63), (256,), (256, 256), (1, 256), (3, 128)]
    init = [torch.randn(s, generator=gen) * 0.05 for s in shapes]
    grads = [[torch.randn(s, generator=gen) * 1e-3 for s in shapes] for _ in range(40)]
    sched = DecayLrScheduler(0.01, 0.1, 3000, 4.5e-4, warmup_step=200)
    start = 6200                                                                    # decayed = 0.1 ** 2 at iteration 6200
    cpu = [torch.nn.Parameter(t.clone()) for t in init]
    oc = torch.optim.Adam(cpu, lr=4.5e-4)
    for q in cpu:                                                                    # a late-run state: moments of the gradient scale, step count 6 200
        oc.state[q]["step"] = torch.tensor(float(start))
        oc.state[q]["exp_avg"] = torch.randn(q.shape, generator=gen) * 1e-4
        oc.state[q]["exp_avg_sq"] = torch.rand(q.shape, generator=gen) * 1e-6 + 1e-9
    sd = copy.deepcopy(oc.state_dict())
    results = {}
    for mode in ("eager", "graph"):
        hip = [torch.nn.Parameter(t.clone().cuda()) for t in init]
        oh = Adam(hip, lr=4.5e-4, lr_on_device=True)
        oh.load_state_dict(copy.deepcopy(sd))
        gbuf = [torch.zeros_like(q) for q in hip]
        for q, g in zip(hip, gbuf):
            q.grad = g
        graph = None
        for k in range(40):
            lr = sched.lr_at(start + k)
            assert lr < 4.5e-4 * 0.011 and lr > 4.5e-4 * 0.0099                    # deep in the decay
            for gr in oh.param_groups:
                gr["lr"] = lr
            for g, src in zip(gbuf, grads[k]):
                g.copy_(src)
            if mode == "graph" and k == 3:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    oh.step()
                # (capture does not execute: nothing to restore)
            if graph is not None:
                oh.sync_lr()
                graph.replay()
            else:
                oh.step()
            dev_lr = oh.param_groups[0]["_lr_dev"]
            assert dev_lr.dtype == torch.float64 and float(dev_lr.item()) == lr, (k, float(dev_lr.item()), lr)    # bit-equal python double
        results[mode] = [q.detach().cpu() for q in hip]
    for k in range(40):
        lr = sched.lr_at(start + k)
        for gr in oc.param_groups:
            gr["lr"] = lr
        for q, g in zip(cpu, grads[k]):
            q.grad = g.clone()
        oc.step()
    for mode, got in results.items():
        worst = max(((a - b.detach()).abs().max() / b.detach().abs().max()).item() for a, b in zip(got, cpu))
        gate("Adam decay phase, 40 steps, HIP (%s) vs torch CPU Adam: max |dp| / max |p|" % mode, worst, 2.4e-7)
    assert all(torch.equal(a, b) for a, b in zip(results["eager"], results["graph"]))


# ------------------------------------------------------------------------------------------------ in-kernel uniforms (Philox4x32-10)
def test_philox_render_equals_render_with_the_same_uniforms_as_tensors(A):
    """nerf_amd_render_rays with u_strat = u_inv = NULL draws every uniform inside the kernels; the oracle's numpy Philox
    (O.philox_uniforms, restated from the paper's constants) produces the same numbers as tensors, and feeding those tensors to the
    explicit-u path gives BIT-IDENTICAL rgb / depth / weights -- i.e. the generator, the counter layout and both consumers (proposal
    pass and resampling pass regenerate the same stratified depths) are pinned.  Also: replay from the seed, sub-batch invariance
    through the ray offset, and a different seed gives a different image."""
    prop, mip = build_nets(A, "he")
    rays, _, _ = _rays_and_u(700, 128, 3)
    rays = dev(rays)
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    P = A.ops.F32
    seed = 0x1234567ABCDEF
    u1, u2 = O.philox_uniforms(seed, 700, 0, 64, 129)
    a = A.ops.render_rays(prop.packed(P), mip.packed(P), P, rays, z_base, None, None, 128, NEAR, FAR, True, want_depth=True, want_weights=True, seed=seed)
    b = A.ops.render_rays(prop.packed(P), mip.packed(P), P, rays, z_base, dev(u1), dev(u2), 128, NEAR, FAR, True, want_depth=True, want_weights=True)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    c = A.ops.render_rays(prop.packed(P), mip.packed(P), P, rays, z_base, None, None, 128, NEAR, FAR, True, want_depth=True, want_weights=True, seed=seed)
    assert all(torch.equal(x, y) for x, y in zip(a[:3], c[:3]))                                  # replay
    lo, cnt = 123, 301                                                                           # any sub-batch, addressed by its ray offset
    d = A.ops.render_rays(prop.packed(P), mip.packed(P), P, rays[lo:lo + cnt].contiguous(), z_base, None, None, 128, NEAR, FAR, True,
                          want_depth=True, want_weights=True, seed=seed, rng_ray_offset=lo)
    assert all(torch.equal(x[lo:lo + cnt], y) for x, y in zip(a[:3], d[:3]))
    e = A.ops.render_rays(prop.packed(P), mip.packed(P), P, rays, z_base, None, None, 128, NEAR, FAR, True, seed=seed + 1)
    assert float((e[0] - a[0]).abs().max()) > 1e-3
    # the generic resample shape (K = 200 > 192: no register prefetch) and the proposal kernel on its own
    dens = A.ops.proposal_forward_samples(prop.packed(P), P, A.ops.samples_rays(rays, 64, z_base=z_base, z_jitter=0.02, seed=seed), (700, 64), "cuda")
    dens2 = A.ops.proposal_forward_samples(prop.packed(P), P, A.ops.samples_rays(rays, 64, z_base=z_base, u=dev(u1), z_jitter=0.02), (700, 64), "cuda")
    assert torch.equal(dens, dens2)
    _, u2b = O.philox_uniforms(seed, 700, 0, 64, 200)
    za = A.ops.resample(dens, None, z_base, None, 0.02, rays, None, 200, want_below=True, seed=seed)
    zb = A.ops.resample(dens, None, z_base, dev(u1), 0.02, rays, dev(u2b), 200, want_below=True)
    assert torch.equal(za[0], zb[0]) and torch.equal(za[1], zb[1])


@pytest.mark.parametrize("width", [128, 200])
def test_narrower_networks_through_zero_padding(A, width):
    """--prop_net_width / --nerf_net_width below 256 (procedures.py:176-177): the 256-wide kernels evaluate the zero-padded network, which
    is the same function.  Forward, end-to-end render and every parameter gradient of a training step against the CPU oracle / torch
    autograd on networks of that width (also a width that is not a multiple of 16)."""
    from nerf_amd.addtional import ProposalLoss, ProposalNetwork, getBounds
    from nerf_amd.mip_methods import maxBlurFilter
    from nerf_amd.mip_model import MipNeRF
    from nerf_amd.nerf_base import NeRF
    from nerf_amd.utils import inverseSample
    torch.manual_seed(100 + width)
    prop, mip = ProposalNetwork(10, width), MipNeRF(10, 4, width)
    with torch.no_grad():                                                     # O(1) activations so that every layer matters
        for m in list(prop.modules()) + list(mip.modules()):
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(4.0); m.bias.normal_(0.0, 0.05)
    psd = {k: v.detach().clone() for k, v in prop.state_dict().items()}
    msd = {k: v.detach().clone() for k, v in mip.state_dict().items()}
    prop, mip = prop.cuda(), mip.cuda()
    A.pkg.set_precision("fp32")
    rays, u1, u2 = _rays_and_u(200, 64, 61)
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    d64 = lambda sd: {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        want_rgb, want_w, ex = O.render_rays(psd, msd, rays, u1, u2, NEAR, FAR, 64, white_bkg=True)
        x_rgb, x_w, x_ex = O.render_rays(d64(psd), d64(msd), rays.double(), u1.double(), u2.double(), NEAR, FAR, 64, white_bkg=True)
    floor = max(max_abs(want_rgb, x_rgb), max_abs(want_w, x_w), max_abs(ex, x_ex))   # fp32 noise of the reference itself
    tol_img = max(1e-4, 1.5 * floor)
    prop.eval(); mip.eval()
    rgb, depth, w, _ = A.ops.render_rays(prop.packed(A.ops.F32), mip.packed(A.ops.F32), A.ops.F32, dev(rays), z_base, dev(u1), dev(u2), 64,
                                         NEAR, FAR, True, want_depth=True, want_weights=True)
    assert max_abs(rgb.cpu(), want_rgb) <= tol_img and max_abs(w.cpu(), want_w) <= tol_img and max_abs(depth.cpu(), ex) <= tol_img, (tol_img, floor)
    assert tol_img <= 5e-4
    # a training step (train.py:164-199): HIP gradients vs torch.autograd of the oracle on the same fine depths
    prop.train(); mip.train()
    n = 48
    r, zc, tgt = dev(rays[:n]), (z_base[None, :] + dev(u1[:n]) * (4.0 / 64)).contiguous(), torch.rand(n, 3).cuda()
    pts = (r[:, None, :3] + r[:, None, 3:] * zc[:, :, None]).contiguous()
    pw = maxBlurFilter(ProposalNetwork.get_weights(F.softplus(prop.forward(pts)), zc, r[:, 3:]), 0.01)
    fl, below = inverseSample(pw, zc, 65, sort=True, u=u2[:n])
    fl = fl[..., :-1].contiguous()
    rend, wts, _ = NeRF.render(mip.forward(NeRF.length2pts(r, fl)), fl, r[:, 3:])
    (ProposalLoss()(getBounds(pw, below), wts.detach()) + torch.mean((rend - tgt) ** 2)).backward()
    p64 = {k: v.double().requires_grad_(True) for k, v in psd.items()}
    m64 = {k: v.double().requires_grad_(True) for k, v in msd.items()}
    r64, z64, fl64 = r.cpu().double(), zc.cpu().double(), fl.detach().cpu().double()
    pw64 = O.max_blur(O.sigma_to_weights(F.softplus(O.proposal_forward(p64, r64[:, None, :3] + r64[:, None, 3:] * z64[:, :, None])), z64, r64[:, 3:]), 0.01)
    rend64, wts64, _ = O.composite(O.mip_forward(m64, O.length2pts(r64, fl64)), fl64, r64[:, 3:])
    (torch.mean((rend64 - tgt.cpu().double()) ** 2) + O.proposal_loss(O.get_bounds(pw64, below.cpu()), wts64.detach())).backward()
    for net, ref in ((prop, p64), (mip, m64)):
        for name, prm in net.named_parameters():
            want = ref[name].grad
            assert prm.grad is not None and tuple(prm.grad.shape) == tuple(want.shape), name
            top = want.abs().max().item()
            # (fp32 kernels against the fp64 value: first-layer tensors carry cancellation noise of a few 1e-3, test_train_step_gradients)
            assert (prm.grad.cpu().double() - want).abs().max().item() <= 1e-2 * max(top, 1e-12), (name, top)


def test_philox_render_refnerf_and_many_fine_samples(A):
    """In-kernel uniforms on the Ref-NeRF render entry (nerf_amd_render_rays_ref with NULL uniform tensors) and beyond 256 inverse-CDF
    draws per ray (blocks 64.. of the counter layout): both equal the tensor-fed render of oracle.philox_uniforms' values bit for bit."""
    prop, mip = build_nets(A, "small")
    net = build_ref(A, "small")
    rays, _, _ = _rays_and_u(260, 64, 23)
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    seed, off = 0x1234ABCD5678, 1000
    u1, u2 = O.philox_uniforms(seed, 260, off, 64, 65)
    a = A.ops.render_rays_ref(prop.packed(A.ops.F32), net.packed(A.ops.F32), A.ops.F32, dev(rays), z_base, None, None, 64, NEAR, FAR, True,
                              seed=seed, rng_ray_offset=off)
    b = A.ops.render_rays_ref(prop.packed(A.ops.F32), net.packed(A.ops.F32), A.ops.F32, dev(rays), z_base, dev(u1), dev(u2), 64, NEAR, FAR, True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    u1, u2 = O.philox_uniforms(seed, 260, off, 64, 401)                      # n_fine = 400: u_inv blocks 0..2 of every ray
    a = A.ops.render_rays(prop.packed(A.ops.F32), mip.packed(A.ops.F32), A.ops.F32, dev(rays), z_base, None, None, 400, NEAR, FAR, True,
                          seed=seed, rng_ray_offset=off)
    b = A.ops.render_rays(prop.packed(A.ops.F32), mip.packed(A.ops.F32), A.ops.F32, dev(rays), z_base, dev(u1), dev(u2), 400, NEAR, FAR, True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_philox_uniform_tensor_and_device_seed(A):
    """nerf_amd_philox_uniforms == the oracle's restatement of the inverse-CDF stream, bit for bit, with the key given as an argument or
    read from device memory; nerf_amd_advance_seed replaces the device key by a different one, deterministically."""
    seed = 0x0F1E2D3C4B5A6978
    want = O.philox_uniforms(seed, 37, 0, 64, 250)[1]
    got = A.ops.philox_uniforms((37, 250), seed=seed, device=torch.device("cuda", 0))
    assert torch.equal(got.cpu(), want)
    sd = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed], dtype=torch.int64).cuda()
    assert torch.equal(A.ops.philox_uniforms((37, 250), seed_dev=sd).cpu(), want)
    A.ops.advance_seed(sd)
    s1 = int(sd.item())
    assert s1 != seed and not torch.equal(A.ops.philox_uniforms((37, 250), seed_dev=sd).cpu(), want)
    sd2 = torch.tensor([seed], dtype=torch.int64).cuda()
    A.ops.advance_seed(sd2)
    assert int(sd2.item()) == s1
    # (ABI 121) nerf_amd_philox_stream: both streams for rows that are GLOBAL rays off .. off + N - 1
    for off in (0, 4096, (1 << 33) + 5):
        ws, wi = O.philox_uniforms(seed, 29, off, 64, 193)
        assert torch.equal(A.ops.philox_stream((29, 64), seed, off, strat=True).cpu(), ws)
        assert torch.equal(A.ops.philox_stream((29, 193), seed, off).cpu(), wi)


def test_generic_route_draws_the_fused_kernels_philox_streams(A):
    """on the layer-by-layer route `rng="philox"` used to become torch.rand on the device generator (the seed drawn from the CPU
    generator unused).  It now draws the render kernels' own streams per chunk (global ray index), so (i) a seeded generic render is
    reproducible under torch.manual_seed alone and (ii) it is the image the SAME pipeline produces from the explicit Philox tensors."""
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.mip_model import MipNeRF
    from nerf_amd.procedures import render_image, _render_rays_by_calls
    torch.manual_seed(3)
    mip, prop = MipNeRF(10, 4, 320).cuda().eval(), ProposalNetwork(10, 256).cuda().eval()           # width 320: generic route
    assert mip._generic()
    pose = O.pose_spherical(20.0, -30.0, 4.0).cuda()
    focal = O.fov2focal(0.6911112070083618, (100, 100))
    with torch.no_grad():
        torch.manual_seed(11); a = render_image(mip, prop, pose, 100, focal, NEAR, FAR, 64, white_bkg=True)["rgb"]
        torch.cuda.manual_seed(999)                                                                   # the device generator must not matter
        torch.manual_seed(11); b = render_image(mip, prop, pose, 100, focal, NEAR, FAR, 64, white_bkg=True)["rgb"]
        torch.manual_seed(12); c = render_image(mip, prop, pose, 100, focal, NEAR, FAR, 64, white_bkg=True)["rgb"]
        assert torch.equal(a, b) and not torch.equal(a, c)
        torch.manual_seed(11)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        fx, fy = (float(focal[1]), float(focal[0])) if isinstance(focal, (tuple, list)) else (float(focal), float(focal))
        rays = A.ops.generate_rays(pose[:3], 100, 100, fx, fy, pose.device)
        rays = rays.view(100, 100, 6).reshape(2, 50, 2, 50, 6).permute(0, 2, 1, 3, 4).reshape(-1, 6).contiguous()
        z_base = torch.linspace(NEAR, FAR, 64).cuda()
        u1, u2 = A.ops.philox_stream((10000, 64), seed, 0, strat=True), A.ops.philox_stream((10000, 65), seed, 0)
        rgb, _, _ = _render_rays_by_calls(mip, prop, rays, z_base, u1, u2, 64, NEAR, FAR, True, False)
        img = torch.zeros(3, 100, 100, device="cuda")
        img[:] = rgb.view(2, 2, 50, 50, 3).permute(4, 0, 2, 1, 3).reshape(3, 100, 100)
        assert torch.equal(a, img)


def test_bottle_neck_noise_in_kernel_philox(A):
    """Ref-NeRF's train-mode bottle-neck perturbation (ref_model.py:84-85) drawn INSIDE the training forward (round 5): (i) the deviates as
    a tensor (nerf_amd_philox_normal) equal the oracle's restatement of Philox -> 16-bit uniforms -> Box-Muller to the hardware
    transcendentals' accuracy, for any sample offset and with the key in device memory; (ii) the training forward that draws them in
    place is BIT-IDENTICAL -- outputs, aux record, activation dump, ReLU masks -- to the same forward handed that tensor, fp32 and bf16;
    (iii) the module: reproducible under torch.manual_seed, another seed = another perturbation, `noise_rng = "torch"` restores the
    device-generator draw; gradients flow (the backward needs no noise: the dump holds the perturbed bottle-neck)."""
    seed, std = 0x1234567890ABCDEF, 0.1
    want = O.philox_normal(seed, 300, std, 0)
    got = A.ops.philox_normal(300, std, seed)
    assert max_abs(got.cpu(), want) <= 2e-5 * std * 10 and float(got.std()) > 0.09
    off = (1 << 33) + 77
    assert max_abs(A.ops.philox_normal(40, std, seed, sample_offset=off).cpu(), O.philox_normal(seed, 40, std, off)) <= 2e-5
    assert torch.equal(A.ops.philox_normal(100, std, seed)[60:], A.ops.philox_normal(40, std, seed, sample_offset=60))
    sd = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed], dtype=torch.int64).cuda()
    assert torch.equal(A.ops.philox_normal(300, std, seed_dev=sd), got)
    net = build_ref(A, "small").train()
    gen = torch.Generator().manual_seed(9)
    M = 1000                                                  # ragged: not a tile multiple
    pts = torch.cat((torch.rand(M, 3, generator=gen) * 2 - 1, F.normalize(torch.randn(M, 3, generator=gen), dim=-1)), -1).cuda()
    import gc
    for P in (A.ops.F32, A.ops.BF16):
        # The dump holds bytes no kernel writes (the padding samples of a ragged last tile, K groups a slot does not use), so "bit-identical
        # dumps" is only defined when both forwards write into the SAME buffer: nothing of an earlier forward may still hold the persistent
        # dump's lease (round 6: the comparison used to depend on which recycled block torch.empty handed a second, unleased dump)
        a = b = c = e = x = y = None
        gc.collect()
        blob = net.packed(P)
        noise = A.ops.philox_normal(M, std, seed)
        a = A.ops.ref_forward_train(blob, P, pts, noise, net.kernel_flags)
        dump_ptr = a[2].data_ptr()
        a = [t.clone() for t in a]                            # (the dump is a lease on a persistent buffer: copy before the next forward reuses it)
        gc.collect()
        b = A.ops.ref_forward_train(blob, P, pts, None, net.kernel_flags, noise_std=std, noise_seed=seed)
        assert b[2].data_ptr() == dump_ptr                    # the same persistent buffer: unwritten bytes are the first forward's
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        c = A.ops.ref_forward_train(blob, P, pts, None, net.kernel_flags, noise_std=std, noise_seed_dev=sd)
        assert torch.equal(a[0], c[0])
        e = A.ops.ref_forward_train(blob, P, pts, None, net.kernel_flags)                    # no perturbation at all
        assert not torch.equal(a[0], e[0])
    a = b = c = e = x = y = None
    A.pkg.set_precision("fp32")
    pos, d = pts[None, :, :3].contiguous(), pts[None, :, 3:].contiguous()
    outs = []
    for s_ in (11, 11, 12):
        torch.manual_seed(s_)
        p_ = pos.clone().requires_grad_(True)
        rgbo, nrm = net.forward(p_, d)
        outs.append(rgbo.detach().clone())
        (rgbo.sum() + nrm.sum()).backward()
        assert float(net.bottle_neck.weight.grad.abs().max()) > 0
        net.zero_grad()
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    with torch.no_grad():                                     # train mode without gradients: the same perturbation as a tensor into the plain kernel
        torch.manual_seed(11)
        rn, _ = net.forward(pos, d)
    assert max_abs(rn, outs[0]) <= 1e-6
    net.noise_rng = "torch"
    torch.manual_seed(11); torch.cuda.manual_seed(3)
    t1 = net.forward(pos.clone().requires_grad_(True), d)[0].detach().clone()
    torch.manual_seed(11); torch.cuda.manual_seed(4)
    t2 = net.forward(pos.clone().requires_grad_(True), d)[0].detach().clone()
    assert not torch.equal(t1, t2) and not torch.equal(t1, outs[0])       # (the device generator's stream now)


def test_philox_uniforms_are_uniform():
    """Statistical sanity of the in-kernel stream (through its oracle twin, bit-equal to the kernels by the test above): one-sample
    Kolmogorov-Smirnov against U[0,1) on 1.2e6 draws of each stream, lag-1 / cross-stream correlations, and the 24-bit lattice."""
    import numpy as np
    from scipy import stats
    u1, u2 = O.philox_uniforms(20260928, 6400, 77, 64, 129)
    for u in (u1.numpy().ravel(), u2.numpy().ravel()):
        assert u.min() >= 0.0 and u.max() < 1.0
        assert stats.kstest(u.astype(np.float64), "uniform").pvalue > 1e-3
        assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 5e-3
        assert np.all(u * 2.0 ** 24 == np.floor(u * 2.0 ** 24))
    assert abs(np.corrcoef(u1.numpy()[:, :64].ravel(), u2.numpy()[:, :64].ravel())[0, 1]) < 5e-3
    assert abs(np.corrcoef(u1.numpy()[:-1].ravel(), u1.numpy()[1:].ravel())[0, 1]) < 5e-3            # neighbouring rays


def test_render_image_default_rng_is_in_kernel_and_seedable(A):
    """The drop-in render_image now draws its uniforms in the kernels by default: reproducible under torch.manual_seed, different
    across seeds, statistically the same image as the reference-stream render (both are Monte-Carlo estimates of one integral)."""
    prop, mip = build_nets(A, "he")
    A.pkg.set_precision("fp32")
    pose = dev(O.pose_spherical(40.0, -30.0, 4.0)[:3])
    f = O.fov2focal(0.6911112070083618, (100, 100))
    imgs = []
    for seed in (5, 5, 6):
        torch.manual_seed(seed)
        with torch.no_grad():
            imgs.append(A.procedures.render_image(mip, prop, pose, 100, f, NEAR, FAR, 64, white_bkg=True, render_depth=True))
    assert torch.equal(imgs[0]["rgb"], imgs[1]["rgb"]) and torch.equal(imgs[0]["depth_img"], imgs[1]["depth_img"])
    assert float((imgs[0]["rgb"] - imgs[2]["rgb"]).abs().max()) > 1e-4
    torch.manual_seed(5)
    with torch.no_grad():
        ref = A.procedures.render_image(mip, prop, pose, 100, f, NEAR, FAR, 64, white_bkg=True, render_depth=True, rng="reference")
    # (per-pixel values are single-sample Monte-Carlo estimates through a random network: only image statistics are comparable)
    assert float((imgs[0]["rgb"].mean() - ref["rgb"].mean()).abs()) < 5e-3 and float((imgs[0]["rgb"].std() - ref["rgb"].std()).abs()) < 1e-2


def test_he_weights_conditioning(A):
    """What the 1e-4 gate means on the 'he' stress weights (O(1) activations through ten positional-encoding octaves).  The same
    rays are rendered by the HIP fp32 path, by the CPU oracle in fp32 (the reference's arithmetic) and by the CPU oracle in fp64 (the
    exact value of the same expressions).  The fp32 reference itself is ~1e-4 from the exact value: on this weight set 1e-4 is the
    noise floor of fp32 evaluation (MKL's and the MFMA chain's summation orders differ, the 2^9 octave amplifies a position ulp to
    2e-4 rad), not slack in the kernels.  Gates: the HIP path differs from the fp32 reference by no more than 1.5 x that floor, and is
    as close to the exact value as the reference is (factor 2).  The other 'he' tests therefore use 2e-4; 'small' (reference-style)
    weights hold 1e-4 with two orders of magnitude to spare."""
    prop, mip = build_nets(A, "he")
    psd, msd = W.proposal_state("he"), W.mip_state("he")
    rays, u1, u2 = _rays_and_u(300, 128, 41)
    dd = lambda sd: {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        r32 = O.render_rays(psd, msd, rays, u1, u2, NEAR, FAR, 128, white_bkg=True)
        r64 = O.render_rays(dd(psd), dd(msd), rays.double(), u1.double(), u2.double(), NEAR, FAR, 128, white_bkg=True)
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    rgb, depth, w, _ = A.ops.render_rays(prop.packed(A.ops.F32), mip.packed(A.ops.F32), A.ops.F32, dev(rays), z_base, dev(u1), dev(u2), 128,
                                         NEAR, FAR, True, want_depth=True, want_weights=True)
    for got, a32, a64 in ((rgb.cpu(), r32[0], r64[0]), (w.cpu(), r32[1], r64[1]), (depth.cpu(), r32[2], r64[2])):
        hip_ref, ref_exact, hip_exact = max_abs(got, a32), max_abs(a32, a64), max_abs(got, a64)
        assert ref_exact >= 5e-5                                 # the reference's own fp32 noise floor on this set
        assert hip_ref <= max(1e-4, 1.5 * ref_exact)              # HIP differs from the fp32 reference by no more than that floor
        assert hip_exact <= 2.0 * ref_exact


def test_no_torch_fallback_in_the_product(A):
    """"No library GEMM anywhere" is a property, not a coincidence of the tested shapes.  (i) The product contains no torch re-evaluation of
    an op: nerf_amd/autograd_bridge.py has no expression, no Linear, no VJP -- HipOp.backward is the HIP backward or NotImplementedError.
    (ii) A differentiable call the HIP backward does not cover (a compositing row longer than BWD_MAX_SAMPLES; MipNeRF positions that
    require a gradient) is refused when the FORWARD is called, not inside loss.backward().  (iii) Under no_grad the same calls
    run.  (iv) An empty batch is differentiable (zero gradients) without launching anything."""
    import inspect
    import torch_spec as spec
    from nerf_amd import autograd_bridge as ab
    src = inspect.getsource(ab)
    for word in ("torch.mm", "torch.bmm", "F.linear", "addmm", "_expr", "enable_grad", "import torch.nn"):
        assert word not in src, word
    assert not hasattr(ab, "allow_torch_vjp") and not hasattr(ab, "mip_expr")
    A.pkg.set_precision("fp32")
    gen = torch.Generator().manual_seed(3)
    N, S = 9, A.ops.BWD_MAX_SAMPLES + 44
    rgbo = torch.cat((torch.rand(N, S, 3, generator=gen), torch.randn(N, S, 1, generator=gen)), -1).cuda().requires_grad_(True)
    z = torch.sort(torch.rand(N, S, generator=gen) * 4 + 2, dim=-1)[0].cuda()
    dirs = torch.randn(N, 3, generator=gen).cuda()
    with pytest.raises(NotImplementedError):
        A.nerf_base.NeRF.render(rgbo, z, dirs)
    with pytest.raises(NotImplementedError):
        A.nerf_base.NeRF.getNormedWeight(rgbo[..., 3], z)
    with torch.no_grad():
        rgb, w, _ = A.nerf_base.NeRF.render(rgbo, z, dirs)              # forward-only: any row length
    ref = rgbo.detach()
    want_w = spec.weights_expr(ref[..., 3], z * dirs.norm(dim=-1, keepdim=True), A.ops.ACT_RELU)
    assert max_abs(w.cpu(), want_w.cpu()) <= 2e-6
    # ... and a row the backward kernel does hold differentiates, equal to autograd of the specification
    S2 = A.ops.BWD_MAX_SAMPLES
    r2 = rgbo.detach()[:, :S2].clone().requires_grad_(True)
    rgb2, _, _ = A.nerf_base.NeRF.render(r2, z[:, :S2].contiguous(), dirs)
    rgb2.sum().backward()
    ref2 = rgbo.detach()[:, :S2].clone().requires_grad_(True)
    ww = spec.weights_expr(ref2[..., 3], z[:, :S2] * dirs.norm(dim=-1, keepdim=True), A.ops.ACT_RELU)
    (ww[:, :, None] * ref2[..., :3]).sum().backward()
    assert max_abs(r2.grad.cpu(), ref2.grad.cpu()) <= 2e-5 * max(1.0, ref2.grad.abs().max().item())
    _, mip = build_nets(A, "small")
    mip.train()
    pts = torch.randn(5, 7, 6, generator=gen).cuda().requires_grad_(True)
    with pytest.raises(NotImplementedError):
        mip.forward(pts)
    for p_ in mip.parameters():
        p_.grad = None
    out = mip.forward(torch.zeros((0, 7, 6), device="cuda"))              # empty batch under autograd: zero gradients, no launch
    assert out.shape == (0, 7, 4)
    out.sum().backward()
    assert all(p_.grad is not None and float(p_.grad.abs().max()) == 0.0 for p_ in mip.parameters())


@pytest.mark.parametrize("width", [128, 200])
def test_refnerf_narrower_network_through_zero_padding(A, width):
    """`-t --nerf_net_width W` with W < 256 (train.py:80: RefNeRF(10, ide_level, hidden_unit = W); the reference also needs output_dim = W):
    the 256-wide Ref-NeRF kernels evaluate the zero-padded network -- the same function.  Eval forward (colours, density, normals) against
    the oracle and every parameter gradient of a train-mode forward against fp64 autograd of the oracle's expression."""
    from nerf_amd.ref_model import RefNeRF
    torch.manual_seed(300 + width)
    net = RefNeRF(10, 4, hidden_unit=width, output_dim=width, perturb_bottle_neck_w=0.0)
    with torch.no_grad():                                                     # O(1) activations so that every layer matters
        for m in net.modules():
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(4.0); m.bias.normal_(0.0, 0.05)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    A.pkg.set_precision("fp32")
    gen = torch.Generator().manual_seed(width)
    pts = torch.cat((torch.randn(37, 9, 3, generator=gen), F.normalize(torch.randn(37, 9, 3, generator=gen), dim=-1)), -1)
    with torch.no_grad():
        want, want_n = O.ref_forward(sd, pts)
        got, got_n = net.eval().forward(pts.cuda())
    assert max_abs(got.cpu(), want) <= 2e-5 * max(1.0, want.abs().max().item()) and max_abs(got_n.cpu(), want_n) <= 2e-5
    net.train()
    g1, g2 = torch.randn(37, 9, 4, generator=gen), torch.randn(37, 9, 3, generator=gen)
    rgbo, nrm = net.forward(pts.cuda())
    ((rgbo * g1.cuda()).sum() + (nrm * g2.cuda()).sum()).backward()
    s64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    r64, n64 = O.ref_forward(s64, pts.double())
    ((r64 * g1.double()).sum() + (n64 * g2.double()).sum()).backward()
    for name, prm in net.named_parameters():
        wantg = s64[name].grad
        assert prm.grad is not None and tuple(prm.grad.shape) == tuple(wantg.shape), name
        top = wantg.abs().max().item()
        assert (prm.grad.cpu().double() - wantg).abs().max().item() <= 1e-2 * max(top, 1e-12), (name, top)
    with pytest.raises(NotImplementedError):
        RefNeRF(10, 4, hidden_unit=128, output_dim=256).cuda().eval().forward(pts.cuda())


@pytest.mark.parametrize("level,width", [(1, 256), (2, 256), (3, 256), (2, 128)])
def test_refnerf_other_ide_levels(A, level, width):
    """`--ide_level` 1..3 (procedures.py:211; RefNeRF(10, ide_level), train.py:80): the level-4 kernel evaluates the module with its
    directional layers embedded into the level-4 column layout (the terms of level d are a prefix of level 4's).  Eval forward against the
    oracle (which restates ref_func.py for any level) and parameter gradients against its fp64 autograd; level 5 raises."""
    from nerf_amd.ref_model import RefNeRF
    torch.manual_seed(500 + 10 * level + width)
    net = RefNeRF(10, level, hidden_unit=width, output_dim=width, perturb_bottle_neck_w=0.0)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(4.0); m.bias.normal_(0.0, 0.05)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    assert sd["dir_block1.0.weight"].shape[1] == 129 + 2 * ((1 << level) - 1 + level)
    net = net.cuda()
    A.pkg.set_precision("fp32")
    gen = torch.Generator().manual_seed(level)
    pts = torch.cat((torch.randn(29, 11, 3, generator=gen), F.normalize(torch.randn(29, 11, 3, generator=gen), dim=-1)), -1)
    with torch.no_grad():
        want, want_n = O.ref_forward(sd, pts, deg=level)
        got, got_n = net.eval().forward(pts.cuda())
    assert max_abs(got.cpu(), want) <= 2e-5 * max(1.0, want.abs().max().item()) and max_abs(got_n.cpu(), want_n) <= 2e-5
    net.train()
    g1, g2 = torch.randn(29, 11, 4, generator=gen), torch.randn(29, 11, 3, generator=gen)
    rgbo, nrm = net.forward(pts.cuda())
    ((rgbo * g1.cuda()).sum() + (nrm * g2.cuda()).sum()).backward()
    s64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    r64, n64 = O.ref_forward(s64, pts.double(), deg=level)
    ((r64 * g1.double()).sum() + (n64 * g2.double()).sum()).backward()
    for name, prm in net.named_parameters():
        wantg = s64[name].grad
        assert prm.grad is not None and tuple(prm.grad.shape) == tuple(wantg.shape), name
        # (O(1) activations + a loss on the normalised normals: a ReLU whose fp32 pre-activation is within rounding of zero falls on the other
        # side than in fp64 for isolated samples, which moves single entries by a few %; the tensors agree in the L2 sense)
        diff, top = prm.grad.cpu().double() - wantg, max(wantg.abs().max().item(), 1e-12)
        assert diff.norm().item() <= 3e-2 * max(wantg.norm().item(), 1e-12) and diff.abs().max().item() <= 0.1 * top, \
            "%s: |err|_2 %.3e of %.3e, max %.3e of %.3e" % (name, diff.norm().item(), wantg.norm().item(), diff.abs().max().item(), top)
    with pytest.raises(ValueError):                                                    # ref_func.py:63 "Only deg_view of at most 5 is numerically stable", like the
        RefNeRF(10, 6)                                                                 # reference; level 5 itself runs on the generic path (test_refnerf_outside_the_compiled_shapes)


@pytest.mark.parametrize("width", [128, 64, 100])
def test_narrow_proposal_tile_policy(A, width):
    """ProposalNetwork(10, hidden <= 128) -- the reference's class default (addtional.py:61), `--prop_net_width 128` (procedures.py:176) --
    is evaluated by its own kernel (NERF_AMD_NET_PROPOSAL_128: half the K groups and feature blocks, three column tiles per wave) on every
    forward-only path.  It must be THE SAME FUNCTION as the 256-wide kernel on zero-padded tensors -- bit for bit in both precisions: the
    added K groups only ever add zero products, in the same order -- and equal the oracle; ragged sizes; the fused stratified-sample
    fetch with in-kernel Philox draws; the whole render path; widths below 128 pad up to it."""
    from nerf_amd.addtional import ProposalNetwork
    torch.manual_seed(500 + width)
    prop = ProposalNetwork(10, width)
    with torch.no_grad():                                                     # O(1) activations so that every layer matters
        for m in prop.modules():
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(4.0); m.bias.normal_(0.0, 0.05)
    psd = {k: v.detach().clone() for k, v in prop.state_dict().items()}
    prop = prop.cuda().eval()
    gen = torch.Generator().manual_seed(width)
    for P, tol in ((A.ops.F32, 2e-5), (A.ops.BF16, 2e-2)):
        narrow, wide = prop.packed(P), prop.packed(P, wide=True)
        assert getattr(narrow, "_nerf_amd_layout", 0) == A.ops.PROP_W128 and getattr(wide, "_nerf_amd_layout", 0) == 0
        assert narrow.numel() == A.ops.lib.nerf_amd_packed_bytes(A.ops.NET_PROPOSAL_128, P) < wide.numel()
        for M in (1, 33, 383, 384, 385, 1000, 50001):
            pts = (torch.rand(M, 3, generator=gen) * 4.0 - 2.0).cuda()
            got_n = A.ops.proposal_forward(narrow, P, pts)
            got_w = A.ops.proposal_forward(wide, P, pts)
            assert torch.equal(got_n, got_w), (width, P, M, max_abs(got_n.cpu(), got_w.cpu()))
            if M <= 1000:
                with torch.no_grad():
                    want = O.proposal_forward(psd, pts.cpu(), emulate_bf16=(P == A.ops.BF16))
                scale = max(1.0, want.abs().max().item())
                assert max_abs(got_n.cpu(), want) <= tol * scale, (width, P, M)
    # the fused stratified fetch (rows 2-4) with in-kernel uniforms, and the whole render path through nerf_amd_render_rays
    _, mip = build_nets(A, "small")
    rays, _, _ = _rays_and_u(777, 64, 91)
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    for P in (A.ops.F32, A.ops.BF16):
        kw = dict(z_base=z_base, z_jitter=(FAR - NEAR) / 128, seed=99)
        d_n = A.ops.proposal_forward_samples(prop.packed(P), P, A.ops.samples_rays(dev(rays), 64, **kw), (777, 64), "cuda")
        d_w = A.ops.proposal_forward_samples(prop.packed(P, wide=True), P, A.ops.samples_rays(dev(rays), 64, **kw), (777, 64), "cuda")
        assert torch.equal(d_n, d_w)
        a = A.ops.render_rays(prop.packed(P), mip.packed(P), P, dev(rays), z_base, None, None, 128, NEAR, FAR, True, want_depth=True, seed=5)
        b = A.ops.render_rays(prop.packed(P, wide=True), mip.packed(P), P, dev(rays), z_base, None, None, 128, NEAR, FAR, True, want_depth=True, seed=5)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # a layout flag the entry point does not know is refused
    with pytest.raises(RuntimeError):
        A.ops.check(A.ops.lib.nerf_amd_proposal_forward(None, A.ops.F32 | 0x400, None, None, None), "nerf_amd_proposal_forward")


@pytest.mark.parametrize("width", [128, 64, 100])
def test_narrow_mip_tile_policy(A, width):
    """MipNeRF(10, 4, hidden <= 128) (`--nerf_net_width 128`, procedures.py:177) on its own kernel (NERF_AMD_NET_MIP_128: the 128-wide layers
    at half the K groups and feature blocks, lin_block2.4 widening to the 256-wide heads): the same function as the 256-wide kernel on
    zero-padded tensors bit for bit in both precisions, equal to the oracle, ragged sizes, rays + depths fetch, the whole render path
    with BOTH networks narrow; the integrated PE falls back to the 256-wide blob."""
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.mip_model import MipNeRF
    torch.manual_seed(700 + width)
    mip, prop = MipNeRF(10, 4, width), ProposalNetwork(10, width)
    with torch.no_grad():
        for m in list(mip.modules()) + list(prop.modules()):
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(4.0); m.bias.normal_(0.0, 0.05)
    msd = {k: v.detach().clone() for k, v in mip.state_dict().items()}
    psd = {k: v.detach().clone() for k, v in prop.state_dict().items()}
    mip, prop = mip.cuda().eval(), prop.cuda().eval()
    gen = torch.Generator().manual_seed(width + 1)
    for P, tol in ((A.ops.F32, 2e-5), (A.ops.BF16, 2e-2)):
        narrow, wide = mip.packed(P), mip.packed(P, wide=True)
        assert getattr(narrow, "_nerf_amd_layout", 0) == A.ops.FINE_W128 and getattr(wide, "_nerf_amd_layout", 0) == 0
        assert narrow.numel() == A.ops.lib.nerf_amd_packed_bytes(A.ops.NET_MIP_128, P) < wide.numel()
        for M in (1, 33, 255, 256, 257, 1000, 50001):
            pts = torch.cat((torch.rand(M, 3, generator=gen) * 4.0 - 2.0, torch.randn(M, 3, generator=gen)), -1).cuda()
            got_n, got_w = A.ops.mip_forward(narrow, P, pts), A.ops.mip_forward(wide, P, pts)
            assert torch.equal(got_n, got_w), (width, P, M, max_abs(got_n.cpu(), got_w.cpu()))
            if M <= 1000:
                with torch.no_grad():
                    want = O.mip_forward(msd, pts.cpu(), emulate_bf16=(P == A.ops.BF16))
                scale = max(1.0, want.abs().max().item())
                assert max_abs(got_n.cpu(), want) <= tol * scale, (width, P, M)
    rays, u1, u2 = _rays_and_u(200, 128, 17)
    z_base = torch.linspace(NEAR, FAR, 64).cuda()
    for P in (A.ops.F32, A.ops.BF16):
        a = A.ops.render_rays(prop.packed(P), mip.packed(P), P, dev(rays), z_base, dev(u1), dev(u2), 128, NEAR, FAR, True, want_depth=True, want_weights=True)
        b = A.ops.render_rays(prop.packed(P, wide=True), mip.packed(P, wide=True), P, dev(rays), z_base, dev(u1), dev(u2), 128, NEAR, FAR, True,
                              want_depth=True, want_weights=True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    with torch.no_grad():
        want_rgb, want_w, want_d = O.render_rays(psd, msd, rays, u1, u2, NEAR, FAR, 128, white_bkg=True)
        x_rgb, x_w, _ = O.render_rays({k: v.double() for k, v in psd.items()}, {k: v.double() for k, v in msd.items()}, rays.double(), u1.double(), u2.double(),
                                      NEAR, FAR, 128, white_bkg=True)
    tol_img = max(1e-4, 1.5 * max(max_abs(want_rgb, x_rgb), max_abs(want_w, x_w)))         # (O(1) activations: the fp32 reference's own noise floor)
    a = A.ops.render_rays(prop.packed(A.ops.F32), mip.packed(A.ops.F32), A.ops.F32, dev(rays), z_base, dev(u1), dev(u2), 128, NEAR, FAR, True, want_weights=True)
    assert max_abs(a[0].cpu(), want_rgb) <= tol_img and max_abs(a[2].cpu(), want_w) <= tol_img and tol_img <= 5e-4
    # the narrow layout has no integrated-PE kernel: the C-ABI refuses the combination, the module hands out its 256-wide blob for it
    z = torch.sort(torch.rand(200, 129, generator=gen) * 4 + 2, dim=-1)[0].cuda()
    dn = A.ops.dirs_norm(dev(rays))
    with pytest.raises(RuntimeError):
        A.ops.mip_forward_samples(mip.packed(A.ops.F32), A.ops.F32, A.ops.samples_rays(dev(rays), 128, z=z, ipe_radius=1e-3, ipe_dir_norm=dn), (200, 128), "cuda")
    with torch.no_grad():
        out = mip.forward_rays(dev(rays), z, 128, ipe_radius=1e-3)
    assert out.shape == (200, 128, 4) and bool(torch.isfinite(out).all())
    # ... and no fused-compositing kernel (`mip_forward_composite(mip.packed(P), ...)` with a narrow net walked the 352-fragment
    # blob with the 256-wide kernel): ops takes the two launches and equals the wide blob's fused launch, the C-ABI refuses the flag loudly
    import ctypes as C
    for P in (A.ops.F32, A.ops.BF16):
        for wb in (False, True):
            n3 = A.ops.mip_forward_composite(mip.packed(P), P, dev(rays), z, 128, wb, NEAR, FAR, want_depth=True, want_weights=True)
            w3 = A.ops.mip_forward_composite(mip.packed(P, wide=True), P, dev(rays), z, 128, wb, NEAR, FAR, want_depth=True, want_weights=True)
            assert max_abs(n3[0], w3[0]) <= 2e-6 and max_abs(n3[1], w3[1]) <= 2e-6 and max_abs(n3[2], w3[2]) <= 1e-6
    s = A.ops.samples_rays(dev(rays), 128, z=z)
    rgb = torch.empty(200, 3, device="cuda")
    rc = A.ops.lib.nerf_amd_mip_forward_composite(C.c_void_p(mip.packed(A.ops.BF16).data_ptr()), A.ops.BF16 | A.ops.FINE_W128, C.byref(s), 1, NEAR, FAR,
                                                  C.c_void_p(rgb.data_ptr()), None, None, None)
    assert rc != 0


def test_refnerf_render_with_scene_contraction(A):
    """Mip-NeRF 360 scene contraction on the Ref-NeRF render path (round 4: a flag of the sample fetch of nerf_amd_render_rays_ref too; it
    used to raise).  Not in the reference -- the build's own definition, oracle.render_rays_ref(contracted=True): every sample position,
    proposal and merged fine, is contracted before its encoding.  fp32 against the oracle on unbounded depths, the flag really changes
    the image, and the drop-in render_image accepts it for a RefNeRF."""
    prop, _ = build_nets(A, "small")
    net = build_ref(A, "small")
    near, far = 0.2, 30.0
    rays, u1, u2 = _rays_and_u(300, 64, 41)
    rays = rays.clone(); rays[:, :3] *= 0.3                                   # cameras inside the unit ball, samples far outside it
    z_base = torch.linspace(near, far, 64).cuda()
    cam = torch.tensor([0.0, 0.6, -0.8])
    with torch.no_grad():
        want_rgb, _, extras = O.render_rays_ref(W.proposal_state("small"), W.ref_state("small"), rays, u1, u2, near, far, 64, white_bkg=True, cam_z=cam,
                                                contracted=True)
    rgb, depth, nimg, _ = A.ops.render_rays_ref(prop.packed(A.ops.F32), net.packed(A.ops.F32), A.ops.F32, dev(rays), z_base, dev(u1), dev(u2), 64, near, far, True,
                                                want_depth=True, cam_dir=cam.cuda(), contract=True)
    plain, _, _, _ = A.ops.render_rays_ref(prop.packed(A.ops.F32), net.packed(A.ops.F32), A.ops.F32, dev(rays), z_base, dev(u1), dev(u2), 64, near, far, True)
    assert max_abs(rgb.cpu(), want_rgb) <= 1e-4 and max_abs(nimg.cpu(), extras["normal_img"]) <= 1e-4
    gate("Ref-NeRF contracted render: depth vs oracle", max_abs(depth.cpu(), extras["depth_img"]), 1e-4)           # (measured 1.1e-8)
    assert float((rgb - plain).abs().max()) > 5e-6                             # (reference-style weights: a small but real effect)
    A.pkg.set_precision("fp32")
    pose = A.utils.pose_spherical(20.0, -25.0, 0.4)[:3].cuda()
    with torch.no_grad():
        out = A.procedures.render_image(net, prop, pose, 50, 60.0, near, far, 64, white_bkg=True, contract=True)
    assert out["rgb"].shape == (3, 50, 50) and bool(torch.isfinite(out["rgb"]).all())


@pytest.mark.parametrize("L,cat,width", [(6, True, 256), (10, False, 256), (4, False, 96)])
def test_shallow_encodings_and_cat_origin(A, golden, L, cat, width):
    """Fewer encoding octaves / cat_origin=False (constructor arguments: mip_model.py:15-18, addtional.py:61, ref_model.py:17-24): the
    compiled kernels evaluate all three networks with the module's encoding columns placed inside the [x | 10 octaves] layout and zeros
    elsewhere (nerf_amd/_packed.py `_column_segments`, RefNeRF._embed_pos).  Forward values and the golden's gradient rows against the REAL
    reference (G21), every parameter gradient against fp64 autograd of the oracle, the density-gradient normals of the proposal network,
    and the bf16 path close to fp32."""
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.mip_model import MipNeRF
    from nerf_amd.ref_model import RefNeRF
    from test_oracle_golden import shallow_states
    g = golden("g21_shallow_encodings")
    tag = "L%d_%s_w%d" % (L, "cat" if cat else "nocat", width)
    msd, psd, rsd = shallow_states(L, cat, width)
    mip, prop = MipNeRF(L, 4, width, cat_origin=cat), ProposalNetwork(L, width, cat_origin=cat)
    ref = RefNeRF(L, 4, hidden_unit=width, output_dim=width, cat_origin=cat, perturb_bottle_neck_w=0.0)
    mip.load_state_dict(msd); prop.load_state_dict(psd); ref.load_state_dict(rsd)
    mip, prop, ref = mip.cuda(), prop.cuda(), ref.cuda()
    A.pkg.set_precision("fp32")
    pts = g["pts"].cuda()
    scale = lambda t: max(1.0, t.abs().max().item())
    with torch.no_grad():                                                  # eval: the narrow-tile kernels at width 96, the wide ones at 256
        y, d = mip.eval().forward(pts), prop.eval().forward(pts[..., :3].contiguous())
        rgbo, nrm = ref.eval().forward(pts)
    assert max_abs(y.cpu(), g[tag + "_mip"]) <= 1e-5 * scale(g[tag + "_mip"])
    assert max_abs(d.cpu(), g[tag + "_prop"]) <= 1e-5 * scale(g[tag + "_prop"])
    assert max_abs(rgbo.cpu(), g[tag + "_ref_rgbo"]) <= 2e-5 * scale(g[tag + "_ref_rgbo"]) and max_abs(nrm.cpu(), g[tag + "_ref_normal"]) <= 2e-5
    # training forward + HIP backward (the wide kernels on the placed tensors)
    mip.train(); prop.train(); ref.train()
    (mip.forward(pts) * g["G4"].cuda()).sum().backward()
    (prop.forward(pts[..., :3].contiguous()) * g["G1"].cuda()).sum().backward()
    r2, n2 = ref.forward(pts)
    ((r2 * g["G4"].cuda()).sum() + (n2 * g["G3"].cuda()).sum()).backward()
    d64 = lambda sd: {k: v.double().requires_grad_(True) for k, v in sd.items()}
    m64, p64, r64 = d64(msd), d64(psd), d64(rsd)
    p_ = g["pts"].double()
    (O.mip_forward(m64, p_, Lp=L, cat_origin=cat) * g["G4"].double()).sum().backward()
    (O.proposal_forward(p64, p_[..., :3], L=L, cat_origin=cat) * g["G1"].double()).sum().backward()
    a64, b64 = O.ref_forward(r64, p_, Lp=L, cat_origin=cat)
    ((a64 * g["G4"].double()).sum() + (b64 * g["G3"].double()).sum()).backward()
    for net, want in ((mip, m64), (prop, p64), (ref, r64)):
        for name, prm in net.named_parameters():
            wg = want[name].grad
            assert prm.grad is not None and tuple(prm.grad.shape) == tuple(wg.shape), name
            diff, top = prm.grad.cpu().double() - wg, max(wg.abs().max().item(), 1e-12)
            assert diff.norm().item() <= 1e-2 * max(wg.norm().item(), 1e-12) and diff.abs().max().item() <= 5e-2 * top, \
                "%s %s: |err|_2 %.3e of %.3e, max %.3e of %.3e" % (type(net).__name__, name, diff.norm().item(), wg.norm().item(), diff.abs().max().item(), top)
    for net, key, name in ((mip, "_mip_g0", "lin_block1.0.weight"), (mip, "_mip_gskip", "lin_block2.0.weight"), (mip, "_mip_grgb", "rgb_layer.0.weight"),
                           (prop, "_prop_g0", "layers.0.weight"), (ref, "_ref_g0", "spa_block1.0.weight"), (ref, "_ref_gskip", "spa_block2.0.weight")):
        got, want = dict(net.named_parameters())[name].grad[:8].cpu(), g[tag + key]      # the real reference's own fp32 gradient rows
        assert (got - want).norm().item() <= 1e-2 * max(want.norm().item(), 1e-12), (key, (got - want).norm().item(), want.norm().item())
    # density-gradient normals of the proposal network (train.py:165-168): the encoding's derivative incl. the raw-position column
    x = pts[..., :3].detach().clone().requires_grad_(True)
    normals = RefNeRF.get_grad(prop.forward(x), x)
    x64 = p_[..., :3].clone().requires_grad_(True)
    g64, = torch.autograd.grad(O.proposal_forward({k: v.detach() for k, v in p64.items()}, x64, L=L, cat_origin=cat).sum(), x64)
    want_n = g64 / torch.maximum(torch.full_like(g64[..., :1], 1e-5), g64.norm(dim=-1, keepdim=True))
    assert max_abs(normals.cpu(), want_n) <= 1e-3
    # bf16 (the default precision): same function up to the operand rounding
    A.pkg.set_precision("bf16")
    with torch.no_grad():
        yb, db = mip.eval().forward(pts), prop.eval().forward(pts[..., :3].contiguous())
    A.pkg.set_precision("fp32")
    assert max_abs(yb[..., :3].cpu(), g[tag + "_mip"][..., :3]) <= 0.05 and max_abs(db.cpu(), g[tag + "_prop"]) <= 0.05 * scale(g[tag + "_prop"])
    assert RefNeRF(11, 4)._generic() and not RefNeRF(L, 4, hidden_unit=width, output_dim=width, cat_origin=cat)._generic()


# ------------------------------------------------------------------------------------------------ shapes LARGER than the compiled ones
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_generic_gemm_stride_forms(A, prec):
    """nerf_amd_gemm (nerf_amd/csrc/generic_kernels.hip) in the three stride forms the layer-by-layer path uses -- forward x W^T (+ bias,
    activation), input gradient dy W with the ReLU mask, weight gradient dy^T x with the contraction over the samples split over
    workgroups -- at ragged sizes, against fp64 matmul of the same (bf16-rounded in bf16 mode) operands; the split sum is reproducible."""
    code = A.ops.F32 if prec == "fp32" else A.ops.BF16
    gen = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=gen)
    q = (lambda t: t) if prec == "fp32" else (lambda t: t.to(torch.bfloat16).float())
    for M, N, K in ((300, 77, 131), (129, 513, 64), (1, 1, 5), (2500, 320, 383)):
        x, w, b, dy = rnd(M, K), rnd(N, K) * 0.1, rnd(N), rnd(M, N)
        want = q(x).double() @ q(w).double().t() + b.double()
        tol = lambda ref: 2e-5 * max(1.0, ref.abs().max().item())
        for act, f in ((0, lambda t: t), (1, torch.relu), (2, torch.sigmoid)):
            got = A.ops.gemm(code, x.cuda(), w.cuda().t(), bias=b.cuda(), act=act)
            assert max_abs(got.cpu(), f(want)) <= tol(f(want)), (M, N, K, act)
        buf = torch.full((M, N + 5), -7.0).cuda()                                   # strided output: a column range of a wider buffer
        A.ops.gemm(code, x.cuda(), w.cuda().t(), out=buf[:, 3: 3 + N], bias=b.cuda(), act=1)
        assert max_abs(buf[:, 3: 3 + N].cpu(), torch.relu(want)) <= tol(want) and float(buf[:, :3].min()) == -7.0 and float(buf[:, 3 + N:].max()) == -7.0
        mask = torch.relu(rnd(M, K))
        dx = A.ops.gemm(code, dy.cuda(), w.cuda(), mask=mask.cuda())
        want_dx = (q(dy).double() @ q(w).double()) * (mask > 0)
        assert max_abs(dx.cpu(), want_dx) <= tol(want_dx), (M, N, K, "dx")
        dw = A.ops.gemm(code, dy.cuda().t(), x.cuda())
        want_dw = q(dy).double().t() @ q(x).double()
        assert max_abs(dw.cpu(), want_dw) <= 2e-5 * max(1.0, want_dw.abs().max().item()) * (4 if M > 1000 else 1), (M, N, K, "dw")
    # the weight-gradient shape: a contraction over 70 001 samples is split over workgroups; fixed summation order
    M, N, K = 70001, 96, 130
    dy, x = rnd(M, N), rnd(M, K)
    assert A.ops.lib.nerf_amd_gemm_workspace_bytes(N, K, M) > 0
    a1 = A.ops.gemm(code, dy.cuda().t(), x.cuda())
    a2 = A.ops.gemm(code, dy.cuda().t(), x.cuda())
    assert torch.equal(a1, a2)
    want = q(dy).double().t() @ q(x).double()
    assert max_abs(a1.cpu(), want) <= 1e-4 * want.abs().max().item()
    db = A.ops.gemm(code, dy.cuda().t(), torch.ones(M, 1).cuda())
    assert max_abs(db.cpu().reshape(-1), q(dy).double().sum(0)) <= 1e-4 * q(dy).double().sum(0).abs().max().item()
    xd = x.cuda()                                                                    # the C-ABI refuses an operand with no unit stride
    assert A.ops.lib.nerf_amd_gemm(code, 4, 4, 4, xd.data_ptr(), 2, 3, xd.data_ptr(), 1, 4, a1.data_ptr(), 4, None, 0, None, 0, None, None) != 0
    assert b"stride" in A.ops.lib.nerf_amd_last_error()


def test_layer_products_on_bf16_rows(A, golden):
    """Round 5 (ABI 124): the inference route of the networks outside the compiled shapes under bf16 precision -- nerf_amd_rows_gemm
    (nerf_amd/csrc/rows_gemm_kernels.hip: 256 x 256 tiles, both operands by LDS DMA) on activations kept as bf16 rows.  (i) the product
    against fp64 on the bf16-rounded operands -- ragged sizes, the heads (N = 1, 3, 11 of a 256-wide tile), column-range views with
    finite junk around them, every activation, both output types; (ii) what the C-ABI refuses; (iii) MipNeRF / ProposalNetwork / RefNeRF
    (`--ide_level 5`) forwards on this route against the fp32-row route (which rounds the same operands to bf16 on their way into LDS:
    the routes differ by accumulation order in the concatenated layers only) and against the fp32 oracle."""
    from nerf_amd import generic_path
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.mip_model import MipNeRF
    from nerf_amd.ref_model import RefNeRF
    from test_oracle_golden import generic_ref_state
    ops = A.ops
    for M, N, K, act, dt, c0 in ((1000, 512, 512, 1, torch.bfloat16, 0), (257, 320, 63, 1, torch.bfloat16, 0), (4099, 320, 383, 1, torch.bfloat16, 320),
                                 (777, 1, 320, 0, torch.float32, 0), (777, 3, 320, 2, torch.float32, 0), (513, 11, 256, 0, torch.float32, 0),
                                 (300, 256, 201, 1, torch.bfloat16, 256), (5, 128, 32, 0, torch.bfloat16, 0), (1, 260, 9, 2, torch.float32, 8)):
        g = torch.Generator().manual_seed(M + N + K)
        x32, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
        buf = torch.full((M, c0 + ops._pad(K, 8) + 8), 7.0, dtype=torch.bfloat16, device="cuda")
        ops.rows_to_bf16(x32.cuda(), buf, c0)
        assert torch.equal(buf[:, c0:c0 + K].cpu(), x32.bfloat16()) and float(buf[:, c0 + K:c0 + ops._pad(K, 8)].abs().sum()) == 0.0      # RNE, zero padding
        layer = ops.PackedLinear(w.cuda(), b.cuda())
        got = ops.rows_gemm(buf[:, c0:c0 + K], layer, act, out_dtype=dt)
        assert got.shape == (M, N) and got.dtype == dt
        want = x32.bfloat16().double() @ w.bfloat16().double().t() + b.double()
        want = want.clamp(min=0) if act == 1 else (torch.sigmoid(want) if act == 2 else want)
        err = float(((got.float().cpu().double() - want).abs() / (1.0 + want.abs())).max())
        gate("rows_gemm %d x %d x %d act %d -> %s: vs fp64 on the rounded operands" % (M, N, K, act, str(dt).split(".")[1]), err, 1e-5 if dt == torch.float32 else 4e-3)
    assert ops.rows_gemm(torch.empty((0, 64), dtype=torch.bfloat16, device="cuda"), layer.__class__(torch.zeros(8, 64).cuda(), None)).shape == (0, 8)
    # (ii) refusals: an X off its 16-byte alignment, a row stride that is not a multiple of 8, a weight that is not packed, a ragged bf16 output
    xb = torch.zeros((64, 72), dtype=torch.bfloat16, device="cuda")
    lay = ops.PackedLinear(torch.zeros(8, 64).cuda(), torch.zeros(8).cuda())
    out = torch.zeros((64, 8), dtype=torch.bfloat16, device="cuda")
    call = lambda x_ptr, ldx, ldw, n_pad, N, ldc: ops.lib.nerf_amd_rows_gemm(64, N, 64, x_ptr, ldx, lay.weight.data_ptr(), ldw, n_pad, lay.bias.data_ptr(), 0,
                                                                            out.data_ptr(), ldc, 1, None)
    assert call(xb.data_ptr(), 72, 64, 256, 8, 8) == 0
    for bad in ((xb.data_ptr() + 2, 72, 64, 256, 8, 8), (xb.data_ptr(), 70, 64, 256, 8, 8), (xb.data_ptr(), 72, 48, 256, 8, 8), (xb.data_ptr(), 72, 64, 8, 8, 8),
                (xb.data_ptr(), 72, 64, 256, 6, 8)):
        assert call(*bad) != 0 and b"nerf_amd_rows_gemm" in ops.lib.nerf_amd_last_error(), bad
    with pytest.raises(RuntimeError):
        ops.rows_gemm(xb[:, :60], lay)                                                  # 60 columns against a 64-column layer
    # (iii) the networks
    sc = lambda t: max(1.0, t.abs().max().item())
    msd = O.init_linear_params(O.mip_shapes(10, 4, 320, True), 41, std=0.06, bias_std=0.05)
    psd = O.init_linear_params(O.proposal_shapes(10, 512, True), 42, std=0.06, bias_std=0.05)
    mip, prop = MipNeRF(10, 4, 320), ProposalNetwork(10, 512)
    mip.load_state_dict(msd); prop.load_state_dict(psd)
    mip, prop = mip.cuda().eval(), prop.cuda().eval()
    gen = torch.Generator().manual_seed(5)
    pts = torch.cat((torch.rand(301, 3, 3, generator=gen) * 3 - 1.5, torch.randn(301, 3, 3, generator=gen)), dim=-1)
    rsd = generic_ref_state(10, 5, 256)
    ref = RefNeRF(10, 5, hidden_unit=256, output_dim=256, perturb_bottle_neck_w=0.0)
    ref.load_state_dict(rsd)
    ref = ref.cuda().eval()
    assert mip._generic() and prop._generic() and ref._generic()
    gd = golden("g22_generic_refnerf")
    A.pkg.set_precision("bf16")
    try:
        res = {}
        for rows in (True, False):
            generic_path.ROWS_ROUTE = rows
            with torch.no_grad():
                res[rows] = (mip.forward(pts.cuda()), prop.forward(pts[..., :3].contiguous().cuda()), *ref.forward(gd["pos"].cuda(), gd["dirs"].cuda()))
        assert "_rows_packed" in mip.__dict__ and "_rows_packed" in ref.__dict__                    # the route ran (and cached its packed weights)
        with torch.no_grad():                                                                        # a parameter update re-packs
            before = mip.forward(pts.cuda())
            mip.opacity_head[0].bias.add_(1.0)
            after = mip.forward(pts.cuda())
            mip.opacity_head[0].bias.sub_(1.0)
        gate("bf16 rows: opacity after bias += 1 (re-pack on a parameter update)", max_abs(after[..., 3] - 1.0, before[..., 3]), 1e-5)
        assert torch.equal(after[..., :3], before[..., :3])
        with torch.no_grad():
            again = mip.forward(pts.cuda())                                                          # ((b + 1) - 1 is not b in fp32: the eval value after the round trip)
        mip.train()                                                                                  # train mode: packed per call, never cached (PackedWeightsMixin's rule:
        with torch.no_grad():                                                                        # a graph-replayed step moves no version counter)
            tr = mip.forward(pts.cuda())
        assert not mip.__dict__["_rows_packed"] and torch.equal(tr, again)
        mip.eval()
    finally:
        generic_path.ROWS_ROUTE = True
        A.pkg.set_precision("fp32")
    names = ("MipNeRF (N,S,4)", "proposal density", "RefNeRF rgbo", "RefNeRF normal")
    with torch.no_grad():
        want = (O.mip_forward(msd, pts, Lp=10, cat_origin=True), O.proposal_forward(psd, pts[..., :3], L=10, cat_origin=True),
                gd["L10_d5_w256_rgbo"], gd["L10_d5_w256_normal"])
    for name, a, b, w in zip(names, res[True], res[False], want):
        assert a.shape == b.shape == w.shape, name
        gate("bf16 rows route vs fp32-row route, %s" % name, max_abs(a.cpu(), b.cpu()) / sc(w), 1e-2)
        gate("bf16 rows route vs the fp32 oracle, %s" % name, max_abs(a.cpu(), w) / sc(w), 0.06)


@pytest.mark.parametrize("L,cat,w_mip,w_prop", [(10, True, 320, 512), (12, True, 128, 64), (11, False, 288, 300)])
def test_networks_larger_than_the_compiled_shapes(A, L, cat, w_mip, w_prop):
    """`--nerf_net_width` / `--prop_net_width` above 256 (procedures.py:176-177) and more than 10 position octaves (mip_model.py:15-18,
    addtional.py:61): the networks run layer by layer on the generic MFMA GEMM (nerf_amd/generic_path.py).  Forward against the oracle,
    every parameter gradient against fp64 autograd of the oracle, bf16 close to fp32, and no position gradient on offer."""
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.mip_model import MipNeRF
    msd = O.init_linear_params(O.mip_shapes(L, 4, w_mip, cat), 900 + w_mip, std=0.06, bias_std=0.05)
    psd = O.init_linear_params(O.proposal_shapes(L, w_prop, cat), 901 + w_prop, std=0.06, bias_std=0.05)
    mip, prop = MipNeRF(L, 4, w_mip, cat_origin=cat), ProposalNetwork(L, w_prop, cat_origin=cat)
    mip.load_state_dict(msd); prop.load_state_dict(psd)
    assert mip._generic() and (prop._generic() or (w_prop <= 256 and L <= 10))
    mip, prop = mip.cuda(), prop.cuda()
    A.pkg.set_precision("fp32")
    gen = torch.Generator().manual_seed(L * 100 + w_mip)
    pts = torch.cat((torch.rand(37, 11, 3, generator=gen) * 3 - 1.5, torch.randn(37, 11, 3, generator=gen)), dim=-1)
    G4, G1 = torch.randn(37, 11, 4, generator=gen), torch.randn(37, 11, generator=gen)
    scale = lambda t: max(1.0, t.abs().max().item())
    with torch.no_grad():
        want_y, want_d = O.mip_forward(msd, pts, Lp=L, cat_origin=cat), O.proposal_forward(psd, pts[..., :3], L=L, cat_origin=cat)
        y, d = mip.eval().forward(pts.cuda()), prop.eval().forward(pts[..., :3].contiguous().cuda())
    # (octaves 10, 11 multiply the position by 1024 / 2048 before sin / cos: the fp32 reference's own rounding of that argument is ~1e-4)
    tol = 1e-5 if L <= 10 else 5e-4
    assert y.shape == (37, 11, 4) and max_abs(y.cpu(), want_y) <= tol * scale(want_y), max_abs(y.cpu(), want_y)
    assert d.shape == (37, 11) and max_abs(d.cpu(), want_d) <= tol * scale(want_d), max_abs(d.cpu(), want_d)
    mip.train(); prop.train()
    (mip.forward(pts.cuda()) * G4.cuda()).sum().backward()
    (prop.forward(pts[..., :3].contiguous().cuda()) * G1.cuda()).sum().backward()
    d64 = lambda sd: {k: v.double().requires_grad_(True) for k, v in sd.items()}
    m64, p64 = d64(msd), d64(psd)
    (O.mip_forward(m64, pts.double(), Lp=L, cat_origin=cat) * G4.double()).sum().backward()
    (O.proposal_forward(p64, pts[..., :3].double(), L=L, cat_origin=cat) * G1.double()).sum().backward()
    for net, want in ((mip, m64), (prop, p64)):
        for name, prm in net.named_parameters():
            wg = want[name].grad
            assert prm.grad is not None and tuple(prm.grad.shape) == tuple(wg.shape), name
            diff, top = prm.grad.cpu().double() - wg, max(wg.abs().max().item(), 1e-12)
            assert diff.norm().item() <= 1e-2 * max(wg.norm().item(), 1e-12) and diff.abs().max().item() <= 5e-2 * top, \
                "%s %s: |err|_2 %.3e of %.3e, max %.3e of %.3e" % (type(net).__name__, name, diff.norm().item(), wg.norm().item(), diff.abs().max().item(), top)
    A.pkg.set_precision("bf16")
    with torch.no_grad():
        yb = mip.eval().forward(pts.cuda())
    A.pkg.set_precision("fp32")
    assert max_abs(yb[..., :3].cpu(), want_y[..., :3]) <= 0.06
    with pytest.raises(NotImplementedError):                                          # no position gradient on the generic path: refused at forward time
        mip.train().forward(pts.cuda().requires_grad_(True))
    with torch.no_grad():                                                             # scene contraction: a stage in front of the encoder on this path
        yc = mip.eval().forward(pts.cuda() * 4.0, contract=True)                     # (values: test_scene_contraction_on_the_layer_by_layer_route)
    assert yc.shape == want_y.shape and bool(torch.isfinite(yc).all())


@pytest.mark.parametrize("L,deg,width,srgb", [(10, 5, 256, False), (10, 4, 320, False), (11, 3, 288, True)])
def test_refnerf_outside_the_compiled_shapes(A, golden, L, deg, width, srgb):
    """`-t --ide_level 5` (procedures.py:211: 36 spherical-harmonic terms), `-t --nerf_net_width 320` (train.py:80), 11 position octaves with
    use_srgb: a RefNeRF the fused kernel is not compiled for runs layer by layer on the generic MFMA GEMM with the element-wise stages of
    generic_ref_kernels.hip between the products (nerf_amd/generic_path.py `ref_forward`).  Against the REAL reference (golden G22): forward
    values, predicted normals, RefNeRF.get_grad of the density, the golden's gradient rows; every parameter gradient against fp64 autograd of the
    oracle; the train-mode bottle-neck noise; bf16 close to fp32."""
    from nerf_amd.ref_model import RefNeRF
    from test_oracle_golden import generic_ref_state
    g = golden("g22_generic_refnerf")
    tag = "L%d_d%d_w%d" % (L, deg, width)
    sd = generic_ref_state(L, deg, width)
    net = RefNeRF(L, deg, hidden_unit=width, output_dim=width, use_srgb=srgb, perturb_bottle_neck_w=0.0)
    net.load_state_dict(sd)
    assert net._generic()
    net = net.cuda()
    A.pkg.set_precision("fp32")
    pos, dirs = g["pos"].cuda(), g["dirs"].cuda()
    sc = lambda t: max(1.0, t.abs().max().item())
    tol = 2e-5 if L <= 10 else 5e-4          # (octave 10 multiplies the position by 1024 before sin / cos: fp32 argument rounding)
    with torch.no_grad():
        rgbo, nrm = net.eval().forward(pos, dirs)
        rgbo6, nrm6 = net.forward(torch.cat((pos, dirs), dim=-1))                         # the (N,S,6) form of the call
    assert rgbo.shape == (5, 7, 4) and nrm.shape == (5, 7, 3) and torch.equal(rgbo, rgbo6) and torch.equal(nrm, nrm6)
    assert max_abs(rgbo.cpu(), g[tag + "_rgbo"]) <= tol * sc(g[tag + "_rgbo"]), max_abs(rgbo.cpu(), g[tag + "_rgbo"])
    assert max_abs(nrm.cpu(), g[tag + "_normal"]) <= tol
    # training: get_grad (d density / d position, first order, ref_model.py:119-125), then the loss backward over the same graph (train.py:179-186)
    net.train()
    p = pos.clone().requires_grad_(True)
    r2, n2 = net.forward(p, dirs)
    normals = RefNeRF.get_grad(r2[..., -1], p)
    assert max_abs(normals.cpu(), g[tag + "_density_grad"]) <= 2e-3, max_abs(normals.cpu(), g[tag + "_density_grad"])
    ((r2 * g["G4"].cuda()).sum() + (n2 * g["G3"].cuda()).sum()).backward()
    assert p.grad is None                                                              # (positions get a gradient inside get_grad only, like on the fused path)
    s64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    a64, b64 = O.ref_forward(s64, g["pos"].double(), g["dirs"].double(), Lp=L, deg=deg, use_srgb=srgb)
    ((a64 * g["G4"].double()).sum() + (b64 * g["G3"].double()).sum()).backward()
    for name, prm in net.named_parameters():
        wg = s64[name].grad
        assert prm.grad is not None and tuple(prm.grad.shape) == tuple(wg.shape), name
        diff, top = prm.grad.cpu().double() - wg, max(wg.abs().max().item(), 1e-12)
        assert diff.norm().item() <= 1e-2 * max(wg.norm().item(), 1e-12) and diff.abs().max().item() <= 5e-2 * top, \
            "%s: |err|_2 %.3e of %.3e, max %.3e of %.3e" % (name, diff.norm().item(), wg.norm().item(), diff.abs().max().item(), top)
    grads = dict(net.named_parameters())
    for key, name in (("_g_spa0", "spa_block1.0.weight"), ("_g_dir0", "dir_block1.0.weight"), ("_g_dirskip", "dir_block2.0.weight"),
                      ("_g_heads", "norm_col_tint_head.weight"), ("_g_rho_tau", "rho_tau_head.weight"), ("_g_bottle", "bottle_neck.weight")):
        want = g[tag + key]                                                              # the real reference's own fp32 gradient rows
        got = grads[name].grad[: want.shape[0]].cpu()
        assert (got - want).norm().item() <= 1e-2 * max(want.norm().item(), 1e-12), (key, (got - want).norm().item(), want.norm().item())
    # train-mode perturbation of the bottle-neck (ref_model.py:84-85): the torch.normal draw of the forward, reproduced for the oracle
    noisy = RefNeRF(L, deg, hidden_unit=width, output_dim=width, use_srgb=srgb, perturb_bottle_neck_w=0.1)
    noisy.load_state_dict(sd)
    noisy = noisy.cuda().train()
    noisy.noise_rng = "torch"
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    with torch.no_grad():
        rn, _ = noisy.forward(pos, dirs)
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    noise = torch.normal(0, 0.1, (5, 7, 128), device="cuda").cpu()
    with torch.no_grad():
        want_n, _ = O.ref_forward(sd, g["pos"], g["dirs"], Lp=L, deg=deg, use_srgb=srgb, noise=noise)
    assert max_abs(rn.cpu(), want_n) <= 10 * tol * sc(want_n) and max_abs(rn.cpu(), g[tag + "_rgbo"]) > 1e-4
    A.pkg.set_precision("bf16")
    with torch.no_grad():
        rb, nb = net.eval().forward(pos, dirs)
    A.pkg.set_precision("fp32")
    assert max_abs(rb[..., :3].cpu(), g[tag + "_rgbo"][..., :3]) <= 0.1 and max_abs(nb.cpu(), g[tag + "_normal"]) <= 0.1, (max_abs(rb[..., :3].cpu(), g[tag + "_rgbo"][..., :3]), max_abs(nb.cpu(), g[tag + "_normal"]))
    # density-gradient normals of a generic-shape proposal network (`--prop_normal --prop_net_width 320`, train.py:165-168)
    from nerf_amd.addtional import ProposalNetwork
    psd = O.init_linear_params(O.proposal_shapes(L, 320, True), 71, std=0.06, bias_std=0.05)
    prop = ProposalNetwork(L, 320)
    prop.load_state_dict(psd)
    prop = prop.cuda().train()
    x = pos.clone().requires_grad_(True)
    pn = RefNeRF.get_grad(prop.forward(x), x)
    x64 = g["pos"].double().clone().requires_grad_(True)
    g64, = torch.autograd.grad(O.proposal_forward({k: v.double() for k, v in psd.items()}, x64, L=L).sum(), x64)
    want_pn = g64 / torch.maximum(torch.full_like(g64[..., :1], 1e-5), g64.norm(dim=-1, keepdim=True))
    assert max_abs(pn.cpu(), want_pn) <= 2e-3


def test_refnerf_train_step_outside_the_compiled_shapes(A):
    """The reference's whole Ref-NeRF training iteration (train.py:164-199 with `-t --ide_level 5 --prop_normal --prop_net_width 320`) on the
    generic path: both networks layer by layer on nerf_amd_gemm, RefNeRF.get_grad twice (proposal density and fine density w.r.t. their
    positions), the normal / back-face / coarse-normal losses, backward to every parameter -- against the oracle's restatement of the step
    evaluated in fp64 (its pieces are pinned by G17 at ide_level 4 and by G22 at ide_level 5)."""
    from nerf_amd.addtional import ProposalLoss, ProposalNetwork, getBounds
    from nerf_amd.mip_methods import maxBlurFilter
    from nerf_amd.nerf_base import NeRF
    from nerf_amd.ref_model import BackFaceLoss, RefNeRF, WeightedNormalLoss
    from nerf_amd.utils import inverseSample
    A.pkg.set_precision("fp32")
    N, C, Fn = 24, 16, 32
    gen = torch.Generator().manual_seed(55)
    psd = O.init_linear_params(O.proposal_shapes(10, 320, True), 551, std=0.06, bias_std=0.05)
    rsd = O.init_linear_params(O.ref_shapes(10, 5, 256, 128, 256, True), 552, std=0.06, bias_std=0.05)
    prop, net = ProposalNetwork(10, 320), RefNeRF(10, 5)
    prop.load_state_dict(psd); net.load_state_dict(rsd)
    assert prop._generic() and net._generic()
    prop, net = prop.cuda().train(), net.cuda().train()
    net.noise_rng = "torch"
    dirs = F.normalize(torch.randn(N, 3, generator=gen) * 0.3 + torch.tensor([0.0, 0.0, -1.0]), dim=-1) * (0.8 + 0.4 * torch.rand(N, 1, generator=gen))
    rays_c = torch.cat((torch.tensor([0.0, 0.0, 4.0]).expand(N, 3) + 0.1 * torch.randn(N, 3, generator=gen), dirs), dim=-1).contiguous()
    zc_c = (torch.linspace(NEAR, FAR - (FAR - NEAR) / C, C) + torch.rand(N, C, generator=gen) * (FAR - NEAR) / C).contiguous()
    u_c, tgt_c = torch.rand(N, Fn + 1, generator=gen), torch.rand(N, 3, generator=gen)
    noise_c = 0.1 * torch.randn(N, C + Fn, 128, generator=gen)
    rays, zc, tgt, noise = rays_c.cuda(), zc_c.cuda(), tgt_c.cuda(), noise_c.cuda()
    real_normal = torch.normal
    torch.normal = lambda *a, **k: noise
    try:
        pts = (rays[:, None, :3] + rays[:, None, 3:] * zc[:, :, None]).contiguous().requires_grad_(True)
        dens = prop.forward(pts)
        coarse_grad = -RefNeRF.get_grad(dens, pts)
        pw = maxBlurFilter(ProposalNetwork.get_weights(F.softplus(dens), zc, rays[:, 3:]), 0.01)
        fl, below = inverseSample(pw, zc, Fn + 1, sort=True, u=u_c)
        samples, fl, below, sort_ids = NeRF.coarseFineMerge(rays, zc, fl, below)
        pos, d = samples.split((3, 3), dim=-1)
        pos = pos.contiguous().requires_grad_(True)
        rgbo, nrm = net.forward(pos, d.contiguous())
        dgrad = -RefNeRF.get_grad(rgbo[..., -1], pos)
        rgbo[..., -1] = F.softplus(rgbo[..., -1] + 0.5)
        rend, wts, _ = NeRF.render(rgbo, fl, rays[:, 3:], net.density_act)        # the reference's positional quirk (train.py:182)
        nl = WeightedNormalLoss()(wts, dgrad, nrm)
        bf = BackFaceLoss()(wts, nrm, d)
        cnl = WeightedNormalLoss()(pw, RefNeRF.coarse_grad_select(dgrad, sort_ids, C).detach(), coarse_grad)
        img = torch.mean((rend - tgt) ** 2)
        pl = ProposalLoss()(getBounds(pw, below), wts.detach())
        loss = pl + img + 4e-4 * (nl + 0.1 * cnl) + 0.1 * bf
        loss.backward()
    finally:
        torch.normal = real_normal
    d64 = lambda sd: {k: v.double().requires_grad_(True) for k, v in sd.items()}
    p64, r64 = d64(psd), d64(rsd)
    o = O.ref_train_step(p64, r64, rays_c.double(), zc_c.double(), u_c.double(), noise_c.double(), tgt_c.double(), Fn, Lp=10, deg=5)
    o["loss"].backward()
    assert torch.equal(sort_ids.cpu(), o["sort_ids"]) and torch.equal(below.cpu(), o["below_merged"])     # (the fp32 and fp64 sorts agree on this seed)
    # get_grad NORMALISES the gradient (ref_model.py:124-125): where the raw density gradient is small the unit vector is ill-conditioned and
    # fp32 against fp64 moves it by more than the rounding of a well-conditioned sample -- so: the typical sample to 1e-4, at most 2 % of them
    # beyond 2e-3 (the weighted normal losses below, which is what the vectors are for, are gated to 5e-4 relative)
    for got, want in ((dgrad, o["density_grad"]), (coarse_grad, o["coarse_grad"])):
        err = (got.cpu().double() - want).abs().amax(dim=-1).flatten()
        assert err.median().item() <= 1e-4 and (err > 2e-3).double().mean().item() <= 0.02, (err.median().item(), (err > 2e-3).double().mean().item(), err.max().item())
    assert max_abs(wts.detach().cpu().double(), o["weights"]) <= 5e-5 and max_abs(rend.detach().cpu().double(), o["rendered"]) <= 5e-5
    for name, val in (("normal_loss", nl), ("bf_loss", bf), ("coarse_normal_loss", cnl), ("img_loss", img), ("prop_loss", pl), ("loss", loss)):
        assert abs(val.item() - float(o[name])) <= 5e-4 * max(1.0, abs(float(o[name]))), (name, val.item(), float(o[name]))
    for mod, want in ((net, r64), (prop, p64)):
        for name, prm in mod.named_parameters():
            wg = want[name].grad
            assert prm.grad is not None and tuple(prm.grad.shape) == tuple(wg.shape), name
            diff, top = prm.grad.cpu().double() - wg, max(wg.abs().max().item(), 1e-14)
            assert diff.norm().item() <= 3e-2 * max(wg.norm().item(), 1e-14) and diff.abs().max().item() <= 0.1 * top, \
                "%s %s: |err|_2 %.3e of %.3e, max %.3e of %.3e" % (type(mod).__name__, name, diff.norm().item(), wg.norm().item(), diff.abs().max().item(), top)


def test_render_image_with_an_ide_level_5_refnerf(A):
    """render_image (procedures.py:34-97) with `-t --ide_level 5`: the Ref-NeRF tile body (coarse / fine merge, forward, softplus(sigma + 0.5),
    compositing with the normal image) call by call on the mirrored ops, the network on the generic path -- same seed, same images as the
    oracle's restatement of the reference's loop, 1e-4 abs."""
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.ref_model import RefNeRF
    rsd = O.init_linear_params(O.ref_shapes(10, 5, 256, 128, 256, True), 78, std=0.05, bias_std=0.02)
    psd = W.proposal_state("small")
    ref, prop = RefNeRF(10, 5), ProposalNetwork(10, 256)
    ref.load_state_dict(rsd); prop.load_state_dict(psd)
    ref, prop = ref.cuda().eval(), prop.cuda().eval()
    A.pkg.set_precision("fp32")
    pose = O.pose_spherical(37.0, -30.0, 4.0)[:3]
    focal = O.fov2focal(0.6911112070083618, (50, 50))
    torch.manual_seed(98)                                                                # a 50 x 50 image is ONE tile (procedures.py:20-32): the reference's two draws
    u1, u2 = torch.rand((50, 50, 64)).view(-1, 64), torch.rand((2500, 33))
    rays = torch.cat((pose[:, -1].expand(2500, -1), O.ray_dirs_image(pose, 50, 50, focal).reshape(-1, 3)), dim=-1)
    with torch.no_grad():
        want_rgb, _, ex = O.render_rays_ref(psd, rsd, rays, u1, u2, NEAR, FAR, 32, white_bkg=True, cam_z=pose[:, -2], deg=5)
    torch.manual_seed(98)
    with torch.no_grad():
        res = A.procedures.render_image(ref, prop, pose.cuda(), 50, focal, NEAR, FAR, 32, white_bkg=True, render_depth=True, render_normal=True, rng="reference")
    assert list(res.keys()) == ["rgb", "depth_img", "normal_img"] and res["rgb"].shape == (3, 50, 50)
    img = lambda t, ch: t.view(50, 50, ch).permute(2, 0, 1)
    assert max_abs(res["rgb"].cpu(), img(want_rgb, 3)) <= 1e-4
    assert max_abs(res["depth_img"][0].cpu(), img(ex["depth_img"].reshape(-1, 1), 1)[0]) <= 1e-4
    assert max_abs(res["normal_img"][0].cpu(), img(ex["normal_img"].reshape(-1, 1), 1)[0]) <= 1e-4
    with torch.no_grad():                                                                # default rng on this route: device-generator uniforms per chunk
        res2 = A.procedures.render_image(ref, prop, pose.cuda(), 50, focal, NEAR, FAR, 32, white_bkg=True)
    assert torch.isfinite(res2["rgb"]).all() and abs(float(res2["rgb"].mean()) - float(want_rgb.mean())) <= 0.05


def test_render_image_with_a_wide_network(A):
    """render_image (procedures.py:34-97) with `--nerf_net_width 320`: the fused render entry has no packed layout for it, so the tile body
    runs call by call on the mirrored ops with the fine network on the generic GEMM path -- same seed, same image as the oracle's
    restatement of the reference's loop (its tile order and CPU RNG draw order), 1e-4 abs."""
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.mip_model import MipNeRF
    msd = O.init_linear_params(O.mip_shapes(10, 4, 320), 77, std=0.05, bias_std=0.02)
    psd = W.proposal_state("small")
    mip, prop = MipNeRF(10, 4, 320), ProposalNetwork(10, 256)
    mip.load_state_dict(msd); prop.load_state_dict(psd)
    mip, prop = mip.cuda().eval(), prop.cuda().eval()
    A.pkg.set_precision("fp32")
    pose = O.pose_spherical(37.0, -30.0, 4.0)[:3]
    focal = O.fov2focal(0.6911112070083618, (50, 50))
    torch.manual_seed(99)
    with torch.no_grad():
        want = O.render_image(psd, msd, pose, 50, focal, NEAR, FAR, 32, white_bkg=True, render_depth=True)
    torch.manual_seed(99)
    with torch.no_grad():
        res = A.procedures.render_image(mip, prop, pose.cuda(), 50, focal, NEAR, FAR, 32, white_bkg=True, render_depth=True, rng="reference")
    assert res["rgb"].shape == (3, 50, 50)
    assert max_abs(res["rgb"].cpu(), want["rgb"]) <= 1e-4 and max_abs(res["depth_img"][0].cpu(), want["depth_img"][0]) <= 1e-4
    with torch.no_grad():                                                            # default rng on this route: device-generator uniforms per chunk
        res2 = A.procedures.render_image(mip, prop, pose.cuda(), 50, focal, NEAR, FAR, 32, white_bkg=True)
    assert torch.isfinite(res2["rgb"]).all() and float(res2["rgb"].min()) >= 0.0 and float(res2["rgb"].max()) <= 1.0 + 1e-5
    assert abs(float(res2["rgb"].mean()) - float(want["rgb"].mean())) <= 0.05        # (other uniforms: the same image up to sampling noise)
