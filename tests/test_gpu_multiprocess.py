"""SURVEY.md section 8e on real hardware: two processes (one torch.distributed rank each, gloo rendezvous, both on the box's single
GPU) render the two halves of an image through the HIP path and gather it; the result must equal the same shards rendered in one
process -- since round 5 the single-process `render_image(seed=...)` itself, bit for bit (in-kernel Philox keyed by the global ray index).
Also the flat-buffer gradient all-reduce on device tensors."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, SAMPLES, NEAR, FAR = 100, 128, 2.0, 6.0          # 100 = 2 x 2 tiles of 50: the tile-ordered ray list is exercised


def _nets():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import weights as W
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.mip_model import MipNeRF
    prop, mip = ProposalNetwork(10, 256), MipNeRF(10, 4, 256)
    prop.load_state_dict(W.proposal_state("small"))
    mip.load_state_dict(W.mip_state("small"))
    return prop.cuda().eval(), mip.cuda().eval()


def _pose_focal():
    from nerf_amd.utils import fov2Focal, pose_spherical
    return pose_spherical(30.0, -30.0, 4.0)[:3].cuda(), fov2Focal(0.6911112070083618, (H, H))


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import nerf_amd
    from nerf_amd import parallel
    nerf_amd.set_precision("fp32")
    prop, mip = _nets()
    pose, focal = _pose_focal()
    with torch.no_grad():
        img = parallel.render_image_sharded(mip, prop, pose, H, focal, NEAR, FAR, SAMPLES, white_bkg=True, render_depth=True, seed=5)
    # one "training step" worth of gradients: rank-dependent values, reduced into the mean
    for k, p in enumerate(list(mip.parameters()) + list(prop.parameters())):
        p.grad = torch.full_like(p, float(rank + 1) * (k + 1))
    n = parallel.allreduce_gradients([mip, prop])
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (k + 1))) for k, p in enumerate(list(mip.parameters()) + list(prop.parameters())))
    # a 3-way split of the same image inside this rank (sub-groups of one: gather=False slices stitched by hand) is the same image
    parts = []
    with torch.no_grad():
        from nerf_amd.procedures import render_image
        for r in range(3):
            s0, e0 = parallel.shard_range(H * H, r, 3, align=256)
            parts.append(render_image(mip, prop, pose, H, focal, NEAR, FAR, SAMPLES, white_bkg=True, render_depth=True, seed=5, _shard=(s0, e0)))
    rgb3 = parts[0]["to_image"](torch.cat([q["rgb_rays"] for q in parts]), 3)
    if rank == 0:
        torch.save({"rgb": img["rgb"].cpu(), "depth": img["depth_img"].cpu(), "n_reduced": n, "grads_ok": ok, "world3_equal": bool(torch.equal(rgb3, img["rgb"]))}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_render_and_gradient_allreduce(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    out_path = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, 29533, out_path), nprocs=2, join=True)
    got = torch.load(out_path)
    assert got["grads_ok"] and got["n_reduced"] == 530052 + 214017                     # fine + proposal parameters
    # ... and the image: the SAME as the single-process render_image with that Philox key, bit for bit (N-independent by construction:
    # uniforms keyed by the global ray index; round 4's per-rank torch.rand draws made the image depend on the world size)
    sys.path.insert(0, ROOT)
    import nerf_amd
    from nerf_amd.procedures import render_image
    nerf_amd.set_precision("fp32")
    prop, mip = _nets()
    pose, focal = _pose_focal()
    with torch.no_grad():
        one = render_image(mip, prop, pose, H, focal, NEAR, FAR, SAMPLES, white_bkg=True, render_depth=True, seed=5)
        other = render_image(mip, prop, pose, H, focal, NEAR, FAR, SAMPLES, white_bkg=True, render_depth=True, seed=6)
    assert torch.equal(got["rgb"], one["rgb"].cpu()) and torch.equal(got["depth"], one["depth_img"].cpu())
    assert not torch.equal(one["rgb"], other["rgb"])
    assert got["world3_equal"]


@pytest.mark.parametrize("wide", [False, True])
def test_three_way_shards_with_integrated_pe_equal_the_whole_image(wide):
    """with `ipe=` the direction norm of mip_methods.py:31 is ONE norm over the rays of the call.  A shard used to
    take the norm of its own rays only (i_m_ddt = 1 - dd / norm off by ~sqrt(world)), so the gathered image depended on the world size.
    The norm is now taken over the whole tile-ordered ray list before a shard is cut: three shards stitched together ARE the
    single-process image, bit for bit, on the fused route and (`wide`: hidden width 320 -> layer by layer) on the generic one; a shard
    rendered with its OWN norm differs (the regression this guards)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    import nerf_amd
    from nerf_amd import ops, parallel
    from nerf_amd.procedures import render_image
    nerf_amd.set_precision("fp32")
    if wide:
        import weights as W
        from nerf_amd.addtional import ProposalNetwork
        from nerf_amd.mip_model import MipNeRF
        prop, mip = ProposalNetwork(10, 320), MipNeRF(10, 4, 320)
        prop.load_state_dict(W.proposal_state("small", hidden=320))
        mip.load_state_dict(W.mip_state("small", hidden=320))
        prop, mip = prop.cuda().eval(), mip.cuda().eval()
    else:
        prop, mip = _nets()
    pose, focal = _pose_focal()
    with torch.no_grad():
        one = render_image(mip, prop, pose, H, focal, NEAR, FAR, SAMPLES, white_bkg=True, render_depth=True, seed=5, ipe=True)
        plain = render_image(mip, prop, pose, H, focal, NEAR, FAR, SAMPLES, white_bkg=True, render_depth=True, seed=5)
        parts = []
        for r in range(3):
            s0, e0 = parallel.shard_range(H * H, r, 3, align=256)
            parts.append(render_image(mip, prop, pose, H, focal, NEAR, FAR, SAMPLES, white_bkg=True, render_depth=True, seed=5, ipe=True, _shard=(s0, e0)))
    rgb3 = parts[0]["to_image"](torch.cat([q["rgb_rays"] for q in parts]), 3)
    dep3 = parts[0]["to_image"](torch.cat([q["depth_rays"] for q in parts]).unsqueeze(-1), 1)
    assert torch.equal(rgb3, one["rgb"]) and torch.equal(dep3.expand(3, -1, -1), one["depth_img"])
    assert not torch.equal(one["rgb"], plain["rgb"])                                    # the integrated PE is really on
    if not wide:
        # the regression: a shard encoded with the norm of its own rays is another image
        s0, e0 = parallel.shard_range(H * H, 1, 3, align=256)
        fx, fy = (float(focal[1]), float(focal[0])) if isinstance(focal, (tuple, list)) else (float(focal), float(focal))
        rays = ops.generate_rays(pose, H, H, fx, fy, pose.device)
        rays = rays.view(H, H, 6).reshape(2, 50, 2, 50, 6).permute(0, 2, 1, 3, 4).reshape(-1, 6)[s0:e0].contiguous()
        P = ops.current_precision()
        z_base = torch.linspace(NEAR, FAR, 64).cuda()
        with torch.no_grad():
            own, _, _, _ = ops.render_rays(prop.packed(P), mip.packed(P, wide=True), P, rays, z_base, None, None, SAMPLES, NEAR, FAR, True,
                                           ipe_radius=2.0 / (12.0 ** 0.5) / fx, seed=5, rng_ray_offset=s0)
        assert not torch.equal(own, parts[1]["rgb_rays"])


# ------------------------------------------------------------------------------------------------ data-parallel training step
N_TRAIN, C_TRAIN, F_TRAIN = 64, 32, 64


def _train_grads(prop, mip, rays, tgt, u1, u2):
    """Body of train.py:164-199 (non-Ref) -> parameter gradients of the summed loss, through the nerf_amd training path."""
    import torch.nn.functional as F
    from nerf_amd.addtional import ProposalLoss, ProposalNetwork, getBounds
    from nerf_amd.mip_methods import maxBlurFilter
    from nerf_amd.nerf_base import NeRF
    from nerf_amd.utils import inverseSample
    res = (FAR - NEAR) / C_TRAIN
    z_c = torch.linspace(NEAR, FAR - res, C_TRAIN).cuda() + u1 * res
    pts = (rays[:, None, :3] + rays[:, None, 3:] * z_c[:, :, None]).contiguous()
    pw = maxBlurFilter(ProposalNetwork.get_weights(F.softplus(prop.forward(pts)), z_c, rays[:, 3:]), 0.01)
    z_f, below = inverseSample(pw, z_c, F_TRAIN + 1, sort=True, u=u2)
    z_f = z_f[..., :-1].contiguous()
    rend, wts, _ = NeRF.render(mip.forward(NeRF.length2pts(rays, z_f)), z_f, rays[:, 3:])
    loss = ProposalLoss()(getBounds(pw, below), wts.detach()) + torch.mean((rend - tgt) ** 2)
    for p in list(mip.parameters()) + list(prop.parameters()):
        p.grad = None
    loss.backward()
    return [p.grad.clone() for p in list(mip.parameters()) + list(prop.parameters())]


def _train_inputs(rank):
    g = torch.Generator().manual_seed(100 + rank)
    d = torch.nn.functional.normalize(torch.randn(N_TRAIN, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, -1.0]), dim=-1)
    rays = torch.cat((torch.tensor([0.0, 0.0, 4.0]).expand(N_TRAIN, 3), d), -1).cuda().contiguous()
    return rays, torch.rand(N_TRAIN, 3, generator=g).cuda(), torch.rand(N_TRAIN, C_TRAIN, generator=g).cuda(), torch.rand(N_TRAIN, F_TRAIN + 1, generator=g).cuda()


def _train_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import nerf_amd
    from nerf_amd import parallel
    nerf_amd.set_precision("fp32")
    prop, mip = _nets()
    prop.train(); mip.train()
    parallel.broadcast_parameters([mip, prop])
    _train_grads(prop, mip, *_train_inputs(rank))
    parallel.allreduce_gradients([mip, prop])                      # mean over the ranks = gradient of the mean loss over all rays
    if rank == 0:
        torch.save([p.grad.cpu() for p in list(mip.parameters()) + list(prop.parameters())], out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_step_gradients(tmp_path):
    """ddp_train.py's step: every rank its own rays, one flat all-reduce; the reduced gradient equals the mean of the two ranks'
    single-process gradients."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    out_path = str(tmp_path / "grads.pt")
    mp.spawn(_train_worker, args=(2, 29541, out_path), nprocs=2, join=True)
    got = torch.load(out_path)
    sys.path.insert(0, ROOT)
    import nerf_amd
    nerf_amd.set_precision("fp32")
    prop, mip = _nets()
    prop.train(); mip.train()
    g0 = _train_grads(prop, mip, *_train_inputs(0))
    g1 = _train_grads(prop, mip, *_train_inputs(1))
    for a, b0, b1 in zip(got, g0, g1):
        want = (0.5 * (b0 + b1)).cpu()
        assert (a - want).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item())



def test_native_amp_flow_matches_plain_bf16():
    """train.py's `opt_mode native` flow -- torch.autocast around the step, GradScaler around backward -- selects the bf16 kernels and
    gives the same parameter gradients as the plain bf16 path (a power-of-two loss scale is exact in bf16/fp32 arithmetic)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    import torch.nn.functional as F
    import nerf_amd
    from nerf_amd import ops
    from nerf_amd.addtional import ProposalLoss, ProposalNetwork, getBounds
    from nerf_amd.mip_methods import maxBlurFilter
    from nerf_amd.nerf_base import NeRF
    from nerf_amd.utils import inverseSample
    rays, tgt, u1, u2 = _train_inputs(3)
    res = (FAR - NEAR) / C_TRAIN

    def loss_of(prop, mip):
        z_c = torch.linspace(NEAR, FAR - res, C_TRAIN).cuda() + u1 * res
        pts = (rays[:, None, :3] + rays[:, None, 3:] * z_c[:, :, None]).contiguous()
        pw = maxBlurFilter(ProposalNetwork.get_weights(F.softplus(prop.forward(pts)), z_c, rays[:, 3:]), 0.01)
        z_f, below = inverseSample(pw, z_c, F_TRAIN + 1, sort=True, u=u2)
        z_f = z_f[..., :-1].contiguous()
        rend, wts, _ = NeRF.render(mip.forward(NeRF.length2pts(rays, z_f)), z_f, rays[:, 3:])
        return ProposalLoss()(getBounds(pw, below), wts.detach()) + torch.mean((rend - tgt) ** 2)

    grads = []
    for amp in (False, True):
        nerf_amd.set_precision("bf16" if not amp else None)          # None: follow torch.autocast, like the reference's layers
        prop, mip = _nets()
        prop.train(); mip.train()
        params = list(mip.parameters()) + list(prop.parameters())
        if amp:
            scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
            opt = torch.optim.SGD(params, lr=0.0)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                assert ops.current_precision() == ops.BF16
                loss = loss_of(prop, mip)
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
        else:
            loss_of(prop, mip).backward()
        grads.append([p.grad.clone() for p in params])
    nerf_amd.set_precision("fp32")
    for a, b in zip(*grads):
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item())


def test_training_step_captured_in_a_hipgraph():
    """The whole step (HIP training forwards, GEMM-chain / HIP backward, Adam, weight re-pack) can be captured once and replayed:
    nothing in it synchronises or allocates outside torch's graph pool, and the ctypes-launched kernels land on the capture stream.
    Three replays == three eager steps from the same state on the same batch."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    import torch.nn.functional as F
    import nerf_amd
    from nerf_amd.addtional import ProposalLoss, ProposalNetwork, getBounds
    from nerf_amd.mip_methods import maxBlurFilter
    from nerf_amd.nerf_base import NeRF
    from nerf_amd.utils import inverseSample
    nerf_amd.set_precision("fp32")
    rays, tgt, u1, u2 = _train_inputs(5)
    res = (FAR - NEAR) / C_TRAIN
    base = torch.linspace(NEAR, FAR - res, C_TRAIN).cuda()

    def make():
        prop, mip = _nets()
        prop.train(); mip.train()
        opt = torch.optim.Adam(list(mip.parameters()) + list(prop.parameters()), lr=1e-3, capturable=True)

        def step():
            z_c = base + u1 * res
            pts = (rays[:, None, :3] + rays[:, None, 3:] * z_c[:, :, None]).contiguous()
            pw = maxBlurFilter(ProposalNetwork.get_weights(F.softplus(prop.forward(pts)), z_c, rays[:, 3:]), 0.01)
            z_f, below = inverseSample(pw, z_c, F_TRAIN + 1, sort=True, u=u2)
            z_f = z_f[..., :-1].contiguous()
            rend, wts, _ = NeRF.render(mip.forward(NeRF.length2pts(rays, z_f)), z_f, rays[:, 3:])
            loss = ProposalLoss()(getBounds(pw, below), wts.detach()) + torch.mean((rend - tgt) ** 2)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        return prop, mip, step

    prop_e, mip_e, step_e = make()
    for _ in range(4):                                                # 1 warm-up + 3
        step_e()
    prop_g, mip_g, step_g = make()
    step_g()                                                          # warm-up (lazy kernel attributes, optimizer state) outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step_g()                                                      # (capture does not execute)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(list(mip_g.parameters()) + list(prop_g.parameters()), list(mip_e.parameters()) + list(prop_e.parameters())):
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())
    # and the re-pack inside the graph followed the updated weights: the captured forward renders what an eager forward renders now
    with torch.no_grad():
        probe = torch.rand(64, 8, 6).cuda()
        assert torch.allclose(mip_g.eval().forward(probe), mip_e.eval().forward(probe), atol=1e-5)


def test_refnerf_training_step_captured_in_a_hipgraph():
    """The Ref-NeRF step with prop_normal (train.py:164-199: get_grad normals on both networks, normal / back-face losses, Adam) is
    sync-free too -- constant tables live on the device, coarse_grad_select has no data-dependent shape, the device-side VJP re-uses
    its graph -- so it can be captured and replayed: three replays == three eager steps on the same batch and bottle-neck noise."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.nn.functional as F
    import weights as W
    import nerf_amd
    from nerf_amd.addtional import ProposalLoss, ProposalNetwork, getBounds
    from nerf_amd.mip_methods import maxBlurFilter
    from nerf_amd.nerf_base import NeRF
    from nerf_amd.ref_model import BackFaceLoss, RefNeRF, WeightedNormalLoss
    from nerf_amd.utils import inverseSample
    nerf_amd.set_precision("fp32")
    rays, tgt, u1, u2 = _train_inputs(6)
    res = (FAR - NEAR) / C_TRAIN
    base = torch.linspace(NEAR, FAR - res, C_TRAIN).cuda()
    real_normal = torch.normal
    noise = {}

    def fixed_normal(mean, std, size, **kw):                          # one tensor per shape, created by the eager warm-up steps
        if tuple(size) not in noise:
            noise[tuple(size)] = (torch.randn(tuple(size), generator=torch.Generator().manual_seed(8)) * 0.1).cuda()
        return noise[tuple(size)]

    def make():
        prop, _ = _nets()
        net = RefNeRF(10, 4)
        net.load_state_dict(W.ref_state("small"))
        prop, net = prop.train(), net.cuda().train()
        net.noise_rng = "torch"                                      # the fixed perturbation enters through torch.normal (default: in-kernel Philox,
                                                                      # whose host-drawn key a capture would bake in -- TrainStep keys it by a device scalar)
        opt = torch.optim.Adam(list(net.parameters()) + list(prop.parameters()), lr=1e-3, capturable=True)

        def step():
            z_c = base + u1 * res
            pts = (rays[:, None, :3] + rays[:, None, 3:] * z_c[:, :, None]).contiguous().requires_grad_(True)
            dens = prop.forward(pts)
            coarse_grad = -RefNeRF.get_grad(dens, pts)
            pw = maxBlurFilter(ProposalNetwork.get_weights(F.softplus(dens), z_c, rays[:, 3:]), 0.01)
            fl, below = inverseSample(pw, z_c, F_TRAIN + 1, sort=True, u=u2)
            samples, fl, below, sort_ids = NeRF.coarseFineMerge(rays, z_c, fl, below)
            pos, dd = samples.split((3, 3), dim=-1)
            pos = pos.contiguous().requires_grad_(True)
            rgbo, nrm = net.forward(pos, dd.contiguous())
            dgrad = -RefNeRF.get_grad(rgbo[..., -1], pos)
            rgbo[..., -1] = F.softplus(rgbo[..., -1] + 0.5)
            rend, wts, _ = NeRF.render(rgbo, fl, rays[:, 3:], net.density_act)
            loss = (ProposalLoss()(getBounds(pw, below), wts.detach()) + torch.mean((rend - tgt) ** 2) + 0.1 * BackFaceLoss()(wts, nrm, dd)
                    + 4e-4 * (WeightedNormalLoss()(wts, dgrad, nrm)
                              + 0.1 * WeightedNormalLoss()(pw, RefNeRF.coarse_grad_select(dgrad, sort_ids, C_TRAIN).detach(), coarse_grad)))
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        return prop, net, step

    torch.normal = fixed_normal
    try:
        prop_e, net_e, step_e = make()
        for _ in range(4):                                            # 1 warm-up + 3
            step_e()
        prop_g, net_g, step_g = make()
        step_g()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step_g()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
    finally:
        torch.normal = real_normal
    for a, b in zip(list(net_g.parameters()) + list(prop_g.parameters()), list(net_e.parameters()) + list(prop_e.parameters())):
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())


@pytest.mark.gpu
def test_device_resident_train_step_eager_and_replayed():
    """nerf_amd.training.TrainStep = the iteration of train.py:151-218 with pose, pixel table, seed, Adam step count and learning rate in
    device memory.  (i) Replaying the captured hipGraph reproduces the eager iterations from the same state -- through a change of image /
    pose AND a change of the learning rate between iterations, which a graph with baked launch arguments would miss; (ii) every iteration
    draws new random numbers (the device seed advances, the losses differ); (iii) it learns (the loss on a fixed image falls)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import nerf_oracle as O                                      # (test infrastructure: poses / focal only)
    import nerf_amd
    from nerf_amd.optim import Adam
    from nerf_amd.training import TrainStep
    nerf_amd.set_precision("fp32")
    gen = torch.Generator().manual_seed(3)
    imgs = [torch.rand(3, 40, 40, generator=gen).cuda() for _ in range(2)]
    poses = [O.pose_spherical(a, -30.0, 4.0)[:3].contiguous().cuda() for a in (20.0, 140.0)]
    focal = O.fov2focal(0.6911112070083618, (40, 40))
    lrs = [1e-3, 1e-3, 1e-3, 4e-4, 4e-4, 2e-4]                              # "scheduler": rewritten on the host between iterations

    def run(graphed):
        prop, mip = _nets()
        prop.train(); mip.train()
        opt = Adam(list(mip.parameters()) + list(prop.parameters()), lr=lrs[0], lr_on_device=True)
        step = TrainStep(prop, mip, opt, (40, 40), focal, NEAR, FAR, ray_num=96, coarse_pnum=32, fine_pnum=64, seed=1234)
        step.set_image(imgs[0], poses[0])
        losses, seeds = [], []
        if graphed:
            step.capture(warmup=2)                                           # iterations 0, 1 eager, then recorded
            losses += [None, None]
        for it in range(2 if graphed else 0, len(lrs)):
            for gr in opt.param_groups:
                gr["lr"] = lrs[it]
            loss, _ = step(imgs[it % 2], poses[it % 2]) if it >= 3 else step()
            losses.append(float(loss.item()))
            seeds.append(int(step.seed.item()))
        return prop, mip, losses, seeds

    prop_e, mip_e, loss_e, seed_e = run(False)
    prop_g, mip_g, loss_g, seed_g = run(True)
    for a, b in zip(list(mip_g.parameters()) + list(prop_g.parameters()), list(mip_e.parameters()) + list(prop_e.parameters())):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    assert seed_e[-1] == seed_g[-1] and len(set(seed_e)) == len(seed_e)       # the seed advanced every iteration, identically on both paths
    for a, b in zip(loss_e[2:], loss_g[2:]):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b))
    assert len(set(round(v, 9) for v in loss_e)) == len(loss_e)              # new rays / uniforms every iteration
    # (iii) a longer eager run on one image: the training loss falls
    prop, mip = _nets()
    prop.train(); mip.train()
    opt = Adam(list(mip.parameters()) + list(prop.parameters()), lr=1e-3, lr_on_device=True)
    step = TrainStep(prop, mip, opt, (40, 40), focal, NEAR, FAR, ray_num=256, coarse_pnum=32, fine_pnum=64, seed=7)
    calls = []
    hooked = TrainStep(prop, mip, opt, (40, 40), focal, NEAR, FAR, ray_num=32, coarse_pnum=32, fine_pnum=32, seed=1, grad_hook=lambda: calls.append(1))
    hooked.set_image(imgs[0], poses[0])
    hooked(); hooked()
    assert len(calls) == 2                                                   # the gradient hook (ddp_train.py's all-reduce slot) runs once per iteration
    with pytest.raises(RuntimeError):
        hooked.capture()
    step.set_image(imgs[0] * 0.0 + 0.6, poses[0])                            # a constant-colour image: learnable in a few dozen iterations
    step.capture(warmup=2)
    first = sum(float(step()[1].item()) for _ in range(5)) / 5
    for _ in range(120):
        step()
    last = sum(float(step()[1].item()) for _ in range(5)) / 5
    assert last < 0.5 * first, (first, last)
    step.set_crop((0.5, 0.5))                                                 # the centre-crop phase of train.py:155: another pixel table
    assert step.graph is None                                                 # -> the captured graph is dropped, the eager path just works
    assert torch.isfinite(step()[0]).item()
    step.capture(warmup=1)
    assert torch.isfinite(step()[0]).item()


@pytest.mark.gpu
def test_device_resident_train_step_refnerf_branch():
    """TrainStep with a RefNeRF fine network and prop_normal (train.py:165-168,176-187): hipGraph replays == eager iterations from the same
    state (the bottle-neck noise pinned to one tensor per shape so that both paths see the same draw), and the step trains (loss falls)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import weights as W
    from oracle import nerf_oracle as O
    import nerf_amd
    from nerf_amd.optim import Adam
    from nerf_amd.ref_model import RefNeRF
    from nerf_amd.training import TrainStep
    nerf_amd.set_precision("fp32")
    gen = torch.Generator().manual_seed(5)
    img = (torch.rand(3, 40, 40, generator=gen) * 0.2 + 0.4).cuda()
    pose = O.pose_spherical(20.0, -30.0, 4.0)[:3].contiguous().cuda()
    focal = O.fov2focal(0.6911112070083618, (40, 40))
    real_normal, noise = torch.normal, {}

    def fixed_normal(mean, std, size, **kw):
        if tuple(size) not in noise:
            noise[tuple(size)] = (torch.randn(tuple(size), generator=torch.Generator().manual_seed(8)) * std).cuda()
        return noise[tuple(size)]

    def run(graphed, iters):
        prop, _ = _nets()
        net = RefNeRF(10, 4)
        net.load_state_dict(W.ref_state("small"))
        prop, net = prop.train(), net.cuda().train()
        opt = Adam(list(net.parameters()) + list(prop.parameters()), lr=5e-4, lr_on_device=True)
        step = TrainStep(prop, net, opt, (40, 40), focal, NEAR, FAR, ray_num=64, coarse_pnum=32, fine_pnum=32, seed=99, prop_normal=True)
        assert step.is_ref and step.prop_normal
        step.set_image(img, pose)
        if graphed:
            step.capture(warmup=2)
        losses = [float(step()[1].item()) for _ in range(iters - (2 if graphed else 0))]
        return prop, net, losses, int(step.seed.item())

    torch.normal = fixed_normal
    try:
        prop_e, net_e, loss_e, seed_e = run(False, 6)
        prop_g, net_g, loss_g, seed_g = run(True, 6)
    finally:
        torch.normal = real_normal
    assert seed_e == seed_g
    for a, b in zip(list(net_g.parameters()) + list(prop_g.parameters()), list(net_e.parameters()) + list(prop_e.parameters())):
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())
    for a, b in zip(loss_e[2:], loss_g):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b))
    torch.normal = fixed_normal
    try:
        _, _, long_losses, _ = run(True, 150)
    finally:
        torch.normal = real_normal
    assert sum(long_losses[-10:]) < 0.6 * sum(long_losses[:10]), (long_losses[:10], long_losses[-10:])


@pytest.mark.gpu
@pytest.mark.parametrize("branch", ["mip", "ref"])
def test_train_step_losses_equal_the_oracles_step_on_the_same_draws(branch):
    """TrainStep's iteration against the oracle's restatement of train.py:164-199 on IDENTICAL rays, coarse depths, uniforms (and bottle-neck
    noise, in-kernel Philox since round 5): the in-kernel draws are functions of the device seed, so the test re-derives them with the same entry points and hands them to
    the oracle.  The Ref-NeRF branch includes the reference's positional quirk (train.py:182: `mip_net.density_act` lands in `mul_norm`, so
    the depths are NOT scaled by |d|) -- with un-normalised ray directions a scaled composite gives another loss, so this pins it."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.nn.functional as F
    import weights as W
    from oracle import nerf_oracle as O
    import nerf_amd
    from nerf_amd import ops
    from nerf_amd.optim import Adam
    from nerf_amd.ref_model import RefNeRF
    from nerf_amd.training import TrainStep
    from nerf_amd.utils import randomFromOneImage
    nerf_amd.set_precision("fp32")
    N, C_N, F_N, SEED = 48, 16, 32, 2468
    gen = torch.Generator().manual_seed(5)
    img = torch.rand(3, 40, 40, generator=gen).cuda()
    pose = O.pose_spherical(20.0, -30.0, 4.0)[:3].contiguous().cuda()
    focal = O.fov2focal(0.6911112070083618, (40, 40))          # |d| != 1 for these rays: the mul_norm quirk is visible
    prop, mip = _nets()
    prop.train()
    if branch == "ref":
        net = RefNeRF(10, 4)
        net.load_state_dict(W.ref_state("small"))
        net = net.cuda().train()
    else:
        net = mip.train()
    opt = Adam(list(net.parameters()) + list(prop.parameters()), lr=0.0, lr_on_device=True)
    step = TrainStep(prop, net, opt, (40, 40), focal, NEAR, FAR, ray_num=N, coarse_pnum=C_N, fine_pnum=F_N, seed=SEED, prop_normal=(branch == "ref"))
    step.set_image(img, pose)
    # the step's own draws, re-derived from the same device seed through the same entry points
    seed = torch.full((1,), SEED, dtype=torch.int64, device="cuda")
    pixels, coords = randomFromOneImage(img, (1.0, 1.0))
    fx, fy = float(focal[1]), float(focal[0])
    _, z_c, tgt, rays = ops.sample_training_rays_dev(pixels, coords, pose, fx, fy, NEAR, FAR, N, C_N, seed)
    u = ops.philox_uniforms((N, F_N + 1), seed_dev=seed)
    # the bottle-neck perturbation too (round 5): drawn inside the training forward from the step's device seed and the sample index --
    # the same deviates as a tensor for the oracle (RefNeRF's default noise std 0.1; sample m = ray * (F_N + C_N) + position in the merged row)
    noise = ops.philox_normal(N * (F_N + C_N), 0.1, seed_dev=seed).view(N, F_N + C_N, 128)
    loss, img_loss = step()
    rays_c, zc_c, tgt_c, u_c = rays.cpu(), z_c.cpu(), tgt.cpu(), u.cpu()
    if branch == "ref":
        out = O.ref_train_step(W.proposal_state("small"), W.ref_state("small"), rays_c, zc_c, u_c, noise.cpu(), tgt_c, F_N)
        want_loss, want_img = float(out["loss"]), float(out["img_loss"])
        # the quirk matters on these rays: compositing with |d|-scaled depths gives other weights (which the proposal / normal / back-face
        # losses see; the image loss of these nearly colour-constant test networks does not)
        _, w_scaled, _ = O.composite(torch.cat((out["rgbo_raw"][..., :3], F.softplus(out["rgbo_raw"][..., 3:] + 0.5)), -1), out["z_merged"], rays_c[:, 3:], mul_norm=True)
        assert float((w_scaled - out["weights"]).abs().max()) > 1e-3
    else:
        P, M = W.proposal_state("small"), W.mip_state("small")
        pts = rays_c[:, None, :3] + rays_c[:, None, 3:] * zc_c[:, :, None]
        pw = O.max_blur(O.sigma_to_weights(F.softplus(O.proposal_forward(P, pts)), zc_c, rays_c[:, 3:]), 0.01)
        z_f, below = O.inverse_sample(pw, zc_c, u_c, sort=True)
        z_f = z_f[..., :-1]
        rend, wts, _ = O.composite(O.mip_forward(M, O.length2pts(rays_c, z_f)), z_f, rays_c[:, 3:])
        want_img = float(torch.mean((rend - tgt_c) ** 2))
        want_loss = want_img + float(O.proposal_loss(O.get_bounds(pw, below), wts))
    assert abs(float(img_loss) - want_img) <= 2e-5 * max(1.0, want_img), (float(img_loss), want_img)
    assert abs(float(loss) - want_loss) <= 2e-4 * max(1.0, abs(want_loss)), (float(loss), want_loss)


# ------------------------------------------------------------------------------------------------ bench.py's real N > 1 path on the box
def _bench_line(tmp_path, tag, *args, gpus=1):
    """Run bench.py as the driver does (stand-alone for N = 1; for N = 2 it re-executes itself under torch.distributed.run) with the gloo
    control-flow backend -- two ranks on the box's one GPU, which RCCL refuses -- and return the parsed JSON line."""
    import json
    import subprocess
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--cpu-rays", "2500"] + list(args)
    r = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0, (tag, r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (tag, r.stdout[-2000:])                       # ONE JSON line, from rank 0
    return json.loads(lines[0])


def _key_tree(d, prefix=""):
    """every key path of a record (lists are leaves); the per-rank entries are compared by presence, not by length"""
    out = set()
    for k, v in d.items():
        out.add(prefix + k)
        if isinstance(v, dict):
            out |= _key_tree(v, prefix + k + ".")
    return out


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline")


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("mode", ["render", "render-strong", "train-ddp", "train-ddp-ref"])
def test_bench_n_gt_1_path_runs_on_the_box_and_its_line_is_self_contained(mode, tmp_path):
    """What the driver's scaling run executes -- `bench.py --gpus N` with WORLD_SIZE > 1: process group, per-rank device, barriers around
    the timed region, max-over-ranks, the JSON line -- runs here with two ranks on the one GPU for all three modes (the only thing a
    one-GPU box cannot provide is RCCL across two devices).  The two-rank line must carry every key of the one-rank line (cpu_baseline,
    roofline incl. the per-rank MFMA-stream ceilings, train_step for the headline mode) plus the per-rank step times; render-strong must
    produce the SAME image bit for bit whatever N is (every uniform is a function of the global ray index)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    extra = {"render": [], "render-strong": ["--mode", "render-strong"], "train-ddp": ["--mode", "train-ddp", "--train-rays", "2048"],
             "train-ddp-ref": ["--mode", "train-ddp", "--model", "ref", "--train-rays", "1024"]}[mode]      # (BASELINE configs[3] on the DP training path)
    img = {}
    if mode == "render-strong":
        img = {1: str(tmp_path / "img1.pt"), 2: str(tmp_path / "img2.pt")}
    one = _bench_line(tmp_path, mode + "/1", *extra, *(["--dump-image", img[1]] if img else []), gpus=1)
    two = _bench_line(tmp_path, mode + "/2", *extra, *(["--dump-image", img[2]] if img else []), gpus=2)
    for rec, n in ((one, 1), (two, 2)):
        for k in CONTRACT_KEYS:
            assert k in rec, (mode, n, k)
        assert rec["n_gpus"] == n and rec["steps"] == 2 and rec["warmup"] == 1 and rec["value"] > 0
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rec["roofline"])
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(rec["cpu_baseline"]) and rec["cpu_baseline"]["value"] > 0
        sp = rec["ms_per_step_ranks"]
        assert len(sp["all"]) == n and sp["min"] <= sp["max"] <= rec["ms_per_step"] * 1.0001
        ref = rec["roofline"]["mfma_stream_ref"]
        assert len(ref["per_rank"]) == n and all(r["constant_operands"] > 0 and r["weights_x_relu_activations"] > 0 for r in ref["per_rank"])
    missing = _key_tree(one) - _key_tree(two)
    assert not missing, (mode, sorted(missing))                          # the two-rank line has every key of the one-rank line
    assert two["scaling"] == ("strong" if mode == "render-strong" else "weak")
    if mode == "render":
        assert "train_step" in two and two["train_step"]["rays_16384"]["ms_per_iter"] > 0 and two["train_step"]["rays_16384"]["ms_min"] <= two["train_step"]["rays_16384"]["ms_max"]
        assert len(two["roofline"]["ms_per_launch_ranks"]) == 2
        # weak scaling on ONE device: two ranks time-share it, so the whole-job rate stays within the one-rank rate's neighbourhood
        assert 0.5 * one["value"] < two["value"] < 1.3 * one["value"], (one["value"], two["value"])
    if mode == "train-ddp":
        assert two["allreduce"]["backend"] == "gloo" and two["allreduce"]["us_per_step"] > 0 and len(two["allreduce"]["us_per_step_ranks"]) == 2
        assert two["allreduce"]["elements"] == 530052 + 214017
    if mode == "train-ddp-ref":
        assert two["allreduce"]["backend"] == "gloo" and two["allreduce"]["us_per_step"] > 0
        assert two["allreduce"]["elements"] == 1075854 + 214017 and "Ref-NeRF" in two["config"]["workload"]      # 5.16 MB per step
    if mode == "render-strong":
        a, b = torch.load(img[1]), torch.load(img[2])
        assert a.shape == (800 * 800, 4) and torch.equal(a, b)              # N = 2 == N = 1, bit for bit
