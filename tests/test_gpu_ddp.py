"""Data-parallel training on real hardware:

* the reference's OWN mechanism -- `nn.parallel.DistributedDataParallel(mip_net, device_ids=[gpu])`, ddp_train.py:98 -- wrapped around
  `nerf_amd.MipNeRF`: DDP's autograd hooks fire on the gradients the hand-written backward returns, the reduced gradients are the mean
  over the ranks (the proposal network stays un-reduced, as in the reference);
* the native path -- `parallel.FlatGradients`: the weight-gradient kernels write into ONE persistent flat buffer, ONE all-reduce -- gives
  the same reduced gradients, for both networks;
* every `param_com` primitive (param_com.py:13-54, model_average.py:230-260) on DEVICE modules;
* RCCL itself: a process group with backend "nccl" on the box's GPU, the flat all-reduce eager and CAPTURED inside the training
  iteration's hipGraph.

Two ranks share the box's single GPU, so the two-rank tests rendezvous over gloo (RCCL refuses two ranks on one device); the RCCL test
runs the one rank a 1-GPU box allows.  The N-GPU launch itself is the driver's (bench.py --gpus N)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
NEAR, FAR = 2.0, 6.0
N_TRAIN, C_TRAIN, F_TRAIN = 64, 32, 64


def _nets(tag="small"):
    import weights as W
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.mip_model import MipNeRF
    prop, mip = ProposalNetwork(10, 256), MipNeRF(10, 4, 256)
    prop.load_state_dict(W.proposal_state(tag))
    mip.load_state_dict(W.mip_state(tag))
    return prop.cuda().train(), mip.cuda().train()


def _inputs(rank):
    g = torch.Generator().manual_seed(100 + rank)
    d = torch.nn.functional.normalize(torch.randn(N_TRAIN, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, -1.0]), dim=-1)
    rays = torch.cat((torch.tensor([0.0, 0.0, 4.0]).expand(N_TRAIN, 3), d), -1).cuda().contiguous()
    return rays, torch.rand(N_TRAIN, 3, generator=g).cuda(), torch.rand(N_TRAIN, C_TRAIN, generator=g).cuda(), torch.rand(N_TRAIN, F_TRAIN + 1, generator=g).cuda()


def _loss(prop, mip, rays, tgt, u1, u2):
    """train.py:164-199 (non-Ref) with `mip` possibly a DistributedDataParallel wrapper (the reference calls `.forward` on it)."""
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
    return ProposalLoss()(getBounds(pw, below), wts.detach()) + torch.mean((rend - tgt) ** 2)


def _ddp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import nerf_amd
    from nerf_amd import parallel
    nerf_amd.set_precision("fp32")
    # (1) the reference's mechanism, ddp_train.py:97-99: the fine network wrapped, the proposal network not
    prop, mip = _nets()
    ddp = torch.nn.parallel.DistributedDataParallel(mip, device_ids=[0])
    _loss(prop, ddp, *_inputs(rank)).backward()
    ddp_mip = [p.grad.clone() for p in mip.parameters()]
    ddp_prop = [p.grad.clone() for p in prop.parameters()]
    # (2) the native path: one persistent flat buffer, one all-reduce over both networks
    prop2, mip2 = _nets()
    flat = parallel.FlatGradients([mip2, prop2])
    _loss(prop2, mip2, *_inputs(rank)).backward()
    local = flat.flat.clone()
    assert all(p.grad is flat.views[p] for p in flat.params)             # the kernels wrote into the views, autograd replaced nothing
    n = flat.all_reduce()
    torch.save({"ddp_mip": [g.cpu() for g in ddp_mip], "ddp_prop": [g.cpu() for g in ddp_prop], "flat": flat.flat.cpu(), "local": local.cpu(), "n": n,
                "flat_mip": [p.grad.cpu() for p in mip2.parameters()], "flat_prop": [p.grad.cpu() for p in prop2.parameters()]},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_data_parallel_wrapper_and_flat_gradients(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    mp.spawn(_ddp_worker, args=(2, 29561, str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(str(tmp_path / ("rank%d.pt" % r))) for r in range(2)]
    import nerf_amd
    nerf_amd.set_precision("fp32")
    single = []
    for r in range(2):                                                    # each rank's gradients in this process, no wrapper, ordinary autograd
        prop, mip = _nets()
        _loss(prop, mip, *_inputs(r)).backward()
        single.append(([p.grad.cpu() for p in mip.parameters()], [p.grad.cpu() for p in prop.parameters()]))
    close = lambda a, b: (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item())
    for r in range(2):
        assert got[r]["n"] == 530052 + 214017
        for k, g in enumerate(got[r]["ddp_mip"]):                         # DDP averaged the fine network over the ranks ...
            assert close(g, 0.5 * (single[0][0][k] + single[1][0][k]))
        for k, g in enumerate(got[r]["ddp_prop"]):                        # ... and left the proposal network alone (ddp_train.py:97-99)
            assert torch.equal(g, single[r][1][k])
        for k, g in enumerate(got[r]["flat_mip"]):                        # the flat path: both networks averaged
            assert close(g, 0.5 * (single[0][0][k] + single[1][0][k])) and close(g, got[r]["ddp_mip"][k])
        for k, g in enumerate(got[r]["flat_prop"]):
            assert close(g, 0.5 * (single[0][1][k] + single[1][1][k]))
        # before the reduction the flat buffer held exactly this rank's autograd gradients (same kernels, written in place)
        assert torch.equal(got[r]["local"], torch.cat([g.reshape(-1) for g in single[r][0] + single[r][1]]))
    assert torch.equal(got[0]["flat"], got[1]["flat"])


def test_flat_gradients_accumulation_window_and_adam():
    """FlatGradients without a process group: first backward of a window overwrites (no zeroing pass), a second one accumulates,
    optimizer.step() opens a new window; nerf_amd.optim.Adam steps from the views exactly as from ordinary gradients."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import nerf_amd
    from nerf_amd import parallel
    from nerf_amd.optim import Adam
    nerf_amd.set_precision("fp32")
    prop, mip = _nets()
    opt = Adam(list(mip.parameters()) + list(prop.parameters()), lr=1e-3)
    flat = parallel.FlatGradients([mip, prop], opt)
    flat.flat.fill_(123.0)                                                # stale contents must not survive the first backward of a window
    _loss(prop, mip, *_inputs(0)).backward()
    g0 = flat.flat.clone()
    _loss(prop, mip, *_inputs(1)).backward()                              # same window: accumulates
    g01 = flat.flat.clone()
    prop_r, mip_r = _nets()
    opt_r = Adam(list(mip_r.parameters()) + list(prop_r.parameters()), lr=1e-3)
    _loss(prop_r, mip_r, *_inputs(0)).backward()
    r0 = torch.cat([p.grad.reshape(-1) for p in list(mip_r.parameters()) + list(prop_r.parameters())])
    _loss(prop_r, mip_r, *_inputs(1)).backward()
    r01 = torch.cat([p.grad.reshape(-1) for p in list(mip_r.parameters()) + list(prop_r.parameters())])
    assert torch.equal(g0, r0)
    assert (g01 - r01).abs().max().item() <= 1e-6 * r01.abs().max().item()
    flat.flat.copy_(r01)
    opt.step(); opt_r.step()
    for a, b in zip(list(mip.parameters()) + list(prop.parameters()), list(mip_r.parameters()) + list(prop_r.parameters())):
        assert torch.equal(a, b)
    opt.zero_grad(set_to_none=True)                                       # a caller's habit from train.py:194 must not break the binding
    _loss(prop, mip, *_inputs(0)).backward()                              # new window (the optimizer step opened it): overwrite again
    prop_r.zero_grad(); mip_r.zero_grad()
    _loss(prop_r, mip_r, *_inputs(0)).backward()
    assert torch.equal(flat.flat, torch.cat([p.grad.reshape(-1) for p in list(mip_r.parameters()) + list(prop_r.parameters())]))
    assert all(p.grad is flat.views[p] for p in flat.params)


def _pc_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import nerf_amd
    from nerf_amd import param_com as pc
    nerf_amd.set_precision("fp32")

    def model(r):
        prop, _ = _nets()
        with torch.no_grad():
            for k, p in enumerate(prop.parameters()):
                p.mul_(1.0 + 0.25 * r).add_(0.001 * (k + 1) * (r + 1))      # rank-dependent, still a sane network
        return prop.eval()
    probe = torch.rand(7, 5, 3, generator=torch.Generator().manual_seed(3)).cuda()
    vec = lambda m: torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
    res = {"own": vec(model(rank))}
    m = model(rank)
    with torch.no_grad():
        before = m.forward(probe).clone()
    pc.param_broadcast(m, src_rank=1)
    with torch.no_grad():
        res["bcast"], res["bcast_fwd"], res["fwd_before"] = vec(m), m.forward(probe).cpu(), before.cpu()
    m = model(rank); pc.param_all_reduce(m); res["allred"] = vec(m)
    m = model(rank); pc.param_reduce(m, [0.25, 0.75], rank, dst_rank=0); res["reduce"] = vec(m)
    m, tmp = model(rank), model(5)
    if rank == 0:
        pc.param_recv_avg(m, tmp, [0.25, 0.75], [1], self_rank=0)
        res["avg"], res["tmp"] = vec(m), vec(tmp)
        pc.param_send(m, [1])
    else:
        pc.param_send(m, [0])
        pc.param_recv(m, 0)
        res["recv"] = vec(m)
        with torch.no_grad():
            res["recv_fwd"] = m.forward(probe).cpu()
    assert all(p.is_cuda for p in m.parameters())
    torch.save(res, os.path.join(out_dir, "pc%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_param_com_primitives_on_device_modules(tmp_path):
    """param_com.py:13-54 on device-resident ProposalNetworks (two ranks): values as the reference's per-tensor calls give them, and the
    modules' HIP forward follows the new parameters (the packed-weight cache is dropped by every primitive)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    mp.spawn(_pc_worker, args=(2, 29571, str(tmp_path)), nprocs=2, join=True)
    out = [torch.load(str(tmp_path / ("pc%d.pt" % r))) for r in range(2)]
    a, b = out[0]["own"], out[1]["own"]
    close = lambda x, y: (x - y).abs().max().item() <= 1e-6 * max(1.0, y.abs().max().item())
    for r in range(2):
        assert torch.equal(out[r]["bcast"], b)
        assert close(out[r]["allred"], a + b)
    assert torch.equal(out[0]["bcast_fwd"], out[1]["bcast_fwd"]) and not torch.equal(out[0]["bcast_fwd"], out[0]["fwd_before"])
    assert close(out[0]["reduce"], 0.25 * a + 0.75 * b) and close(out[1]["reduce"], 0.75 * b)
    avg = 0.25 * a + 0.75 * b
    assert close(out[0]["avg"], avg) and torch.equal(out[0]["tmp"], b) and close(out[1]["recv"], avg)
    import nerf_amd
    from nerf_amd.addtional import ProposalNetwork
    nerf_amd.set_precision("fp32")
    ref = ProposalNetwork(10, 256).cuda().eval()
    off = 0
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(out[1]["recv"][off: off + p.numel()].view_as(p)); off += p.numel()
        probe = torch.rand(7, 5, 3, generator=torch.Generator().manual_seed(3)).cuda()
        assert torch.equal(ref.forward(probe).cpu(), out[1]["recv_fwd"])


def _rccl_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))       # ddp_train.py:67 (RCCL on ROCm)
    from oracle import nerf_oracle as O
    import nerf_amd
    from nerf_amd import parallel
    from nerf_amd.optim import Adam
    from nerf_amd.training import TrainStep
    nerf_amd.set_precision("bf16")
    res = {"backend": dist.get_backend()}
    t = torch.arange(8, dtype=torch.float32, device="cuda")
    dist.all_reduce(t)                                                    # the collective itself runs
    res["allreduce_ok"] = bool(torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32) * world))
    gen = torch.Generator().manual_seed(3)
    img = torch.rand(3, 40, 40, generator=gen).cuda()
    pose = O.pose_spherical(20.0, -30.0, 4.0)[:3].contiguous().cuda()
    focal = O.fov2focal(0.6911112070083618, (40, 40))

    def run(graphed, use_flat):
        prop, mip = _nets()
        parallel.broadcast_parameters([mip, prop], src=0)
        opt = Adam(list(mip.parameters()) + list(prop.parameters()), lr=1e-3, lr_on_device=True)
        flat = parallel.FlatGradients([mip, prop], opt) if use_flat else None
        step = TrainStep(prop, mip, opt, (40, 40), focal, NEAR, FAR, ray_num=128, coarse_pnum=32, fine_pnum=64, seed=4321, flat_grads=flat)
        step.set_image(img, pose)
        if graphed:
            step.capture(warmup=2)                                        # records the RCCL all-reduce into the hipGraph
        losses = [float(step()[0].item()) for _ in range(6 if graphed else 8)]
        torch.cuda.synchronize()
        return [p.detach().cpu() for p in list(mip.parameters()) + list(prop.parameters())], losses
    p_plain, l_plain = run(False, False)
    p_eager, l_eager = run(False, True)
    p_graph, l_graph = run(True, True)
    res.update(l_plain=l_plain, l_eager=l_eager, l_graph=l_graph,
               eager_vs_plain=max((a - b).abs().max().item() for a, b in zip(p_eager, p_plain)),
               graph_vs_eager=max((a - b).abs().max().item() for a, b in zip(p_graph, p_eager)))
    if rank == 0:
        torch.save(res, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_process_group_flat_allreduce_eager_and_captured(tmp_path):
    """backend "nccl" = RCCL, as ddp_train.py:67 initialises it: the flat gradient all-reduce runs eagerly and is CAPTURED inside the
    training iteration's hipGraph (TrainStep(flat_grads=...).capture()); replays equal eager iterations, and -- one rank: the mean
    over the ranks is the identity -- both equal the iteration without any collective."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    out_path = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_worker, args=(1, 29581, out_path), nprocs=1, join=True)
    got = torch.load(out_path)
    assert got["backend"] == "nccl" and got["allreduce_ok"]
    assert got["eager_vs_plain"] == 0.0 and got["l_eager"] == got["l_plain"]
    assert got["graph_vs_eager"] <= 1e-5
    assert all(abs(a - b) <= 1e-5 * max(1.0, abs(b)) for a, b in zip(got["l_graph"], got["l_eager"][2:]))


def test_detached_flat_gradients_owner_is_dead():
    """a superseded FlatGradients must not keep working silently.  A newer owner detaches the old one; the old one then refuses
    bind / begin_step / finalize_window / all_reduce, a TrainStep refuses to be built on it, and the modules' kernels write into the NEW
    buffer (whose gradients equal ordinary autograd's)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import nerf_amd
    from nerf_amd import parallel
    from nerf_amd.optim import Adam
    from nerf_amd.training import TrainStep
    nerf_amd.set_precision("fp32")
    prop, mip = _nets()
    opt = Adam(list(mip.parameters()) + list(prop.parameters()), lr=1e-3, lr_on_device=True)
    old = parallel.FlatGradients([mip, prop], opt)
    new = parallel.FlatGradients([mip, prop], opt)                         # detaches `old`
    assert getattr(old, "_dead", False) and not getattr(new, "_dead", False)
    for fn in (old.bind, old.begin_step, old.finalize_window, old.all_reduce):
        with pytest.raises(RuntimeError):
            fn()
    with pytest.raises(ValueError):
        TrainStep(prop, mip, opt, (40, 40), 50.0, NEAR, FAR, ray_num=64, coarse_pnum=C_TRAIN, fine_pnum=F_TRAIN, flat_grads=old)
    _loss(prop, mip, *_inputs(0)).backward()
    assert all(p.grad is new.views[p] for p in new.params)
    prop2, mip2 = _nets()
    _loss(prop2, mip2, *_inputs(0)).backward()
    assert torch.equal(new.flat, torch.cat([p.grad.reshape(-1) for p in list(mip2.parameters()) + list(prop2.parameters())]))
    assert float(new.flat.abs().max()) > 0.0


def test_persistent_buffer_arena_overlapping_forwards_and_growth():
    """the persistent dump arena (ops.leased / ops.scratch) sits on the training-correctness path.  Two training forwards BEFORE
    either backward -- same network twice, and two modules of the same network id -- must not share a dump: the second forward gets a
    fresh allocation.  Gradients equal the NERF_AMD_PERSISTENT_BUFFERS=0 ones bit for bit; the arena grows between steps of different
    sizes; release_buffers() between a forward and its backward neither corrupts that backward nor lets the old lease free a newer one."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gc
    import nerf_amd
    from nerf_amd import ops
    from nerf_amd.nerf_base import NeRF
    nerf_amd.set_precision("bf16")

    def grads_of(n_rays_a, n_rays_b, two_modules, release_between=False):
        prop, mip = _nets()
        mipb = _nets()[1] if two_modules else mip
        g = torch.Generator().manual_seed(5)
        outs = []
        for net, n in ((mip, n_rays_a), (mipb, n_rays_b)):
            pts = torch.cat((torch.rand(n, 16, 3, generator=g) * 2 - 1, torch.nn.functional.normalize(torch.randn(n, 16, 3, generator=g), dim=-1)), -1).cuda()
            outs.append(net.forward(pts))                                  # training forward: leases the activation dump
            if release_between:
                ops.release_buffers()
        tgt = [torch.rand(o.shape, generator=g).cuda() for o in outs]
        # backward in forward order: the FIRST forward's dump must still be intact after the second forward ran
        for o, t in zip(outs, tgt):
            ((o - t) ** 2).sum().backward()
        return [p.grad.clone() for p in mip.parameters()] + ([p.grad.clone() for p in mipb.parameters()] if two_modules else [])

    try:
        for two in (False, True):
            ops.set_persistent_buffers(True)
            ops.release_buffers()
            stats0 = dict(ops.ARENA_STATS)
            a = grads_of(40, 40, two)
            assert ops.ARENA_STATS["persistent"] > stats0["persistent"] and ops.ARENA_STATS["fresh"] > stats0["fresh"], ops.ARENA_STATS
            grown0 = ops.ARENA_STATS["grown"]
            a_big = grads_of(90, 70, two)                                  # a larger step: the arena grows ...
            assert ops.ARENA_STATS["grown"] > grown0
            a_again = grads_of(40, 40, two)                                # ... and a smaller one afterwards reuses it
            c = grads_of(40, 40, two, release_between=True)
            gc.collect()
            d = grads_of(40, 40, two)                                      # leases freed after release_buffers() must not un-busy newer ones
            ops.set_persistent_buffers(False)
            b, b_big = grads_of(40, 40, two), grads_of(90, 70, two)
            for x, y in zip(a, b):
                assert torch.equal(x, y)
            for x, y in zip(a_big, b_big):
                assert torch.equal(x, y)
            for other in (a_again, c, d):
                for x, y in zip(other, b):
                    assert torch.equal(x, y)
            assert any(float(x.abs().max()) > 0 for x in a)
    finally:
        ops.set_persistent_buffers(True)
        nerf_amd.set_precision("fp32")
