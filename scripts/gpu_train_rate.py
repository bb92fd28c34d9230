#!/usr/bin/env python3
"""Training-step rate of the nerf_amd surface (HIP training forwards, hand-written backward kernels, one-launch Adam): the body of the
reference's train.py:164-199 (proposal -> weights -> blur -> inverse sampling -> fine -> composite -> losses -> backward -> Adam)
on synthetic rays, for a few batch sizes.  Prints rays/s (fwd+bwd+step)."""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nerf_amd
from nerf_amd.addtional import ProposalLoss, ProposalNetwork, getBounds
from nerf_amd.mip_methods import maxBlurFilter
from nerf_amd.mip_model import MipNeRF
from nerf_amd.nerf_base import NeRF
from nerf_amd.utils import inverseSample

NEAR, FAR = 2.0, 6.0


def make_step(n_rays, c_n, f_n, precision, graph=False, torch_adam=False, flat=True, graph_warm=3):
    """-> a callable running ONE training step (graph=True: replaying it from a hipGraph captured here after `graph_warm` eager steps).
    flat: gradients in ONE persistent flat buffer the weight-gradient kernels write into (nerf_amd.parallel.FlatGradients: what
    TrainStep / bench.py --mode train-ddp use); False = ordinary autograd accumulation into per-tensor gradients (what train.py does)."""
    nerf_amd.set_precision(precision)
    torch.manual_seed(0)
    prop, mip = ProposalNetwork(10, 256).cuda().train(), MipNeRF(10, 4, 256).cuda().train()
    from nerf_amd.optim import Adam                        # one-launch Adam (nerf_amd_adam_step); torch_adam=True: torch.optim.Adam
    params = list(mip.parameters()) + list(prop.parameters())
    opt = torch.optim.Adam(params, lr=1e-4, capturable=graph) if torch_adam else Adam(params, lr=1e-4)
    fg = None
    if flat:
        from nerf_amd.parallel import FlatGradients
        fg = FlatGradients([mip, prop], opt)
    o = torch.tensor([0.0, 0.0, 4.0]).expand(n_rays, 3)
    d = F.normalize(torch.randn(n_rays, 3) * 0.2 + torch.tensor([0.0, 0.0, -1.0]), dim=-1)
    rays = torch.cat((o, d), -1).cuda().contiguous()
    tgt = torch.rand(n_rays, 3).cuda()
    res = (FAR - NEAR) / c_n
    base = torch.linspace(NEAR, FAR - res, c_n).cuda()
    ploss = ProposalLoss()

    def step():
        z_c = base + torch.rand((n_rays, c_n), device="cuda") * res
        pts = (rays[:, None, :3] + rays[:, None, 3:] * z_c[:, :, None]).contiguous()
        dens = F.softplus(prop.forward(pts))
        pw = maxBlurFilter(ProposalNetwork.get_weights(dens, z_c, rays[:, 3:]), 0.01)
        z_f, below = inverseSample(pw, z_c, f_n + 1, sort=True, u=torch.rand((n_rays, f_n + 1), device="cuda"))
        z_f = z_f[..., :-1].contiguous()
        rgbo = mip.forward(NeRF.length2pts(rays, z_f))
        rend, wts, _ = NeRF.render(rgbo, z_f, rays[:, 3:], white_bkg=True)
        loss = ploss(getBounds(pw, below), wts.detach()) + torch.mean((rend - tgt) ** 2)
        if fg is not None:
            fg.begin_step()
        else:
            opt.zero_grad()
        loss.backward()
        opt.step()

    if not graph:
        return step
    for _ in range(graph_warm):
        step()
    torch.cuda.synchronize()
    # whole step (forward, backward, Adam, re-pack) as ONE hipGraph: the 512-ray step is launch-bound (~200 launches)
    g = torch.cuda.CUDAGraph()
    if fg is None:
        opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    keep = (step, prop, mip, opt, fg)                       # (the graph replays into these objects' memory)
    return lambda: (g.replay(), keep)[0]


def run(n_rays, c_n, f_n, precision, iters=20, warm=5, quiet=False, graph=False, torch_adam=False, flat=True):
    step = make_step(n_rays, c_n, f_n, precision, graph=graph, torch_adam=torch_adam, flat=flat, graph_warm=warm)
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    if not quiet:
        print("train step  %5d rays  %3d+%3d samples  %s : %7.2f ms/iter  %9.0f rays/s" % (n_rays, c_n, f_n, precision, dt * 1e3, n_rays / dt), flush=True)
    return dt


def make_ref_step(n_rays, c_n, f_n, precision, graph=False, graph_warm=3):
    """The Ref-NeRF branch of the step with prop_normal (train.py:164-199): train-mode forward, density-gradient normals from
    RefNeRF.get_grad on both networks, normal / back-face / coarse-normal losses, backward, Adam -> a callable running one step."""
    from nerf_amd.ref_model import BackFaceLoss, RefNeRF, WeightedNormalLoss
    nerf_amd.set_precision(precision)
    torch.manual_seed(0)
    prop, net = ProposalNetwork(10, 256).cuda().train(), RefNeRF(10, 4).cuda().train()
    net.noise_rng = __import__("os").environ.get("REF_NOISE_RNG", "philox")      # A/B: "torch" = the round-4 torch.normal tensor
    from nerf_amd.optim import Adam
    opt = Adam(list(net.parameters()) + list(prop.parameters()), lr=1e-4)
    o = torch.tensor([0.0, 0.0, 4.0]).expand(n_rays, 3)
    d = F.normalize(torch.randn(n_rays, 3) * 0.2 + torch.tensor([0.0, 0.0, -1.0]), dim=-1)
    rays = torch.cat((o, d), -1).cuda().contiguous()
    tgt = torch.rand(n_rays, 3).cuda()
    res = (FAR - NEAR) / c_n
    base = torch.linspace(NEAR, FAR - res, c_n).cuda()

    def step():
        z_c = base + torch.rand((n_rays, c_n), device="cuda") * res
        pts = (rays[:, None, :3] + rays[:, None, 3:] * z_c[:, :, None]).contiguous().requires_grad_(True)
        dens = prop.forward(pts)
        coarse_grad = -RefNeRF.get_grad(dens, pts)
        dens = F.softplus(dens)
        pw = maxBlurFilter(ProposalNetwork.get_weights(dens, z_c, rays[:, 3:]), 0.01)
        fl, below = inverseSample(pw, z_c, f_n + 1, sort=True, u=torch.rand((n_rays, f_n + 1), device="cuda"))
        samples, fl, below, sort_ids = NeRF.coarseFineMerge(rays, z_c, fl, below)
        pos, dd = samples.split((3, 3), dim=-1)              # train.py:177-179: the two views, as the reference passes them
        pos.requires_grad_(True)
        rgbo, nrm = net.forward(pos, dd)
        dgrad = -RefNeRF.get_grad(rgbo[..., -1], pos)
        rgbo[..., -1] = F.softplus(rgbo[..., -1] + 0.5)
        rend, wts, _ = NeRF.render(rgbo, fl, rays[:, 3:], net.density_act)
        nl = WeightedNormalLoss()(wts, dgrad, nrm)
        bf = BackFaceLoss()(wts, nrm, dd)
        cnl = WeightedNormalLoss()(pw, RefNeRF.coarse_grad_select(dgrad, sort_ids, c_n).detach(), coarse_grad)
        loss = ProposalLoss()(getBounds(pw, below), wts.detach()) + torch.mean((rend - tgt) ** 2) + 4e-4 * (nl + 0.1 * cnl) + 0.1 * bf
        opt.zero_grad()
        loss.backward()
        opt.step()

    if not graph:
        return step
    for _ in range(graph_warm):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    keep = (step, prop, net, opt)
    return lambda: (g.replay(), keep)[0]


def run_ref(n_rays, c_n, f_n, precision, iters=10, warm=3, quiet=False, graph=False):
    step = make_ref_step(n_rays, c_n, f_n, precision, graph=graph, graph_warm=warm)
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    if not quiet:
        print("Ref-NeRF train step  %5d rays  %3d+%3d samples  %s : %7.2f ms/iter  %9.0f rays/s" % (n_rays, c_n, f_n, precision, dt * 1e3, n_rays / dt), flush=True)
    return dt


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ref":         # Ref-NeRF step: ref n_rays precision
        run_ref(int(sys.argv[2]), 64, 128, sys.argv[3], graph=len(sys.argv) > 4, iters=50 if len(sys.argv) > 4 else 10)
        sys.exit(0)
    if len(sys.argv) > 1:                                  # one configuration (for profiling): n_rays precision [graph]
        run(int(sys.argv[1]), 64, 128, sys.argv[2], iters=5 if len(sys.argv) < 4 else 50, warm=2 if len(sys.argv) < 4 else 5, graph=len(sys.argv) > 3)
        sys.exit(0)
    for prec in ("bf16", "fp32"):
        for n in (512, 4096, 16384):
            run(n, 64, 128, prec)
    run(512, 32, 64, "bf16")
