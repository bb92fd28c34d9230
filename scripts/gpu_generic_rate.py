#!/usr/bin/env python3
"""Rates of the generic-shape path (nerf_amd/generic_path.py, nerf_amd_gemm): the layer products alone, a render_image call and a training
step with networks wider than the fused kernels' compiled shapes (`--nerf_net_width 512 --prop_net_width 512`)."""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nerf_amd
from nerf_amd import ops, procedures
from nerf_amd.addtional import ProposalLoss, ProposalNetwork, getBounds
from nerf_amd.mip_methods import maxBlurFilter
from nerf_amd.mip_model import MipNeRF
from nerf_amd.nerf_base import NeRF
from nerf_amd.utils import inverseSample, fov2Focal, pose_spherical


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n


def main():
    width = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    for prec, code in (("bf16", ops.BF16), ("fp32", ops.F32)):
        M, N, K = 262144, width, width
        x, w, b, dy = torch.randn(M, K).cuda(), (torch.randn(N, K) * 0.05).cuda(), torch.randn(N).cuda(), torch.randn(M, N).cuda()
        t1 = timed(lambda: ops.gemm(code, x, w.t(), bias=b, act=1))
        t2 = timed(lambda: ops.gemm(code, dy, w, mask=x))
        t3 = timed(lambda: ops.gemm(code, dy.t(), x))
        fl = 2.0 * M * N * K / 1e12
        print("gemm %s  M %d N %d K %d: forward %.3f ms = %.0f TFLOP/s | input grad %.3f ms = %.0f | weight grad %.3f ms = %.0f"
              % (prec, M, N, K, t1 * 1e3, fl / t1, t2 * 1e3, fl / t2, t3 * 1e3, fl / t3), flush=True)
    nerf_amd.set_precision("bf16")
    torch.manual_seed(0)
    prop, mip = ProposalNetwork(10, width).cuda().eval(), MipNeRF(10, 4, width).cuda().eval()
    pose = pose_spherical(30.0, -30.0, 4.0)[:3].cuda()
    size = 400
    focal = fov2Focal(0.6911112070083618, (size, size))
    with torch.no_grad():
        t = timed(lambda: procedures.render_image(mip, prop, pose, size, focal, 2.0, 6.0, 128, white_bkg=True), n=3, warm=1)
    print("render_image %dx%d, 64+128 samples, width %d (bf16): %.1f ms = %.0f k rays/s" % (size, size, width, t * 1e3, size * size / t / 1e3), flush=True)
    prop.train(); mip.train()
    from nerf_amd.optim import Adam
    opt = Adam(list(mip.parameters()) + list(prop.parameters()), lr=1e-4)
    n_rays = 4096
    rays = torch.cat((torch.tensor([0.0, 0.0, 4.0]).expand(n_rays, 3), F.normalize(torch.randn(n_rays, 3) * 0.2 + torch.tensor([0.0, 0.0, -1.0]), dim=-1)), -1).cuda().contiguous()
    tgt = torch.rand(n_rays, 3).cuda()
    base = torch.linspace(2.0, 6.0 - 4.0 / 64, 64).cuda()
    ploss = ProposalLoss()

    def step():
        z_c = base + torch.rand((n_rays, 64), device="cuda") * (4.0 / 64)
        pts = (rays[:, None, :3] + rays[:, None, 3:] * z_c[:, :, None]).contiguous()
        pw = maxBlurFilter(ProposalNetwork.get_weights(F.softplus(prop.forward(pts)), z_c, rays[:, 3:]), 0.01)
        z_f, below = inverseSample(pw, z_c, 129, sort=True, u=torch.rand((n_rays, 129), device="cuda"))
        z_f = z_f[..., :-1].contiguous()
        rend, wts, _ = NeRF.render(mip.forward(NeRF.length2pts(rays, z_f)), z_f, rays[:, 3:], white_bkg=True)
        loss = ploss(getBounds(pw, below), wts.detach()) + torch.mean((rend - tgt) ** 2)
        opt.zero_grad()
        loss.backward()
        opt.step()
    t = timed(step, n=5, warm=2)
    print("training step %d rays, 64+128 samples, width %d (bf16): %.1f ms = %.0f k rays/s" % (n_rays, width, t * 1e3, n_rays / t / 1e3), flush=True)


if __name__ == "__main__":
    main()
