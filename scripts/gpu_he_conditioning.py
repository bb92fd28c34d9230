#!/usr/bin/env python3
"""How well-conditioned is the 1e-4 gate on the 'he' stress weights?  Renders the same rays with (a) the HIP fp32 path, (b) the CPU
oracle in fp32 (= the reference's arithmetic), (c) the CPU oracle in fp64 (the exact value of the same expressions), and prints the
three pairwise max errors for RGB / depth / weights."""
import sys

import torch

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import weights as W
from oracle import nerf_oracle as O
from nerf_amd import ops
from nerf_amd.addtional import ProposalNetwork
from nerf_amd.mip_model import MipNeRF

NEAR, FAR = 2.0, 6.0


def rays_u(n, n_fine, seed):
    gen = torch.Generator().manual_seed(seed)
    pose = O.pose_spherical(30.0, -30.0, 4.0)[:3]
    dirs = O.ray_dirs_image(pose, 64, 64, O.fov2focal(0.69, (64, 64))).reshape(-1, 3)
    pick = torch.randperm(dirs.shape[0], generator=gen)[:n]
    rays = torch.cat((pose[:, -1].expand(n, -1), dirs[pick]), -1)
    return rays, torch.rand(n, 64, generator=gen), torch.rand(n, n_fine + 1, generator=gen)


def main():
    for tag in ("small", "he"):
        prop, mip = ProposalNetwork(10, 256), MipNeRF(10, 4, 256)
        psd, msd = W.proposal_state(tag), W.mip_state(tag)
        prop.load_state_dict(psd); mip.load_state_dict(msd)
        prop, mip = prop.cuda().eval(), mip.cuda().eval()
        rays, u1, u2 = rays_u(300, 128, 17)
        with torch.no_grad():
            r32 = O.render_rays(psd, msd, rays, u1, u2, NEAR, FAR, 128, white_bkg=True)
            d = lambda sd: {k: v.double() for k, v in sd.items()}
            r64 = O.render_rays(d(psd), d(msd), rays.double(), u1.double(), u2.double(), NEAR, FAR, 128, white_bkg=True)
        z_base = torch.linspace(NEAR, FAR, 64).cuda()
        rgb, depth, w, _ = ops.render_rays(prop.packed(ops.F32), mip.packed(ops.F32), ops.F32, rays.cuda(), z_base, u1.cuda(), u2.cuda(), 128,
                                           NEAR, FAR, True, want_depth=True, want_weights=True)
        gpu = (rgb.cpu(), w.cpu(), depth.cpu())
        for name, i in (("rgb", 0), ("weights", 1), ("depth", 2)):
            e = lambda a, b: (a.double() - b.double()).abs().max().item()
            print("%-5s %-7s  |hip - ref32| %.2e   |ref32 - exact| %.2e   |hip - exact| %.2e" % (tag, name, e(gpu[i], r32[i]), e(r32[i], r64[i]), e(gpu[i], r64[i])))


if __name__ == "__main__":
    main()
