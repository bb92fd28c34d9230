#!/usr/bin/env python3
"""Rate of the DROP-IN call: nerf_amd.procedures.render_image(network, prop_net, pose, 800, focal, near, far, 128, ...) with its default
in-kernel uniforms, timed wall-clock around the Python call (ray table, tile reorder, four launches, image assembly), bf16."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import nerf_amd
from nerf_amd.addtional import ProposalNetwork
from nerf_amd.mip_model import MipNeRF
from nerf_amd.procedures import render_image
from nerf_amd.utils import fov2Focal, pose_spherical
import weights as W

nerf_amd.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
prop, mip = ProposalNetwork(10, 256), MipNeRF(10, 4, 256)
prop.load_state_dict(W.proposal_state("small")); mip.load_state_dict(W.mip_state("small"))
prop, mip = prop.cuda().eval(), mip.cuda().eval()
focal = fov2Focal(0.6911112070083618, (800, 800))
poses = [pose_spherical(float(a), -30.0, 4.0)[:3].cuda() for a in range(0, 360, 36)]
with torch.no_grad():
    for p in poses[:2]:
        render_image(mip, prop, p, 800, focal, 2.0, 6.0, 128, white_bkg=True, render_depth=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in poses:
        out = render_image(mip, prop, p, 800, focal, 2.0, 6.0, 128, white_bkg=True, render_depth=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(poses)
print("render_image 800x800, 64+128 samples, rng=philox: %.2f ms per image = %.2f M rays/s (wall clock around the Python call)" % (dt * 1e3, 0.64 / dt))
