#!/usr/bin/env python3
"""render_image with networks outside the compiled shapes, bf16: the bf16-rows inference route (nerf_amd_rows_gemm) beside the fp32-row route
(nerf_amd_gemm), alternated on one box -- MipNeRF / proposal at width 512 (400 x 400, 64 + 128 samples) and Ref-NeRF `--ide_level 5` beside
the fused `--ide_level 4` kernel (200 x 200), the two cases the round-4 review priced the generic path's cliff on."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf_amd
from nerf_amd import generic_path, procedures
from nerf_amd.addtional import ProposalNetwork
from nerf_amd.mip_model import MipNeRF
from nerf_amd.ref_model import RefNeRF
from nerf_amd.utils import fov2Focal, pose_spherical


def timed(fn, n=4, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n


nerf_amd.set_precision("bf16")
torch.manual_seed(0)
pose = pose_spherical(30.0, -30.0, 4.0)[:3].cuda()
cases = []
prop512, mip512 = ProposalNetwork(10, 512).cuda().eval(), MipNeRF(10, 4, 512).cuda().eval()
prop256 = ProposalNetwork(10, 256).cuda().eval()
cases.append(("MipNeRF + proposal, width 512", 400, lambda s, f: procedures.render_image(mip512, prop512, pose, s, f, 2.0, 6.0, 128, white_bkg=True)))
ref5, ref4 = RefNeRF(10, 5).cuda().eval(), RefNeRF(10, 4).cuda().eval()
cases.append(("RefNeRF ide_level 5", 200, lambda s, f: procedures.render_image(ref5, prop256, pose, s, f, 2.0, 6.0, 128, white_bkg=True, render_normal=True)))
if "--chunks" in sys.argv:                               # rays per chunk of the layer-by-layer route: do the activations of a chunk stay in the 256 MB Infinity Cache?
    with torch.no_grad():
        for name, size, fn in cases:
            focal = fov2Focal(0.6911112070083618, (size, size))
            for ch in (4096, 2048, 1024, 512, 256, 4096, 1024):
                procedures.GENERIC_CHUNK_RAYS = ch
                t = timed(lambda: fn(size, focal))
                print("%-32s %dx%d  chunk %5d rays  %7.1f ms = %6.0f k rays/s" % (name, size, size, ch, t * 1e3, size * size / t / 1e3), flush=True)
    sys.exit(0)
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
if only:                                                  # one case, bf16 rows only, a few images: the profiling entry (scripts/gpu_rows_route_prof.sh)
    name, size, fn = cases[1 if only == "ref5" else 0]
    focal = fov2Focal(0.6911112070083618, (size, size))
    with torch.no_grad():
        for _ in range(3):
            fn(size, focal)
    torch.cuda.synchronize()
    sys.exit(0)
with torch.no_grad():
    for name, size, fn in cases:
        focal = fov2Focal(0.6911112070083618, (size, size))
        for rep in range(2):
            for rows in (True, False):
                generic_path.ROWS_ROUTE = rows
                t = timed(lambda: fn(size, focal))
                print("%-32s %dx%d  %-14s %7.1f ms = %6.0f k rays/s" % (name, size, size, "bf16 rows" if rows else "fp32 rows", t * 1e3, size * size / t / 1e3), flush=True)
    generic_path.ROWS_ROUTE = True
    focal = fov2Focal(0.6911112070083618, (200, 200))
    t = timed(lambda: procedures.render_image(ref4, prop256, pose, 200, focal, 2.0, 6.0, 128, white_bkg=True, render_normal=True))
    print("%-32s 200x200  %-14s %7.1f ms = %6.0f k rays/s" % ("RefNeRF ide_level 4", "fused kernel", t * 1e3, 40000 / t / 1e3), flush=True)
