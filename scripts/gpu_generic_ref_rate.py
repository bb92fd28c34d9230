#!/usr/bin/env python3
"""Rates of the generic-shape Ref-NeRF stages (generic_ref_kernels.hip) and of render_image with an `ide_level 5` network (GPU box)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nerf_amd
from nerf_amd import ops, procedures
from nerf_amd.addtional import ProposalNetwork
from nerf_amd.ref_func import ide_table
from nerf_amd.ref_model import RefNeRF
from nerf_amd.utils import fov2Focal, pose_spherical


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n


M = 1 << 20
for deg in (4, 5):
    T = (1 << deg) - 1 + deg
    heads, dirs = torch.randn(M, 11).cuda(), torch.nn.functional.normalize(torch.randn(M, 3), dim=-1).cuda()
    table = ide_table(deg).cuda().contiguous()
    out = torch.empty(M, 2 * T + 1).cuda()
    d_out, g_n, d_heads = torch.randn(M, 2 * T + 1).cuda(), torch.randn(M, 3).cuda(), torch.empty(M, 11).cuda()
    t1 = timed(lambda: ops.ref_dir_inputs(heads, dirs, deg, table, out))
    t2 = timed(lambda: ops.ref_dir_inputs_backward(heads, dirs, deg, table, d_out, g_n, d_heads))
    b1, b2 = M * 4 * (11 + 3 + 2 * T + 1 + 3), M * 4 * (11 + 3 + 2 * T + 1 + 3 + 4)
    print("ref_dir_inputs level %d, %d samples: forward %.3f ms = %.0f GB/s, backward %.3f ms = %.0f GB/s" % (deg, M, t1 * 1e3, b1 / t1 / 1e9, t2 * 1e3, b2 / t2 / 1e9), flush=True)
spec = torch.rand(M, 3).cuda()
t3 = timed(lambda: ops.ref_combine(heads, spec, 0))
print("ref_combine: %.3f ms = %.0f GB/s" % (t3 * 1e3, M * 4 * (11 + 3 + 4) / t3 / 1e9))
x, d_enc = torch.randn(M, 3).cuda(), torch.randn(M, 63).cuda()
t4 = timed(lambda: ops.positional_encoding_backward(d_enc, x, 10, True))
print("positional_encoding_backward L=10: %.3f ms = %.0f GB/s" % (t4 * 1e3, M * 4 * (63 + 3 + 3) / t4 / 1e9))
nerf_amd.set_precision("bf16")
torch.manual_seed(0)
prop, ref5, ref4 = ProposalNetwork(10, 256).cuda().eval(), RefNeRF(10, 5).cuda().eval(), RefNeRF(10, 4).cuda().eval()
pose = pose_spherical(30.0, -30.0, 4.0)[:3].cuda()
size = 200
focal = fov2Focal(0.6911112070083618, (size, size))
with torch.no_grad():
    for name, net in (("ide_level 5 (generic path)", ref5), ("ide_level 4 (fused kernel)", ref4)):
        t = timed(lambda: procedures.render_image(net, prop, pose, size, focal, 2.0, 6.0, 128, white_bkg=True, render_normal=True), n=3, warm=1)
        print("render_image %dx%d, 64+128 samples, RefNeRF %s, bf16: %.1f ms = %.0f k rays/s" % (size, size, name, t * 1e3, size * size / t / 1e3), flush=True)
