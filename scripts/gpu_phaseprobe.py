#!/usr/bin/env python3
"""Diagnostic (library built with -DMLP_PHASEPROBE, NERF_AMD_LIB pointing at it): shader cycles per tile phase of wave 0 of workgroup 0
of the proposal kernel at the bench shape (640 000 rays x 64 samples, bf16, in-kernel uniforms)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import nerf_amd
import nerf_amd.addtional
from nerf_amd import ops
import weights as W

nerf_amd.set_precision("bf16")
prop = nerf_amd.addtional.ProposalNetwork(10, 256).cuda()
prop.load_state_dict(W.proposal_state(sys.argv[1] if len(sys.argv) > 1 else "small"))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 640000
g = torch.Generator().manual_seed(0)
rays = torch.cat((torch.randn(N, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 4.0]), torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)), -1).cuda()
z_base = torch.linspace(2.0, 6.0, 64).cuda()
s = ops.samples_rays(rays, 64, z_base=z_base, z_jitter=4.0 / 64, seed=1234)
for it in range(3):
    out = ops.proposal_forward_samples(prop.packed(ops.BF16), ops.BF16, s, (N, 64), rays.device)
    torch.cuda.synchronize()
    ph = out.view(-1)[:10].view(torch.int64).cpu().tolist()
    tiles = max(1, (N * 64 / 256 + 255) // 256)
    tot = sum(ph)
    # ph[4] = up to "position complete" of each column tile (sample fetch: index split, ray / depth loads, Philox, o + z d); ph[0] = the rest (encoding)
    print("tiles/WG %d  cycles/tile: fetch %.0f + encode %.0f | layer0 %.0f | layers1-3 %.0f | head+store %.0f | total %.0f   (MFMA floor: layer0 %d, layers1-3 %d, head %d)"
          % (tiles, ph[4] / tiles, ph[0] / tiles, ph[1] / tiles, ph[2] / tiles, ph[3] / tiles, tot / tiles, 4 * 8 * 2 * 32, 3 * 16 * 8 * 2 * 32, 16 * 2 * 32))
