#!/usr/bin/env python3
"""Iteration 0 of the PSNR recipe (seed 1) through the CPU oracle and through the HIP path in ONE process, every intermediate laid side by
side: where does the 5e-4 relative difference of the first training loss come from?  (GPU box.)"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import test_gpu_training_psnr as T
import weights as W
from oracle import nerf_oracle as O
import nerf_amd
from nerf_amd.addtional import ProposalNetwork
from nerf_amd.mip_methods import maxBlurFilter
from nerf_amd.mip_model import MipNeRF
from nerf_amd.nerf_base import NeRF
from nerf_amd.utils import inverseSample

T.H, T.C_N, T.F_N, T.RAYS = 40, 32, 64, 512
views = T.analytic_scene(25)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32"
res = (T.FAR - T.NEAR) / T.C_N
md = lambda a, b: (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()

# --- CPU oracle, iteration 0 (run_oracle's draw order)
torch.manual_seed(seed)
psd, msd = W.proposal_state("small"), W.mip_state("small")
rays_all, rgb_all = views[0]
idx = torch.randint(0, rays_all.shape[0], (T.RAYS,))
rays, tgt = rays_all[idx], rgb_all[idx]
z_c = torch.linspace(T.NEAR, T.FAR - res, T.C_N) + torch.rand((T.RAYS, T.C_N)) * res
pts = rays[:, None, :3] + rays[:, None, 3:] * z_c[:, :, None]
dens = F.softplus(O.proposal_forward(psd, pts))
pw = O.max_blur(O.sigma_to_weights(dens, z_c, rays[:, 3:]), 0.01)
u = torch.rand((T.RAYS, T.F_N + 1))
z_f, below = O.inverse_sample(pw, z_c, u, sort=True)
z_f = z_f[..., :-1]
rgbo = O.mip_forward(msd, O.length2pts(rays, z_f))
rend, wts, _ = O.composite(rgbo, z_f, rays[:, 3:], white_bkg=True)
loss_c = torch.mean((rend - tgt) ** 2)
state_after_cpu = torch.get_rng_state()

# --- HIP, iteration 0 (run_hip's draw order, modules constructed after the seed like run_hip does)
nerf_amd.set_precision(mode)
SEED_FIRST = os.environ.get("SEED_BEFORE_MODULES", "0") == "1"      # 1 = run_hip's order until the last day of round 4 (the modules' random initialisation
if SEED_FIRST:                                                      #     then consumes the generator: another stream than the oracle's)
    torch.manual_seed(seed)
prop, mip = ProposalNetwork(10, 256), MipNeRF(10, 4, 256)
prop.load_state_dict(psd); mip.load_state_dict(msd)
prop, mip = prop.cuda().train(), mip.cuda().train()
if not SEED_FIRST:
    torch.manual_seed(seed)
idx_h = torch.randint(0, rays_all.shape[0], (T.RAYS,))
print("batch indices equal:", torch.equal(idx, idx_h), " (differing entries: %d of %d)" % (int((idx != idx_h).sum()), T.RAYS))
rays_h, tgt_h = rays_all.cuda()[idx_h.cuda()].contiguous(), rgb_all.cuda()[idx_h.cuda()]
z_ch = (torch.linspace(T.NEAR, T.FAR - res, T.C_N) + torch.rand((T.RAYS, T.C_N)) * res).cuda()
print("z_coarse max|diff|:", md(z_ch, z_c))
pts_h = (rays_h[:, None, :3] + rays_h[:, None, 3:] * z_ch[:, :, None]).contiguous()
dens_h = F.softplus(prop.forward(pts_h))
print("proposal density max|diff|:", md(dens_h, dens), " scale", dens.abs().max().item())
pw_h = maxBlurFilter(ProposalNetwork.get_weights(dens_h, z_ch, rays_h[:, 3:]), 0.01)
print("blurred proposal weights max|diff|:", md(pw_h, pw))
z_fh, below_h = inverseSample(pw_h, z_ch, T.F_N + 1, sort=True)
z_fh = z_fh[..., :-1].contiguous()
print("fine depths max|diff|:", md(z_fh, z_f), " below equal:", torch.equal(below_h.cpu(), below), " rng state equal after the draws:", torch.equal(torch.get_rng_state(), state_after_cpu))
rgbo_h = mip.forward(NeRF.length2pts(rays_h, z_fh))
print("fine rgbo max|diff|:", md(rgbo_h, rgbo), " sigma range", rgbo[..., 3].min().item(), rgbo[..., 3].max().item())
rend_h, wts_h, _ = NeRF.render(rgbo_h, z_fh, rays_h[:, 3:], white_bkg=True)
print("weights max|diff|:", md(wts_h, wts), " rendered max|diff|:", md(rend_h, rend))
loss_h = torch.mean((rend_h - tgt_h) ** 2)
print("loss cpu %.8f  hip %.8f  rel %.3e" % (loss_c.item(), loss_h.item(), (loss_h.item() - loss_c.item()) / loss_c.item()))
per_ray = (rend_h.detach().cpu() - rend).abs().amax(dim=-1)
bad = torch.nonzero(per_ray > 1e-4).flatten()
print("rays whose rendered colour differs by > 1e-4: %d of %d" % (bad.numel(), T.RAYS))
for r in bad[:6].tolist():
    print("  ray %d: cpu %s hip %s | last-sample sigma cpu %.3e hip %.3e | sum w cpu %.6f hip %.6f"
          % (r, rend[r].tolist(), rend_h[r].tolist(), rgbo[r, -1, 3].item(), rgbo_h[r, -1, 3].item(), wts[r].sum().item(), wts_h[r].sum().item()))
