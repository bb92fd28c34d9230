"""SURVEY.md section 8e on real hardware: two processes (one torch.distributed rank each, gloo rendezvous, both on the box's single
GPU) render the two halves of an image through the HIP path and gather it; the result must equal the same shards rendered in one
process.  Also the flat-buffer gradient all-reduce on device tensors."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, SAMPLES, NEAR, FAR = 64, 128, 2.0, 6.0


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
    if rank == 0:
        torch.save({"rgb": img["rgb"].cpu(), "depth": img["depth_img"].cpu(), "n_reduced": n, "grads_ok": ok}, out_path)
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
    # the same two shards in this process
    sys.path.insert(0, ROOT)
    import nerf_amd
    from nerf_amd import ops, parallel
    from nerf_amd.procedures import RENDER_COARSE_PNUM
    nerf_amd.set_precision("fp32")
    prop, mip = _nets()
    pose, focal = _pose_focal()
    parts_rgb, parts_depth = [], []
    with torch.no_grad():
        for r in range(2):
            start, end = parallel.shard_range(H * H, r, 2, align=256)
            fx, fy = (float(focal[1]), float(focal[0])) if isinstance(focal, (tuple, list)) else (float(focal), float(focal))
            rays = ops.generate_rays(pose, H, H, fx, fy, pose.device, start, end - start)
            g = torch.Generator(device=pose.device).manual_seed(5 * 1000003 + r)
            u1 = torch.rand((end - start, RENDER_COARSE_PNUM), device=pose.device, generator=g)
            u2 = torch.rand((end - start, SAMPLES + 1), device=pose.device, generator=g)
            rgb, depth, _, _ = ops.render_rays(prop.packed(ops.F32), mip.packed(ops.F32), ops.F32, rays, torch.linspace(NEAR, FAR, RENDER_COARSE_PNUM).cuda(),
                                               u1, u2, SAMPLES, NEAR, FAR, True, want_depth=True)
            parts_rgb.append(rgb)
            parts_depth.append(depth)
    want = torch.cat(parts_rgb).view(H, H, 3).permute(2, 0, 1).cpu()
    assert torch.equal(got["rgb"], want)
    assert torch.equal(got["depth"][0], torch.cat(parts_depth).view(H, H).cpu())
