#!/usr/bin/env python3
"""Achieved HBM bandwidth of the stand-alone encode / composite kernels at the bench scale (north_star: "rocprof showing achieved HBM GB/s
on the encode/composite kernels"): positional_encoding (row 3), ipe_feature (row 12), composite (row 10) on 640 000 rays x 128 samples.
Algorithmic bytes per sample: PE 12 in + 240 out; IPE 4 (+ 24 per ray) in + 240 + 12 + 4 out; composite 16 + 4 in, 4 out (+ 40 per ray).
Prints one line per kernel (HIP-event time over `iters` launches); run it under `rocprofv3 --kernel-trace --stats` for the committed table."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import ops

N, S, L = 640000, 128, 10
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
o = torch.tensor([0.0, 0.0, 4.0], device=dev).expand(N, 3)
d = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g) * 0.3 + torch.tensor([0.0, 0.0, -1.0], device=dev), dim=-1)
rays = torch.cat((o, d), -1).contiguous()
z = torch.sort(2.0 + 4.0 * torch.rand(N, S + 1, device=dev, generator=g), dim=-1)[0].contiguous()
pts = (rays[:, None, :3] + rays[:, None, 3:] * z[:, :S, None]).contiguous()
rgbo = torch.rand(N, S, 4, device=dev, generator=g)
dn = ops.dirs_norm(rays)


def timed(fn):
    w1, w2 = fn(), fn()                              # two live outputs: the caching allocator then owns both blocks the timed loop alternates between
    del w1, w2
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        out = fn()
    b.record(); torch.cuda.synchronize()
    del out
    return a.elapsed_time(b) / iters * 1e-3


M = N * S
for name, fn, nbytes in (
        ("positional_encoding (pe_kernel)", lambda: ops.positional_encoding(pts, L), M * (12 + 24 * L)),
        ("ipe_feature (ipe_feature_kernel)", lambda: ops.ipe_feature(z, rays, L, 2.0 / (12 ** 0.5) / 1111.1, dn), M * (4 + 24 * L + 12 + 4) + N * 28),
        ("composite (composite_kernel)", lambda: ops.composite(rgbo, z[:, :S].contiguous(), rays[:, 3:].contiguous(), True, True, ops.ACT_RELU, (2.0, 6.0)), M * 24 + N * 40)):
    t = timed(fn)
    print("%-36s %8.3f ms   %6.2f GB   %6.2f TB/s   (%.2f of 8 TB/s)" % (name, t * 1e3, nbytes / 1e9, nbytes / t / 1e12, nbytes / t / 8e12))

# the other stand-alone rows at the same scale (API-parity entry points; the render path uses the fused forms)
w64 = torch.rand(N, 64, device=dev, generator=g)
z64 = torch.sort(2.0 + 4.0 * torch.rand(N, 64, device=dev, generator=g), dim=-1)[0].contiguous()
sig = torch.randn(N, S, device=dev, generator=g)
zS = z[:, :S].contiguous()
u129 = torch.rand(N, 129, device=dev, generator=g)
below = torch.randint(0, 62, (N, 128), device=dev, generator=g)
for name, fn, nbytes in (
        ("length2pts", lambda: ops.length2pts(rays, zS), M * (4 + 24) + N * 24),
        ("sigma_to_weights", lambda: ops.sigma_to_weights(sig, zS, rays[:, 3:].contiguous(), ops.ACT_RELU), M * 12 + N * 12),
        ("max_blur", lambda: ops.max_blur(w64, 0.01), N * 64 * 8),
        ("inverse_sample (64 bins -> 129, sorted)", lambda: ops.inverse_sample(w64, z64, u129, True, want_below=True), N * (64 * 8 + 129 * 4 + 129 * 12)),
        ("get_bounds", lambda: ops.get_bounds(w64[:, :63].contiguous(), below), N * (63 * 4 + 128 * 8 + 128 * 4))):
    t = timed(fn)
    print("%-40s %8.3f ms   %6.2f GB   %6.2f TB/s" % (name, t * 1e3, nbytes / 1e9, nbytes / t / 1e12))
