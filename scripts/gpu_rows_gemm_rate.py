#!/usr/bin/env python3
"""nerf_amd_rows_gemm (nerf_amd/csrc/rows_gemm_kernels.hip): a correctness sweep against fp64 on the bf16-rounded operands (ragged sizes,
heads, column-range views) and its rate beside nerf_amd_gemm(bf16) at the review's shape 262 144 x 512 x 512."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_amd import ops


def timed(fn, n=20, warm=5):
    """seconds per call on the device (events around n back-to-back launches: the host side of a ctypes call is ~ 30 us, a kernel here 80-250)"""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n / 1e3


def check(M, N, K, act, out_dtype, col0=0, seed=0):
    g = torch.Generator().manual_seed(seed)
    x32 = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    buf = torch.full((M, col0 + ops._pad(K, 8) + 8), 7.0, dtype=torch.bfloat16, device="cuda")          # finite junk around the view
    ops.rows_to_bf16(x32.cuda(), buf, col0)
    x = buf[:, col0:col0 + K]
    layer = ops.PackedLinear(w.cuda(), b.cuda())
    got = ops.rows_gemm(x, layer, act, out_dtype=out_dtype).float().cpu().double()
    xr, wr = x32.bfloat16().double(), w.bfloat16().double()
    want = xr @ wr.t() + b.double()
    if act == 1:
        want = want.clamp(min=0)
    elif act == 2:
        want = torch.sigmoid(want)
    tol = 1e-5 if out_dtype == torch.float32 else 4e-3            # bf16 output: half an ulp = 2^-9 relative
    err = ((got - want).abs() / (1.0 + want.abs())).max().item()
    print("check M %6d N %4d K %4d act %d out %-8s col0 %3d: max err %.2e %s" % (M, N, K, act, str(out_dtype).split(".")[1], col0, err, "ok" if err <= tol else "FAIL"), flush=True)
    return err <= tol


def main():
    if "--pmc" in sys.argv:                              # a few launches of the review's shape for the counter passes (scripts/gpu_rows_gemm_pmc.sh)
        M, N, K = 262144, 512, 512
        xb = torch.empty((M, K), dtype=torch.bfloat16, device="cuda")
        ops.rows_to_bf16(torch.randn(M, K).cuda(), xb)
        layer = ops.PackedLinear((torch.randn(N, K) * 0.05).cuda(), torch.randn(N).cuda())
        out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        for _ in range(5):
            ops.rows_gemm(xb, layer, 1, out=out)
        torch.cuda.synchronize()
        return
    ok = True
    for M, N, K, act, dt, c0 in () if "--rates-only" in sys.argv else ((1000, 512, 512, 1, torch.bfloat16, 0), (257, 320, 63, 1, torch.bfloat16, 0), (4099, 320, 383, 1, torch.bfloat16, 320),
                                 (777, 1, 320, 0, torch.float32, 0), (777, 3, 320, 2, torch.float32, 0), (513, 11, 256, 0, torch.float32, 0),
                                 (300, 256, 201, 1, torch.bfloat16, 256), (5, 128, 32, 0, torch.bfloat16, 0), (65536, 512, 575, 1, torch.bfloat16, 0),
                                 (1, 260, 9, 0, torch.float32, 8)):
        ok &= check(M, N, K, act, dt, c0)
    if not ok:
        print("rows_gemm: CHECK FAILED")
        sys.exit(1)
    for width in (512, 320, 1024, 256):
        M, N, K = 262144, width, width
        x, w, b = torch.randn(M, K).cuda(), (torch.randn(N, K) * 0.05).cuda(), torch.randn(N).cuda()
        xb = torch.empty((M, K), dtype=torch.bfloat16, device="cuda")
        ops.rows_to_bf16(x, xb)
        layer = ops.PackedLinear(w, b)
        out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        t_new = timed(lambda: ops.rows_gemm(xb, layer, 1, out=out))
        t_old = timed(lambda: ops.gemm(ops.BF16, x, w.t(), bias=b, act=1), n=10, warm=3)
        fl = 2.0 * M * N * K / 1e12
        gb = (M * K + M * N) * 2 / 1e9
        print("forward M %d N %d K %d: rows_gemm %.3f ms = %.0f TFLOP/s (%.2f TB/s of rows) | nerf_amd_gemm bf16 %.3f ms = %.0f TFLOP/s"
              % (M, N, K, t_new * 1e3, fl / t_new, gb / t_new / 1e3, t_old * 1e3, fl / t_old), flush=True)


if __name__ == "__main__":
    main()
