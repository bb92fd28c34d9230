#!/usr/bin/env python3
"""PSNR at equal iterations, long form (north_star): the synthetic analytic scene of tests/test_gpu_training_psnr.py
trained for thousands of iterations -- until the held-out view is actually learnt -- through the CPU oracle (torch autograd, fp32),
the HIP fp32 path and the HIP bf16 path, with identical initial weights, batches and uniforms (one seeded CPU generator in the
reference's draw order).  Prints the held-out PSNR of single renders at fixed checkpoints.

usage: gpu_psnr_long.py ITERS [modes=cpu,fp32,bf16] [seed=7] [rays=256] [views=8] [lr multiplier=1] [hold fraction=0]
"""
import sys
import time

import torch

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import test_gpu_training_psnr as T


def main():
    iters = int(sys.argv[1])
    modes = (sys.argv[2] if len(sys.argv) > 2 else "cpu,fp32,bf16").split(",")
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    if len(sys.argv) > 4:
        T.RAYS = int(sys.argv[4])
        T.LR = 1.5e-4 * T.RAYS / 512
    n_views = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    if len(sys.argv) > 6:
        T.LR *= float(sys.argv[6])                           # base-rate multiplier on top of the reference's rule
    hold = float(sys.argv[7]) if len(sys.argv) > 7 else 0.0  # fraction of the run at the full rate before the decay starts
    T.ITERS = iters
    step = max(iters // 40, 1)
    T.CHECKPOINTS = tuple(range(step, iters + 1, step))
    # the reference's schedule shape (nerf_base.DecayLrScheduler, train.py:133) compressed to this run: tests' long_schedule()
    sched = T.long_schedule(T.LR, iters, hold=hold) if hold > 0 else T.long_schedule(T.LR, iters, hold=0.0, decay_r=0.1 ** 0.5)
    T.SCHED = sched
    torch.set_num_threads(min(32, torch.get_num_threads()))
    views = T.analytic_scene(n_views)
    print("iters %d  rays %d  lr %.2e  seed %d  views %d  checkpoints %s" % (iters, T.RAYS, T.LR, seed, n_views, T.CHECKPOINTS), flush=True)
    for m in modes:
        t0 = time.time()
        hist, held = T.run_oracle(views, seed) if m == "cpu" else T.run_hip(views, seed, m)
        tail = T.psnr(sum(hist[-100:]) / 100)
        last = held[-4:]
        print("%-5s seed %d held-out dB every %d it: %s" % (m, seed, step, " ".join("%.2f" % v for v in held)), flush=True)
        print("%-5s seed %d FINAL held-out (mean of the last %d renders) %.3f dB | end-of-run render %.3f dB | train PSNR (last 100 it) %.3f dB | %.0f s"
              % (m, seed, len(last), sum(last) / len(last), held[-1], tail, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
