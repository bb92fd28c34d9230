#!/usr/bin/env python3
"""PSNR at equal iterations with an error bar (north_star): the analytic scene of tests/test_gpu_training_psnr.py
at BASELINE configs[0]'s shape (200x200 views, 32+64 samples, 256 rays per batch), trained with the reference's own learning-rate rule
and scheduler shape through the CPU oracle (torch autograd, fp32), the HIP fp32 path and the HIP bf16 path -- identical initial weights,
batches and uniforms per seed (one seeded CPU generator in the reference's draw order).  One RESULT line per (mode, seed); the summary
is mean +- s.e.m. over the seeds.

    psnr_seeds.py --modes fp32,bf16 --seeds 1,2,3,4,5,6,7,8 --iters 6000            (GPU box)
    psnr_seeds.py --modes cpu --seeds 3 --threads 1 --iters 6000 >> profiles/...log (build container; hours per seed, one process each)
"""
import argparse
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import test_gpu_training_psnr as T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="fp32,bf16")
    ap.add_argument("--seeds", default="1,2,3,4,5,6,7,8")
    ap.add_argument("--iters", type=int, default=6000)
    ap.add_argument("--rays", type=int, default=256)
    ap.add_argument("--coarse", type=int, default=32)
    ap.add_argument("--fine", type=int, default=64)
    ap.add_argument("--size", type=int, default=200, help="views are size x size")
    ap.add_argument("--views", type=int, default=50)
    ap.add_argument("--held", type=int, default=2)
    ap.add_argument("--lr-mult", type=float, default=1.0, help="on top of the reference's rule lr * rays / 512")
    ap.add_argument("--hold", type=float, default=0.5, help="fraction of the run at the full rate before the 100x decay")
    ap.add_argument("--ckpts", type=int, default=3, help="held-out renders at the end of the run (every 100 iterations), averaged")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--resume-dir", default=None, help="cpu mode: save / resume the run state under this directory")
    ap.add_argument("--keep-all", action="store_true", help="cpu mode: keep EVERY 250-iteration state (+ held-out PSNR) for scripts/psnr_windows.py")
    a = ap.parse_args()
    if a.threads > 0:
        torch.set_num_threads(a.threads)
    T.H, T.C_N, T.F_N, T.RAYS, T.ITERS, T.N_HELD = a.size, a.coarse, a.fine, a.rays, a.iters, a.held
    T.LR = 1.5e-4 * a.rays / 512 * a.lr_mult
    T.SCHED = T.long_schedule(T.LR, a.iters, hold=a.hold)
    T.CHECKPOINTS = tuple(a.iters - 100 * k for k in range(a.ckpts - 1, -1, -1))
    views = T.analytic_scene(a.views)
    print("# recipe: %dx%d views %d (held-out %d)  rays %d  samples %d+%d  iters %d  lr %.3e (x%.2f of the reference's rule) hold %.2f  checkpoints %s"
          % (a.size, a.size, a.views, a.held, a.rays, a.coarse, a.fine, a.iters, T.LR, a.lr_mult, a.hold, T.CHECKPOINTS), flush=True)
    res = {}
    for mode in a.modes.split(","):
        for seed in [int(s) for s in a.seeds.split(",")]:
            t0 = time.time()
            if mode == "cpu":
                hist, held = T.run_oracle(views, seed, None if a.resume_dir is None else os.path.join(a.resume_dir, "cpu_seed%d.state" % seed),
                                          keep_all=a.keep_all)
            elif mode == "bf16-fp8dumps":                                # bf16 arithmetic, training dumps in scaled e4m3 (nerf_amd.set_train_dumps)
                import nerf_amd
                nerf_amd.set_train_dumps("fp8")
                try:
                    hist, held = T.run_hip(views, seed, "bf16")
                finally:
                    nerf_amd.set_train_dumps("bf16")
            else:
                hist, held = T.run_hip(views, seed, mode)
            tail = T.psnr(sum(hist[-200:]) / len(hist[-200:]))
            final = sum(held) / len(held)
            res.setdefault(mode, []).append((final, tail))
            print("RESULT mode %s seed %d held-out %.4f dB (renders %s) train-tail %.4f dB  %.0f s"
                  % (mode, seed, final, " ".join("%.3f" % v for v in held), tail, time.time() - t0), flush=True)
    for mode, v in res.items():
        n = len(v)
        for k, name in ((0, "held-out"), (1, "train-tail")):
            xs = [r[k] for r in v]
            m = sum(xs) / n
            sd = math.sqrt(sum((x - m) ** 2 for x in xs) / max(n - 1, 1))
            print("SUMMARY mode %s %s: mean %.3f dB  sd %.3f  s.e.m. %.3f  (n = %d)" % (mode, name, m, sd, sd / math.sqrt(n), n), flush=True)


if __name__ == "__main__":
    main()
