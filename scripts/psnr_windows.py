#!/usr/bin/env python3
"""PSNR at equal iterations WITHOUT the chaos: teacher-forced 250-iteration windows.

A free-running comparison of two training paths measures, after a few thousand iterations, which basin each trajectory fell into -- a
per-seed standard deviation of 0.5-3 dB that no affordable number of seeds averages down to 0.1 dB.  Here every window starts from the
CPU oracle's OWN state at a checkpoint of its run (parameters, both Adam moments, step counts, the position of the CPU generator that
draws batches and uniforms: `psnr_seeds.py --modes cpu --keep-all` keeps one every 250 iterations) and runs the next 250 iterations on

    cpu      the oracle itself, continued                          (= the reference trajectory; from the kept states, no recomputation)
    null     the oracle restarted from the state with ANOTHER thread count (the same program, another summation order: the control)
    fp32     the HIP path, fp32-MFMA kernels
    bf16     the HIP path, bf16 kernels
    fp32-native / bf16-native   the same with the package's own one-launch Adam (device-side step count and float64 learning rate) instead of torch's

on the identical draws.  Compared per window: the training PSNR of those 250 iterations (mean image loss -> dB) and the held-out render
at the window's end.  Variance between trajectories is cancelled because no trajectory is older than 250 iterations.

    psnr_windows.py --modes fp32,bf16 --states psnr_states_r06 --seeds 1,2,3,4 --at 0,500,...      (GPU box)   -> WINDOW lines
    psnr_windows.py --modes null --threads 3 --states ... --seeds 1 --at 6250                      (build container)
    psnr_windows.py --modes cpu  --states ...                                                        (reads the kept states only)
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import test_gpu_training_psnr as T

WINDOW = 250


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="fp32,bf16")
    ap.add_argument("--states", default="psnr_states_r06")
    ap.add_argument("--seeds", default="1,2,3,4")
    ap.add_argument("--at", default="all", help="window starts (iterations, multiples of 250) or 'all' = every kept state that has a successor")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--iters", type=int, default=10000)
    ap.add_argument("--len", type=int, default=WINDOW, help="window length (a multiple of 250; default 250).  --at 6000 --len 4000 = the WHOLE decay phase "
                                                             "teacher-forced once, from the oracle's state at the end of the hold phase")
    a = ap.parse_args()
    WIN = a.len
    if a.threads > 0:
        torch.set_num_threads(a.threads)
    # the recipe of profiles/r05_psnr (psnr_seeds.py --size 40 --views 25 --held 1 --rays 512 --coarse 32 --fine 64 --iters 10000 --lr-mult 3 --hold 0.6)
    T.H, T.C_N, T.F_N, T.RAYS, T.ITERS, T.N_HELD = 40, 32, 64, 512, a.iters, 1
    T.LR = 1.5e-4 * T.RAYS / 512 * 3
    T.SCHED = T.long_schedule(T.LR, a.iters, hold=0.6)
    T.CHECKPOINTS = ()
    views = T.analytic_scene(25)
    for seed in [int(s) for s in a.seeds.split(",")]:
        base = os.path.join(a.states, "cpu_seed%d.state" % seed)
        have = sorted(int(f.rsplit(".it", 1)[1]) for f in os.listdir(a.states) if f.startswith("cpu_seed%d.state.it" % seed))
        starts = [k for k in have if k + WIN in have] if a.at == "all" else [int(k) for k in a.at.split(",")]
        for k in starts:
            f0, f1 = base + ".it%05d" % k, base + ".it%05d" % (k + WIN)
            if not os.path.exists(f0):
                continue
            st = torch.load(f0, weights_only=False)
            for mode in a.modes.split(","):
                t0 = time.time()
                extra = ""
                if mode == "cpu":                                   # the oracle's own continuation: its kept histories
                    if not os.path.exists(f1):
                        continue
                    nx = torch.load(f1, weights_only=False)
                    hist, held = nx["hist"][k:k + WIN], nx["held_at"]
                elif mode == "null":
                    hist, held = T.run_oracle(views, seed, init=st, stop=k + WIN)
                    held = held[0]
                else:                                               # "fp32" / "bf16" (torch's CUDA Adam, like the oracle's torch CPU Adam) or
                    prec, native = (mode[:-7], True) if mode.endswith("-native") else (mode, False)   # "...-native": nerf_amd.optim.Adam(lr_on_device=True)
                    hist, held = T.run_hip(views, seed, prec, init=st, stop=k + WIN, native_adam=native)
                    extra = "  held-out-fp32render %.5f" % held[1] if len(held) > 1 else ""
                    held = held[0]
                assert len(hist) == WIN, len(hist)
                mse = sum(hist) / len(hist)
                print(("WINDOW" if WIN == WINDOW else "LONGWINDOW len %d" % WIN) + " mode %s seed %d start %d lr %.4e train-psnr %.5f held-out %.5f first-loss %.9e last50-psnr %.5f threads %d  %.0f s"
                      % (mode, seed, k, T.SCHED(k), T.psnr(mse), held, hist[0], T.psnr(sum(hist[-200:]) / 200) if WIN > WINDOW else T.psnr(sum(hist[-50:]) / 50), torch.get_num_threads(), time.time() - t0) + extra, flush=True)


if __name__ == "__main__":
    main()
