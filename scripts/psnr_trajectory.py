#!/usr/bin/env python3
"""Per-iteration training loss of ONE seed of a scripts/psnr_seeds.py recipe, written as JSON -- to lay the CPU oracle's trajectory beside
the HIP paths' iteration by iteration (identical initial weights, batches and uniforms): before the chaotic divergence sets in the two must
track each other; a systematic offset from the first iterations on would be a difference between the ALGORITHMS, not rounding.
    psnr_trajectory.py --mode cpu  --seed 1 --iters 400 --out profiles/r04_psnr/short6k/traj_cpu_seed1.json      (build container)
    psnr_trajectory.py --mode fp32 --seed 1 --iters 400 --out gpurun_out/traj_fp32_seed1.json                      (GPU box)"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import test_gpu_training_psnr as T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="cpu")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--total", type=int, default=6000, help="the recipe's length (fixes the learning-rate schedule)")
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    T.H, T.C_N, T.F_N, T.RAYS, T.ITERS, T.N_HELD = 40, 32, 64, 512, a.iters, 1
    T.LR = 1.5e-4 * 512 / 512 * 3
    T.SCHED = T.long_schedule(T.LR, a.total, hold=0.6)
    T.CHECKPOINTS = (a.iters,)
    views = T.analytic_scene(25)
    hist, held = T.run_oracle(views, a.seed) if a.mode == "cpu" else T.run_hip(views, a.seed, a.mode)
    json.dump({"mode": a.mode, "seed": a.seed, "iters": a.iters, "loss_img": hist, "held_out_psnr": held}, open(a.out, "w"))
    print("wrote %s: last loss %.6f held-out %.3f dB" % (a.out, hist[-1], held[-1]))


if __name__ == "__main__":
    main()
