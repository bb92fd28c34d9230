#!/usr/bin/env python3
"""PSNR at EQUAL ITERATIONS with an error bar: over every seed that has the three trajectory files of scripts/psnr_trajectory.py
(profiles/r04_psnr/short6k/traj_{cpu,fp32,bf16}_seed<S>_<N>.json -- same initial weights, batches and uniforms in the three paths), the
per-seed difference of the training PSNR of each 100-iteration window, HIP fp32 - CPU oracle and HIP bf16 - CPU oracle: mean +- s.e.m. over
the seeds.  -> profiles/r04_psnr/short6k/trajectory_summary.md"""
import glob
import json
import math
import os
import re
import sys

D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles/r04_psnr/short6k")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seeds = sorted(int(re.search(r"seed(\d+)_", f).group(1)) for f in glob.glob(os.path.join(D, "traj_cpu_seed*_%d.json" % N)))
seeds = [s for s in seeds if all(os.path.exists(os.path.join(D, "traj_%s_seed%d_%d.json" % (m, s, N))) for m in ("fp32", "bf16"))]
load = lambda m, s: json.load(open(os.path.join(D, "traj_%s_seed%d_%d.json" % (m, s, N))))
runs = {s: {m: load(m, s) for m in ("cpu", "fp32", "bf16")} for s in seeds}
ps = lambda x: -10 * math.log10(x)


def stats(xs):
    n = len(xs)
    m = sum(xs) / n
    sd = math.sqrt(sum((x - m) ** 2 for x in xs) / max(n - 1, 1))
    return m, sd / math.sqrt(n), sd


out = ["# PSNR at equal iterations, paired over %d seeds (%s)" % (len(seeds), ", ".join(map(str, seeds))), "",
       "Recipe: the 6 000-iteration recipe of `summary.md` (40 x 40 views, 512 rays, 32 + 64 samples, lr x 3), first %d iterations; the three paths of a seed start from" % N,
       "identical weights and see identical batches and uniforms (`scripts/psnr_trajectory.py`; CPU oracle = torch autograd fp32).  Per seed and 100-iteration window:",
       "training PSNR = -10 log10(mean image loss of the window); the table is the per-seed DIFFERENCE to the CPU oracle, mean +- s.e.m. (sd) over the seeds.", "",
       "| iterations | CPU oracle, dB (mean over seeds) | HIP fp32 - CPU, dB | HIP bf16 - CPU, dB | HIP bf16 - HIP fp32, dB |", "|---|---|---|---|---|"]
for lo in range(0, N, 100):
    w = lambda h: ps(sum(h[lo:lo + 100]) / len(h[lo:lo + 100]))
    c = [w(runs[s]["cpu"]["loss_img"]) for s in seeds]
    f = [w(runs[s]["fp32"]["loss_img"]) - w(runs[s]["cpu"]["loss_img"]) for s in seeds]
    b = [w(runs[s]["bf16"]["loss_img"]) - w(runs[s]["cpu"]["loss_img"]) for s in seeds]
    bf = [w(runs[s]["bf16"]["loss_img"]) - w(runs[s]["fp32"]["loss_img"]) for s in seeds]
    fm, fb, bb = stats(f), stats(b), stats(bf)
    out.append("| %d-%d | %.3f | %+.3f +- %.3f (%.3f) | %+.3f +- %.3f (%.3f) | %+.3f +- %.3f (%.3f) |" % (lo, lo + 100, sum(c) / len(c), *fm, *fb, *bb))
h = {m: [runs[s][m]["held_out_psnr"][-1] for s in seeds] for m in ("cpu", "fp32", "bf16")}
out += ["", "Held-out view after iteration %d (one render per run): CPU %.2f +- %.2f dB, HIP fp32 %.2f +- %.2f, HIP bf16 %.2f +- %.2f; paired fp32 - CPU %+.2f +- %.2f, bf16 - CPU %+.2f +- %.2f."
        % (N, *stats(h["cpu"])[:2], *stats(h["fp32"])[:2], *stats(h["bf16"])[:2], *stats([a - b_ for a, b_ in zip(h["fp32"], h["cpu"])])[:2],
           *stats([a - b_ for a, b_ in zip(h["bf16"], h["cpu"])])[:2])]
first = [next((i for i in range(N) if abs(runs[s]["fp32"]["loss_img"][i] - runs[s]["cpu"]["loss_img"][i]) > 1e-3 * runs[s]["cpu"]["loss_img"][i]), N) for s in seeds]
out += ["First iteration whose HIP fp32 loss is more than 0.1 %% off the oracle's, per seed: %s." % ", ".join(map(str, first))]
open(os.path.join(D, "trajectory_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
