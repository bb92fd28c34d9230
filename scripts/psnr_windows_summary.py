#!/usr/bin/env python3
"""Summary of the teacher-forced PSNR windows (scripts/psnr_windows.py): paired differences against the oracle's own continuation, per phase of
the learning-rate schedule (warm-up: the window starting at 0; hold: starts 250 ... 5 750; decay: starts >= 6 000), mean +- s.e.m. over the
(seed, window) pairs, for the HIP fp32 path, the HIP bf16 path and the NULL control (the oracle itself with another thread count).
    psnr_windows_summary.py LOG [LOG ...] > profiles/r06_psnr/summary.md"""
import math
import re
import sys

rec = {}
pat = re.compile(r"WINDOW mode (\S+) seed (\d+) start (\d+) lr (\S+) train-psnr (\S+) held-out (\S+) first-loss (\S+) last50-psnr (\S+) threads (\d+)")
for f in sys.argv[1:]:
    for line in open(f):
        m = pat.search(line)
        if m:
            mode, seed, start = m.group(1), int(m.group(2)), int(m.group(3))
            rec[(mode, seed, start)] = {"lr": float(m.group(4)), "train": float(m.group(5)), "held": float(m.group(6)), "first": float(m.group(7)),
                                        "last50": float(m.group(8)), "threads": int(m.group(9))}
            m2 = re.search(r"held-out-fp32render (\S+)", line)
            if m2 and mode == "bf16":                               # bf16-TRAINED weights rendered by the fp32 kernels: its own pseudo-mode
                rec[(mode + "-train/fp32-render", seed, start)] = dict(rec[(mode, seed, start)], held=float(m2.group(1)))


# the whole decay phase teacher-forced ONCE (psnr_windows.py --at 6000 --len 4000): the oracle's state at the end of the hold phase -> 10 000
long_rec = {}
lpat = re.compile(r"LONGWINDOW len (\d+) mode (\S+) seed (\d+) start (\d+) lr (\S+) train-psnr (\S+) held-out (\S+) first-loss (\S+) last50-psnr (\S+) threads (\d+)")
for f in sys.argv[1:]:
    for line in open(f):
        m = lpat.search(line)
        if m:
            key = (m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(1)))
            long_rec[key] = {"train": float(m.group(6)), "held": float(m.group(7)), "tail": float(m.group(9))}
            m2 = re.search(r"held-out-fp32render (\S+)", line)
            if m2 and m.group(2) == "bf16":
                long_rec[(m.group(2) + "-train/fp32-render",) + key[1:]] = dict(long_rec[key], held=float(m2.group(1)))


def phase(start):
    return "warm-up (0-250)" if start == 0 else ("hold (lr 4.5e-4)" if start < 6000 else "decay (100x over 4 000 it)")


def stats(xs):
    n = len(xs)
    if n == 0:
        return float("nan"), float("nan"), float("nan"), 0
    m = sum(xs) / n
    sd = math.sqrt(sum((x - m) ** 2 for x in xs) / max(n - 1, 1))
    return m, sd / math.sqrt(n), sd, n


modes = [m for m in ("fp32", "fp32-native", "bf16", "bf16-native", "bf16-train/fp32-render", "null") if any(k[0] == m for k in rec)]
keys = sorted({(k[1], k[2]) for k in rec if k[0] == "cpu"})
seeds = sorted({k[0] for k in keys})
print("# PSNR at equal iterations, teacher-forced 250-iteration windows (round 6)\n")
print("Every window starts from the CPU oracle's OWN state at a checkpoint of its 10 000-iteration run (parameters, both Adam moments, step counts, the")
print("position of the CPU generator that draws batches and uniforms) and runs the next 250 iterations on the identical draws through: the oracle")
print("itself (`cpu`: its kept trajectory, 1 thread), the oracle restarted with another thread count (`null`: the control -- same program, another")
print("summation order), the HIP fp32 path and the HIP bf16 path.  Figures: training PSNR of those 250 iterations (mean image loss -> dB) and the held-out")
print("render at the window's end, as PAIRED differences against `cpu`.  Recipe: `scripts/psnr_seeds.py --size 40 --views 25 --held 1 --rays 512 --coarse 32")
print("--fine 64 --iters 10000 --lr-mult 3 --hold 0.6` (profiles/r05_psnr), seeds %s, %d windows.\n" % (seeds, len(keys)))
print("## Per phase: paired difference to the oracle's own continuation, dB (mean +- s.e.m.; sd; n windows)\n")
print("| phase | path | training PSNR of the window | held-out PSNR at the window's end | max |difference| train / held-out |")
print("|---|---|---|---|---|")
for ph in ("warm-up (0-250)", "hold (lr 4.5e-4)", "decay (100x over 4 000 it)", "all"):
    for mode in modes:
        dt, dh = [], []
        for seed, start in keys:
            if (ph == "all" or phase(start) == ph) and (mode, seed, start) in rec:
                c, r = rec[("cpu", seed, start)], rec[(mode, seed, start)]
                dt.append(r["train"] - c["train"])
                dh.append(r["held"] - c["held"])
        if not dt:
            continue
        mt, st, sdt, n = stats(dt)
        mh, sh, sdh, _ = stats(dh)
        print("| %s | %s | %+.4f +- %.4f (sd %.4f; n = %d) | %+.4f +- %.4f (sd %.4f) | %.3f / %.3f |"
              % (ph, mode, mt, st, sdt, n, mh, sh, sdh, max(abs(x) for x in dt), max(abs(x) for x in dh)))
print("\n## Every window\n")
print("| seed | start | lr at start | oracle train dB | oracle held-out dB | " + " | ".join("%s - oracle, train / held-out" % m for m in modes) + " |")
print("|---|---|---|---|---|" + "---|" * len(modes))
for seed, start in keys:
    c = rec[("cpu", seed, start)]
    cells = []
    for mode in modes:
        r = rec.get((mode, seed, start))
        cells.append("%+.4f / %+.4f" % (r["train"] - c["train"], r["held"] - c["held"]) if r else "-")
    print("| %d | %d | %.3e | %.3f | %.3f | %s |" % (seed, start, c["lr"], c["train"], c["held"], " | ".join(cells)))
# first-iteration agreement: the window's first loss is computed from the IDENTICAL state on the identical batch
print("\n## First iteration of every window (identical state, identical batch): relative difference of the image loss\n")
for mode in modes:
    xs = [abs(rec[(mode, s, k)]["first"] - rec[("cpu", s, k)]["first"]) / rec[("cpu", s, k)]["first"] for s, k in keys if (mode, s, k) in rec]
    if xs:
        print("* %s: max %.2e, mean %.2e over %d windows" % (mode, max(xs), sum(xs) / len(xs), len(xs)))

if long_rec:
    print("\n## The whole decay phase in ONE window: the oracle's state at iteration 6 000 -> 10 000 on each path\n")
    print("End-of-run figures of a path that inherits the oracle's basin at the end of the hold phase and runs the 4 000 decay iterations itself, against the")
    print("oracle's own end of run: training PSNR of the last 200 iterations and the held-out render at iteration 10 000 (dB, path - oracle).\n")
    lkeys = sorted({k[1:] for k in long_rec if k[0] == "cpu"})
    lmodes = [m for m in ("fp32", "fp32-native", "bf16", "bf16-native", "bf16-train/fp32-render", "null") if any(k[0] == m for k in long_rec)]
    print("| seed | window | oracle train (last 200) | oracle held-out | " + " | ".join("%s - oracle: train (last 200) / held-out" % m for m in lmodes) + " |")
    print("|---|---|---|---|" + "---|" * len(lmodes))
    acc = {m: ([], []) for m in lmodes}
    for k in lkeys:
        c = long_rec[("cpu",) + k]
        cells = []
        for m in lmodes:
            r = long_rec.get((m,) + k)
            if r:
                acc[m][0].append(r["tail"] - c["tail"]); acc[m][1].append(r["held"] - c["held"])
                cells.append("%+.4f / %+.4f" % (r["tail"] - c["tail"], r["held"] - c["held"]))
            else:
                cells.append("-")
        print("| %d | %d -> %d | %.3f | %.3f | %s |" % (k[0], k[1], k[1] + k[2], c["tail"], c["held"], " | ".join(cells)))
    for m in lmodes:
        if acc[m][0]:
            mt, st, sdt, n = stats(acc[m][0])
            mh, sh, sdh, _ = stats(acc[m][1])
            print("\n* %s - oracle over %d seeds: training PSNR (last 200 iterations) %+.4f +- %.4f dB (sd %.4f), held-out %+.4f +- %.4f dB (sd %.4f)" % (m, n, mt, st, sdt, mh, sh, sdh))

# The free-running NULL: the same oracle, the same seeds, run twice (round 5: profiles/r05_psnr/cpu_seed*.log; round 6: the trajectories the
# windows start from) with different thread configurations -- training PSNR of the last 200 iterations at equal iterations, paired per seed.
import glob
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def progress(path):
    out = {}
    for line in open(path):
        m = re.search(r"PROGRESS seed (\d+) it (\d+) train-psnr\(last 200\) (\S+)", line)
        if m:
            out[int(m.group(2))] = float(m.group(3))
    return out


a_dir, b_dir = os.path.join(ROOT, "profiles", "r05_psnr"), os.path.join(ROOT, "profiles", "r06_psnr")
pairs = []
for fb in sorted(glob.glob(os.path.join(b_dir, "cpu_seed*.log"))):
    fa = os.path.join(a_dir, os.path.basename(fb))
    if os.path.exists(fa):
        pairs.append((os.path.basename(fb), progress(fa), progress(fb)))
if pairs:
    print("\n## The free-running null: the oracle against ITSELF\n")
    print("The CPU oracle ran the same seeds in round 5 (`profiles/r05_psnr/cpu_seed*.log`) and again in round 6 (the trajectories the windows start from,")
    print("`profiles/r06_psnr/cpu_seed*.log`) -- the same program on the same draws with another thread configuration, i.e. another fp32 summation order.")
    print("Training PSNR of the last 200 iterations at equal iterations, round 6 - round 5, paired over the %d common seeds:\n" % len(pairs))
    print("| iteration | oracle (r06) - oracle (r05), dB: mean +- s.e.m. (sd; per seed) |")
    print("|---|---|")
    for it in (500, 1000, 2000, 3000, 4000, 5000, 6000, 6500, 7000, 8000, 9000, 10000):
        d = [b[it] - a[it] for _, a, b in pairs if it in a and it in b]
        if d:
            m, se, sd, n = stats(d)
            print("| %d | %+.3f +- %.3f (sd %.3f; %s) |" % (it, m, se, sd, " ".join("%+.2f" % x for x in d)))
    print("\nRound 5's free-running figure for the HIP fp32 path against the oracle, -0.32 +- 0.15 dB over 12 seeds at the end of the run (sd 0.51), is to be read")
    print("against THIS spread: two runs of the oracle itself differ by as much per seed once the trajectories have left each other.")
