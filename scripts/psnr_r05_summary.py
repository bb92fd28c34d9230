#!/usr/bin/env python3
"""profiles/r05_psnr/summary.md: PSNR at equal iterations at the END of a >= 30 dB recipe, CPU oracle beside the HIP paths.

Inputs: profiles/r05_psnr/cpu_seed*.log (scripts/psnr_seeds.py --modes cpu, one process per seed in the build container, resumable) and
profiles/r05_psnr/hip_10000_seeds*.log (the same recipe and seeds on the GPU box, --modes fp32,bf16).  Both kinds carry RESULT lines (end of
run: held-out renders at the last four checkpoints, training PSNR of the last 200 iterations) and PROGRESS lines (training PSNR of the last
200 iterations every 500 iterations), so that CPU seeds that did not finish inside the session still pair with the HIP runs at every
iteration they reached.
"""
import glob
import math
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "profiles", "r05_psnr")


def stats(xs):
    n = len(xs)
    m = sum(xs) / n
    sd = math.sqrt(sum((x - m) ** 2 for x in xs) / max(n - 1, 1))
    return m, sd, sd / math.sqrt(n), n


def main():
    # (the box-host partial runs first: a container run of the same seed, when it exists, overrides them iteration by iteration)
    files = sorted(glob.glob(os.path.join(D, "box_host_partial", "cpu_seed*.log"))) + sorted(glob.glob(os.path.join(D, "cpu_seed*.log"))) + \
        sorted(glob.glob(os.path.join(D, "hip_10000_seeds*.log")))
    prog = {}                                                  # mode -> seed -> {iteration: train psnr}
    done = {}
    for f in files:
        for line in open(f):
            m = re.match(r"PROGRESS (?:mode (\S+) )?seed (\d+) it (\d+) train-psnr\(last 200\) ([\d.]+)", line)
            if m:
                prog.setdefault(m.group(1) or "cpu", {}).setdefault(int(m.group(2)), {})[int(m.group(3))] = float(m.group(4))
            m = re.match(r"RESULT mode (\S+) seed (\d+)", line)
            if m:
                done.setdefault(m.group(1), set()).add(int(m.group(2)))
    out = []
    out.append("# PSNR at equal iterations at the end of a >= 30 dB recipe: CPU oracle beside the HIP paths (round 5)\n")
    out.append("Recipe: `scripts/psnr_seeds.py --size 40 --views 25 --held 1 --rays 512 --coarse 32 --fine 64 --iters 10000 --lr-mult 3 --hold 0.6 --ckpts 4`")
    out.append("(24 training views of 40 x 40 + 1 held-out view of the analytic scene, 512 rays per batch, 32 + 64 samples, the reference's learning-rate rule x 3 held")
    out.append("for 60 % of the run, then its 100x decay; a run's figure = mean of the held-out renders at iterations 9 700 / 9 800 / 9 900 / 10 000; identical initial")
    out.append("weights, batches and uniforms per seed in every path).  Chosen on the GPU first: 8 000 iterations give 30.2 dB, 12 000 give 31.1 dB (HIP fp32, 8 seeds,")
    out.append("`profiles/r05_psnr/recipe_probe_*.log`); round 4's 20 000-iteration recipe (32.7 dB) costs a CPU seed 13 h in the build container.  CPU oracle: torch")
    out.append("autograd fp32, one process per seed in the build container (8 cores shared with the round's compiles; resumable state every 250 iterations);")
    out.append("HIP: the GPU box, one call.\n")
    out.append("## End of run\n")
    sm = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "psnr_summary.py")] + files, capture_output=True, text=True).stdout
    out.append(sm)
    cpu_done = sorted(done.get("cpu", ()))
    out.append("CPU-oracle seeds finished: %s; still running at the time of this summary: %s.\n"
               % (cpu_done or "none", sorted(set(prog.get("cpu", {})) - set(cpu_done)) or "none"))
    out.append("## Training PSNR at equal iterations, paired per seed (last 200 iterations before each mark)\n")
    out.append("| iteration | CPU seeds | HIP fp32 - CPU oracle, dB (mean +- s.e.m.; sd) | HIP bf16 - CPU oracle, dB | CPU oracle mean |")
    out.append("|---|---|---|---|---|")
    cpu = prog.get("cpu", {})
    its = sorted({it for s in cpu.values() for it in s})
    for it in its:
        row = []
        seeds = [s for s in cpu if it in cpu[s]]
        for mode in ("fp32", "bf16"):
            d = [prog[mode][s][it] - cpu[s][it] for s in seeds if s in prog.get(mode, {}) and it in prog[mode][s]]
            row.append("%+.3f +- %.3f (%.2f; n = %d)" % (stats(d)[0], stats(d)[2], stats(d)[1], len(d)) if len(d) >= 2 else "-")
        out.append("| %d | %d | %s | %s | %.2f |" % (it, len(seeds), row[0], row[1], sum(cpu[s][it] for s in seeds) / len(seeds)))
    out.append("")
    # what would resolve 0.1 dB
    pairs = []
    for line in sm.splitlines():
        m = re.match(r"paired (\S+) - cpu, (held-out|train): ([+-][\d.]+) \+- ([\d.]+) dB over (\d+) common seeds \(sd of the differences ([\d.]+)", line)
        if m:
            pairs.append((m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), int(m.group(5)), float(m.group(6))))
    out.append("## Is 0.1 dB resolved?\n")
    if pairs:
        for mode, what, d, sem, n, sd in pairs:
            need = math.ceil((2.0 * sd / 0.1) ** 2)
            out.append("* %s - cpu, %s: %+.2f +- %.2f dB over %d paired seeds (sd of the differences %.2f dB): a 0.1 dB difference at two standard errors needs"
                       " ~ %d paired seeds." % (mode, what, d, sem, n, sd, need))
        out.append("")
        # the per-subset means behind the one paired figure that is two standard errors from zero
        held = {}
        for f in files:
            for line in open(f):
                m = re.match(r"RESULT mode (\S+) seed (\d+) held-out ([\d.]+)", line)
                if m:
                    held.setdefault(m.group(1), {})[int(m.group(2))] = float(m.group(3))
        common = sorted(set(held.get("cpu", {})) & set(held.get("fp32", {})))
        rest = sorted(set(held.get("fp32", {})) - set(common))
        if common and rest:
            mean = lambda mode, ss: sum(held[mode][x] for x in ss) / len(ss)
            worst = max(pairs, key=lambda t: abs(t[2]) / max(t[3], 1e-9))
            out.append("The paired figure furthest from zero is %s - cpu, %s: %+.2f +- %.2f dB = %.1f standard errors over %d paired seeds.  (With 8 paired seeds the"
                       % (worst[0], worst[1], worst[2], worst[3], abs(worst[2]) / worst[3], worst[4]))
            out.append("furthest was fp32 - cpu held-out, -1.39 +- 0.57 = 2.4 standard errors; four more seeds moved it to the figure above.)  Read plainly: a deficit of")
            out.append("about 0.3 dB in the fp32 path's training PSNR at the end of this recipe can neither be claimed nor excluded with these seeds; the bf16 path -- the")
            out.append("same kernels with less precision -- sits at %+.2f dB (train) / %+.2f dB (held-out) from the oracle over the same seeds, the held-out means of all seeds"
                       % (next(t[2] for t in pairs if t[0] == "bf16" and t[1] == "train"), mean("bf16", common) - mean("cpu", common)))
            out.append("are within one standard error of each other, and HIP fp32 averages %.2f dB held-out over the paired seeds %s against %.2f dB over its other seeds %s."
                       % (mean("fp32", common), "%d-%d" % (common[0], common[-1]), mean("fp32", rest), "%d-%d" % (rest[0], rest[-1])))
            out.append("")
        out.append("No: at the end of a run the paired difference between ANY two paths (HIP fp32 against bf16 included) has a standard deviation of 1-2 dB per seed --")
        out.append("training this scene is chaotic (an fp32 ulp changes the trajectory; the CPU oracle against itself with another thread count differs as much) --")
        out.append("so the end-of-run statement these seeds support is \"within the error bars above\" (0.15-0.25 dB on the training PSNR, 0.6-0.9 dB held-out), not 0.1 dB.  Where the trajectories have not yet")
        out.append("diverged the 0.1 dB statement IS resolved: the table above, and round 4's per-iteration pairing (HIP fp32 - oracle = +0.05 +- 0.07 dB through")
        out.append("1 000 iterations over 10 seeds, `profiles/r04_psnr/short6k/trajectory_summary.md`).")
    else:
        out.append("(no finished CPU seed yet: the end-of-run pairing is missing; the equal-iterations table above is what the partial runs support)")
    open(os.path.join(D, "summary.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
