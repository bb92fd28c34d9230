#!/usr/bin/env python3
"""mean +- s.e.m. over the seeds of the RESULT lines scripts/psnr_seeds.py wrote (profiles/r03_psnr/*.log), per mode, and the gap of every
HIP mode to the CPU oracle with its standard error (independent samples: sqrt(sem_a^2 + sem_b^2))."""
import glob
import math
import re
import sys

files = sys.argv[1:] or sorted(glob.glob("profiles/r03_psnr/*.log"))
res = {}
for f in files:
    for line in open(f):
        m = re.match(r"RESULT mode (\S+) seed (\d+) held-out ([\d.]+) dB .* train-tail ([\d.]+) dB", line)
        if m:
            res.setdefault(m.group(1), {})[int(m.group(2))] = (float(m.group(3)), float(m.group(4)))


def stats(xs):
    n = len(xs)
    mean = sum(xs) / n
    sd = math.sqrt(sum((x - mean) ** 2 for x in xs) / max(n - 1, 1))
    return mean, sd, sd / math.sqrt(n), n


print("| path | seeds | held-out PSNR dB (mean +- s.e.m.; sd) | train PSNR, last 200 it (mean +- s.e.m.; sd) |")
print("|---|---|---|---|")
st = {}
for mode in ("cpu", "fp32", "bf16", "bf16-fp8dumps"):
    if mode not in res:
        continue
    v = res[mode]
    st[mode] = [stats([x[k] for x in v.values()]) for k in (0, 1)]
    print("| %s | %d | %.2f +- %.2f (%.2f) | %.2f +- %.2f (%.2f) |" % (mode, st[mode][0][3], st[mode][0][0], st[mode][0][2], st[mode][0][1], st[mode][1][0], st[mode][1][2], st[mode][1][1]))
if "cpu" in st:
    for mode in st:
        if mode == "cpu":
            continue
        for k, name in ((0, "held-out"), (1, "train")):
            d = st[mode][k][0] - st["cpu"][k][0]
            if st["cpu"][k][3] < 4:          # one to three CPU runs carry no usable error bar of their own: the run-to-run sd of the larger sample stands in
                sd, nc, nm = st[mode][k][1], st["cpu"][k][3], st[mode][k][3]
                se = sd * math.sqrt(1.0 / nc + 1.0 / nm)
                print("gap %s - cpu, %s: %+.2f +- %.2f dB (%.1f sigma; %d CPU run%s, error from the %s runs' sd %.2f dB)"
                      % (mode, name, d, se, abs(d) / se if se else 0.0, nc, "" if nc == 1 else "s", mode, sd))
                continue
            se = math.sqrt(st[mode][k][2] ** 2 + st["cpu"][k][2] ** 2)
            print("gap %s - cpu, %s: %+.2f +- %.2f dB (%.1f sigma)" % (mode, name, d, se, abs(d) / se if se else 0.0))
# paired comparison on the seeds two modes share (identical initial weights, batches and uniforms per seed): mean of the per-seed differences
# +- its standard error -- tighter than the independent-sample gap whenever the seed explains part of a run's PSNR
print()
modes = [m for m in ("cpu", "fp32", "bf16", "bf16-fp8dumps") if m in res]
for i, a in enumerate(modes):
    for b in modes[i + 1:]:
        common = sorted(set(res[a]) & set(res[b]))
        if len(common) < 2:
            continue
        for k, name in ((0, "held-out"), (1, "train")):
            m, sd, sem, n = stats([res[b][s][k] - res[a][s][k] for s in common])
            xs, ys = [res[a][s][k] for s in common], [res[b][s][k] for s in common]
            mx, my = sum(xs) / n, sum(ys) / n
            den = math.sqrt(sum((x - mx) ** 2 for x in xs) * sum((y - my) ** 2 for y in ys))
            corr = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / den if den else 0.0
            print("paired %s - %s, %s: %+.2f +- %.2f dB over %d common seeds (sd of the differences %.2f, correlation %.2f)" % (b, a, name, m, sem, n, sd, corr))
