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
            se = math.sqrt(st[mode][k][2] ** 2 + st["cpu"][k][2] ** 2)
            print("gap %s - cpu, %s: %+.2f +- %.2f dB (%.1f sigma)" % (mode, name, d, se, abs(d) / se if se else 0.0))
