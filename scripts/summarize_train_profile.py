#!/usr/bin/env python3
"""Per-kernel HBM traffic of the training step from the two --pmc passes of scripts/gpu_train_profile.sh (PMC=1) next to the
kernel-trace averages of the same command: prints a markdown table (and writes <dir>/train_pmc_summary.md).
HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB: gfx950 reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import os
import re
import sys

d, cfg = sys.argv[1], sys.argv[2]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)


avg = {}
for r in csv.DictReader(open(os.path.join(d, "BASE_%s_kernel_stats.csv" % cfg))):
    avg[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
val = {"FETCH_SIZE": collections.defaultdict(list), "WRITE_SIZE": collections.defaultdict(list)}
for c in val:
    for r in csv.DictReader(open(os.path.join(d, "pmc_%s_%s.csv" % (cfg, c)))):
        if r["Counter_Name"] == c:
            val[c][short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
lines = ["| kernel | launches / step | avg ms | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM GB / launch | TB/s |", "|---|---|---|---|---|---|---|"]
steps = next((v[0] for k, v in avg.items() if k.startswith("mip_bwd_kernel")), 1)
for k, (calls, ms) in sorted(avg.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if k not in val["FETCH_SIZE"] or ms < 0.05 or k.startswith("at::"):
        continue
    f = sum(val["FETCH_SIZE"][k]) / len(val["FETCH_SIZE"][k])
    w = sum(val["WRITE_SIZE"][k]) / max(1, len(val["WRITE_SIZE"][k]))
    gb = (2 * f + w) * 1024 / 1e9
    lines.append("| %s | %g | %.3f | %.4g | %.4g | %.3f | %.2f |" % (k, calls / steps, ms, f, w, gb, gb / ms))
# whole-step traffic (every dispatch of the profiled run, torch's own kernels included) / steps in the run -> profiles/pmc_traffic.json
import json
tot = {c: 0.0 for c in val}
steps_pmc = 0
for c in val:
    rows = [r for r in csv.DictReader(open(os.path.join(d, "pmc_%s_%s.csv" % (cfg, c)))) if r["Counter_Name"] == c]
    tot[c] = sum(float(r["Counter_Value"]) for r in rows)
    if c == "FETCH_SIZE":
        steps_pmc = len(set(r["Dispatch_Id"] for r in rows if "mip_bwd_kernel" in r["Kernel_Name"] or "ref_heads_delta_kernel" in r["Kernel_Name"]))
step_bytes = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / max(1, steps_pmc)
lines += ["", "Whole step (all dispatches of the run / %d steps): %.2f GB of HBM traffic." % (steps_pmc, step_bytes / 1e9)]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import update_pmc_traffic as upt
tj = json.load(open(upt.FILE)) if os.path.exists(upt.FILE) else {}
tj["train_step_%s" % cfg] = {"bytes": step_bytes, "round": os.environ.get("ROUND_TAG", "r06"), "sources": upt.source_hashes("train_step_%s" % cfg),
                             "measured_by": "scripts/gpu_train_profile.sh PMC=1 (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_train_rate.py), all dispatches of a step"}
json.dump(tj, open(upt.FILE, "w"), indent=1)
out = "\n".join(lines) + "\n"
open(os.path.join(d, "train_pmc_summary.md"), "w").write(out)
print(out)
