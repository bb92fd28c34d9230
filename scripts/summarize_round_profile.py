#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of scripts/gpu_round_profile.sh (gpurun_out/round/) into the tracked files under
profiles/: <prefix>_kernel_stats.csv (native --stats table), <prefix>_pmc_p{1..4}.csv (counter rows of our kernels only),
<prefix>_pmc_summary.md and pmc_traffic.json (HBM bytes per launch of the dominant kernel, read by bench.py)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "round")
DST = os.path.join(ROOT, "profiles")
prefix = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
OURS = ("mip_kernel", "proposal_kernel", "ref_kernel", "resample_kernel", "composite_kernel", "raygen_kernel")


def short(name):
    for k in OURS:
        if k in name:
            return k
    return None


shutil.copy(os.path.join(SRC, "kt_kernel_stats.csv"), os.path.join(DST, prefix + "_kernel_stats.csv"))
stats = {}
for r in csv.DictReader(open(os.path.join(SRC, "kt_kernel_stats.csv"))):
    k = short(r["Name"])
    if k:
        stats[k] = float(r["AverageNs"]) / 1e6
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2", "p3", "p4"):
    rows = list(csv.DictReader(open(os.path.join(SRC, p + "_counter_collection.csv"))))
    keep = [r for r in rows if short(r["Kernel_Name"])]
    with open(os.path.join(DST, "%s_pmc_%s.csv" % (prefix, p)), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
    for r in keep:
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if p == "p1" and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            agg[k]["ns_p1"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
mean = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
lines = ["# %s -- rocprofv3 summary of `python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-ref --no-train-rate` on MI355X" % prefix,
         "Source: `scripts/gpu_round_profile.sh` (one `--kernel-trace --stats` run + four separate `--pmc` passes), condensed by "
         "`scripts/summarize_round_profile.py`. Native rocprofv3 stats: `%s_kernel_stats.csv`; counter rows: `%s_pmc_p1..4.csv`." % (prefix, prefix), "",
         "| kernel | avg ms (kernel-trace) | MFMA busy cycles/launch | MFMA pipe busy | clock GHz (profiled pass) | WAIT_ANY | WAIT_INST | FETCH_SIZE KiB | WRITE_SIZE KiB | LDS bank conflicts |",
         "|---|---|---|---|---|---|---|---|---|---|"]
traffic = {}
for k in ("mip_kernel", "ref_kernel", "proposal_kernel", "resample_kernel", "composite_kernel"):
    if k not in mean:
        continue
    m = mean[k]
    gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                       # summed over 8 XCDs
    clk = gui / m["ns_p1"] if m.get("ns_p1") else float("nan")
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / gui if gui else 0.0      # 256 CUs x 4 SIMDs
    wc = m.get("SQ_WAVE_CYCLES", 1.0)
    fetch, write = m.get("FETCH_SIZE", 0.0), m.get("WRITE_SIZE", 0.0)
    traffic[k] = (2.0 * fetch + write) * 1024.0                     # gfx950: FETCH_SIZE reports half of a wide coalesced read stream
    lines.append("| %s | %.3f | %.4g | %.1f %% | %.2f | %.1f %% | %.1f %% | %.4g | %.4g | %.3g |" % (
        k, stats.get(k, float("nan")), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), 100 * busy, clk, 100 * m.get("SQ_WAIT_ANY", 0) / wc,
        100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, fetch, write, m.get("SQ_LDS_BANK_CONFLICT", 0.0)))
lines += ["", "Instruction mix per MFMA (pass 4 `SQ_INSTS_*` / pass 1 `SQ_VALU_MFMA_BUSY_CYCLES` / 32; SQ_INSTS_VALU includes the MFMAs):", ""]
for k in ("mip_kernel", "ref_kernel", "proposal_kernel"):
    if k in mean and mean[k].get("SQ_VALU_MFMA_BUSY_CYCLES"):
        m = mean[k]
        n_mfma = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0
        lines.append("* %s: %.3g MFMA per launch; VALU (non-MFMA) %.2f, SALU %.2f, LDS %.2f per MFMA" % (
            k, n_mfma, (m.get("SQ_INSTS_VALU", 0.0) - n_mfma) / n_mfma, m.get("SQ_INSTS_SALU", 0.0) / n_mfma, m.get("SQ_INSTS_LDS", 0.0) / n_mfma))
lines += ["", "HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B (FETCH_SIZE doubled: gfx950 reports half of a wide coalesced "
          "read stream, MI355X_MICROARCH.md section HBM):", ""]
lines += ["* %s: %.3f GB" % (k, v / 1e9) for k, v in traffic.items()]
lines += ["", "Algorithmic bytes per launch (DESIGN.md section 2): mip_kernel 0.35 GB in + 1.31 GB out = 1.66 GB; proposal 0.18 + 0.16 = 0.34 GB; "
          "resample 0.49 GB in the default Philox mode (0.99 GB with resident uniforms); composite 1.98 GB."]
open(os.path.join(DST, prefix + "_pmc_summary.md"), "w").write("\n".join(lines) + "\n")
# recorded WITH the hashes of the kernel sources of this tree (bench.py flags the figure as stale when they change)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import update_pmc_traffic as upt
tj = json.load(open(upt.FILE)) if os.path.exists(upt.FILE) else {}
if traffic.get("ref_kernel") and not traffic.get("mip_kernel"):            # a profile of `bench.py --model ref` (BENCH_EXTRA="--model ref")
    tj["ref_bf16"] = {"bytes": traffic["ref_kernel"], "round": prefix.split("_")[0], "sources": upt.source_hashes("ref_bf16"),
                      "measured_by": "profiles/%s_pmc_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --model ref`; bytes per launch "
                                     "of ref_kernel)" % prefix}
if traffic.get("mip_kernel"):
    tj["mip_bf16"] = {"bytes": traffic["mip_kernel"], "round": prefix.split("_")[0], "sources": upt.source_hashes("mip_bf16"),
                      "measured_by": "profiles/%s_pmc_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the default bench command; bytes per "
                                     "launch of the dominant kernel)" % prefix}
json.dump(tj, open(upt.FILE, "w"), indent=1)
print("\n".join(lines))
