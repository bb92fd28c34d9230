#!/usr/bin/env python3
"""One training step as a TIMELINE from a rocprofv3 kernel trace (what bounds the 1 024-ray step?).
    summarize_timeline.py <tp_kernel_trace.csv> <marker kernel substring> [n_rays]
Takes the LAST complete step of the trace (from one dispatch of the marker kernel -- the first kernel of a step -- to the next), and
prints every dispatch in order: start offset, duration, gap to the previous kernel's end, workgroups; then the totals -- kernel time,
gap time, launches -- and the share of the step spent in kernels that launch fewer workgroups than the chip has CUs."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
ksum = gsum = 0.0
small = 0.0
print("| # | kernel | start us | dur us | gap us | workgroups |")
print("|---|---|---|---|---|---|")
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*$", "", name)[:70]
    wg = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * max(1, int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1)))) \
        * max(1, int(r.get("Grid_Size_Z", 1)) // max(1, int(r.get("Workgroup_Size_Z", 1))))
    gap = (s - prev_end) / 1e3
    dur = (e - s) / 1e3
    ksum += dur
    gsum += max(gap, 0.0)
    if wg < 256:
        small += dur
    print("| %d | %s | %.1f | %.1f | %.1f | %d |" % (i, name, (s - t0) / 1e3, dur, gap, wg))
    prev_end = max(prev_end, e)
total = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
print()
print("step (marker to marker): %.1f us; kernels %d; sum of kernel durations %.1f us (%.0f %%); gaps between kernels %.1f us (%.0f %%)"
      % (total, len(step), ksum, 100 * ksum / total, total - ksum, 100 * (total - ksum) / total))
print("kernels launching fewer than 256 workgroups (less than one per CU): %.1f us of kernel time (%.0f %% of the step)" % (small, 100 * small / total))
top = sorted(step, key=lambda r: int(r["Start_Timestamp"]) - int(r["End_Timestamp"]))[:8]
print("longest: " + "; ".join("%s %.1f" % (re.sub(r"\(anonymous namespace\)::|^void |\(.*$|<.*$", "", r["Kernel_Name"])[:30], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in top))
