#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv compactly: calls, average us, share, short kernel name."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)[:90]
    print("%5d x %10.1f us  %6.2f %%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]), name))
