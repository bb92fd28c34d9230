#!/usr/bin/env python3
"""the block between <!-- short6k:begin --> and <!-- short6k:end --> = the table and paired lines of
profiles/r04_psnr/short6k/summary.md (scripts/psnr_r04_short_summary.sh regenerates that from whatever seeds have finished)."""
import os
import re
import subprocess

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run(["bash", os.path.join(root, "scripts/psnr_r04_short_summary.sh")], check=True, stdout=subprocess.DEVNULL)
lines = [l.rstrip() for l in open(os.path.join(root, "profiles/r04_psnr/short6k/summary.md"))]
body = [l for l in lines if l.startswith("|") or l.startswith("gap ") or l.startswith("paired ")]
block = "<!-- short6k:begin -->\n" + "\n".join("  " + l if l.startswith("|") else "  " + l + "  " for l in body) + "\n<!-- short6k:end -->"
p = os.path.join(root, "DESIGN.md")
s = open(p).read()
if "SHORT6K_TABLE" in s:
    s = s.replace("SHORT6K_TABLE", block)
else:
    s = re.sub(r"<!-- short6k:begin -->.*?<!-- short6k:end -->", lambda m: block, s, flags=re.S)
# the paired equal-iterations table (scripts/psnr_trajectory_summary.py) between <!-- traj:begin --> and <!-- traj:end -->
subprocess.run(["python", os.path.join(root, "scripts/psnr_trajectory_summary.py"), "1000"], check=True, stdout=subprocess.DEVNULL)
tl = [l.rstrip() for l in open(os.path.join(root, "profiles/r04_psnr/short6k/trajectory_summary.md"))]
tbody = [l for l in tl if l.startswith("|") or l.startswith("Held-out view") or l.startswith("First iteration") or l.startswith("# PSNR at equal")]
tblock = "<!-- traj:begin -->\n" + "\n".join("  " + (l[2:] + ":" if l.startswith("# ") else l) + ("" if l.startswith("|") else "  ") for l in tbody) + "\n<!-- traj:end -->"
if "TRAJ_TABLE" in s:
    s = s.replace("TRAJ_TABLE", tblock)
else:
    s = re.sub(r"<!-- traj:begin -->.*?<!-- traj:end -->", lambda m: tblock, s, flags=re.S)
open(p, "w").write(s)
print(block)
print(tblock)
