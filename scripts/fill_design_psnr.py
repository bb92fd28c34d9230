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
open(p, "w").write(s)
print(block)
