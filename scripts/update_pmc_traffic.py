#!/usr/bin/env python3
"""Record a counter-measured HBM traffic figure in profiles/pmc_traffic.json TOGETHER with the hashes of the kernel sources it was measured
on: bench.py prints `traffic_stale: true` when the tree's sources differ from the ones recorded here.

    update_pmc_traffic.py KEY BYTES ROUND "SOURCE TEXT"      e.g.  mip_bf16 1739052105 r06 "profiles/r06_final_pmc_summary.md ..."
    update_pmc_traffic.py --restamp KEY ROUND                 (the figure was re-measured on this tree and did not change)
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")
# which sources a figure depends on (the dominant kernel of the render step; every kernel of a training step)
SOURCES = {"mip": ("mlp_kernels.hip", "mlp_core.h", "mlp_layout.h", "device_common.h"),
           "ref": ("mlp_kernels.hip", "mlp_core.h", "mlp_layout.h", "device_common.h"),
           "train_step": ("bwd_kernels.hip", "mlp_kernels.hip", "mlp_core.h", "mlp_layout.h", "device_common.h", "sample_kernels.hip")}


def source_hashes(key):
    fam = "train_step" if key.startswith("train_step") else key.split("_")[0]
    out = {}
    for f in SOURCES[fam]:
        with open(os.path.join(ROOT, "nerf_amd", "csrc", f), "rb") as fh:
            out[f] = hashlib.sha256(fh.read()).hexdigest()[:16]
    return out


def main():
    rec = json.load(open(FILE))
    if sys.argv[1] == "--restamp":
        key, rnd = sys.argv[2], sys.argv[3]
        rec[key].update(round=rnd, sources=source_hashes(key))
    else:
        key, nbytes, rnd, src = sys.argv[1], float(sys.argv[2]), sys.argv[3], sys.argv[4]
        rec[key] = {"bytes": nbytes, "round": rnd, "measured_by": src, "sources": source_hashes(key)}
    json.dump(rec, open(FILE, "w"), indent=1)
    print(json.dumps(rec[key]))


if __name__ == "__main__":
    main()
