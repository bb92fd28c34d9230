#!/usr/bin/env python3
"""profiles/r04_psnr/short6k/trajectory_seed<S>_<N>.txt from the three JSON files scripts/psnr_trajectory.py wrote for seed S over N iterations
(traj_cpu_seed<S>_<N>.json, traj_fp32_..., traj_bf16_...): training PSNR of 100-iteration windows, CPU oracle | HIP fp32 | HIP bf16 on the same draws."""
import json
import math
import os
import sys

D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles/r04_psnr/short6k")
S, N = int(sys.argv[1]), int(sys.argv[2])
c, f, b = (json.load(open(os.path.join(D, "traj_%s_seed%d_%d.json" % (m, S, N)))) for m in ("cpu", "fp32", "bf16"))
C, Fh, B = c["loss_img"], f["loss_img"], b["loss_img"]
ps = lambda x: -10 * math.log10(x)
lines = ["# Training loss of seed %d over %d iterations of the 6 000-iteration recipe, same stream of batches and uniforms in all three paths: CPU oracle | HIP fp32 | HIP bf16" % (S, N),
         "# window   mean loss cpu / fp32 / bf16   ->  training PSNR dB cpu / fp32 / bf16   (fp32 - cpu, bf16 - cpu)"]
for lo in range(0, N, 100):
    m = lambda x: sum(x[lo:lo + 100]) / len(x[lo:lo + 100])
    lines.append("%4d-%4d   %.5f / %.5f / %.5f   ->  %.3f / %.3f / %.3f   (%+.3f, %+.3f)"
                 % (lo, lo + 100, m(C), m(Fh), m(B), ps(m(C)), ps(m(Fh)), ps(m(B)), ps(m(Fh)) - ps(m(C)), ps(m(B)) - ps(m(C))))
first = [next((i for i in range(N) if abs(Fh[i] - C[i]) > t * C[i]), None) for t in (1e-3, 1e-2)]
lines.append("# first iteration whose fp32 loss is more than 0.1 %% / 1 %% away from the oracle's: %s / %s" % tuple(first))
lines.append("# held-out view after iteration %d: cpu %.3f dB, fp32 %.3f dB, bf16 %.3f dB" % (N, c["held_out_psnr"][-1], f["held_out_psnr"][-1], b["held_out_psnr"][-1]))
open(os.path.join(D, "trajectory_seed%d_%d.txt" % (S, N)), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
