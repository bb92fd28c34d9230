#!/bin/bash
# The driver's 8-rank scaling run, dry: eight gloo ranks time-sharing the box's ONE GPU (RCCL refuses that, gloo allows
# it) through all three modes of bench.py with the DRIVER'S flags (defaults: steps / warm-up / cpu_baseline / train_step legs included) --
# preflight, control flow, per-rank gathers, rank 0's solo legs while seven ranks wait, the JSON line -- and the wall time of each whole line
# (worst case: eight ranks share one device, so the timed region is ~8x a real node's).   -> gpurun_out/$TAG/n8_dryrun.log
R=${GRAFT_REPO_ROOT:-.}; cd $R; TAG=${TAG:-r06_n8}; mkdir -p gpurun_out/$TAG; LOG=gpurun_out/$TAG/n8_dryrun.log; : > $LOG
export BENCH_BACKEND=gloo
for n in ${NS:-8}; do for m in "" "--mode render-strong" "--mode train-ddp" "--mode train-ddp --model ref --train-rays 4096"; do
  t0=$(date +%s)
  timeout 1700 python bench.py --gpus $n $m > gpurun_out/$TAG/n${n}_out.txt 2> gpurun_out/$TAG/n${n}_err.txt; rc=$?
  echo "=== bench.py --gpus $n $m : rc=$rc  wall $(( $(date +%s) - t0 )) s" | tee -a $LOG
  grep "bench.py preflight" gpurun_out/$TAG/n${n}_err.txt | head -8 >> $LOG
  python - <<PY | tee -a $LOG
import json
try:
    d=json.loads([l for l in open("gpurun_out/$TAG/n${n}_out.txt") if l.startswith("{")][-1])
    print("  n_gpus", d["n_gpus"], "value", d["value"], "ms/step %.2f" % d["ms_per_step"], "per-rank ms", [round(x,1) for x in d["ms_per_step_ranks"]["all"]])
    print("  keys", sorted(k for k in d if k not in ("config",)))
except Exception as e:
    print("  no line:", e)
PY
  [ $rc -ne 0 ] && grep -v "^\[Gloo\]\|hostname of the client\|amdgpu.ids" gpurun_out/$TAG/n${n}_err.txt | tail -30 | tee -a $LOG
done; done
