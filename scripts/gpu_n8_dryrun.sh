#!/bin/bash
# the driver's 8-rank scaling run, dry: eight gloo ranks time-sharing the box's ONE GPU (RCCL refuses that, gloo allows it) through all three
# modes of bench.py -- control flow, per-rank gathers, rank-0 extras while seven ranks wait, the JSON line; N = 4 as well
R=${GRAFT_REPO_ROOT:-.}; cd $R; mkdir -p gpurun_out/r04
export BENCH_BACKEND=gloo BENCH_DUMP_STACKS_AFTER=500
for n in 4 8; do for m in "" "--mode render-strong" "--mode train-ddp --train-rays 4096"; do
  t0=$(date +%s)
  timeout 600 python bench.py --gpus $n --steps 2 --warmup 1 --cpu-rays 2500 $m > gpurun_out/r04/n${n}_out.txt 2> gpurun_out/r04/n${n}_err.txt; rc=$?
  echo "=== bench --gpus $n $m : rc=$rc  $(( $(date +%s) - t0 )) s"
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r04/n${n}_out.txt") if l.startswith("{")][-1])
    print("  n_gpus", d["n_gpus"], "value %.4g" % d["value"], "ms %.2f" % d["ms_per_step"], "ranks", [round(x,1) for x in d["ms_per_step_ranks"]["all"]],
          "streams", [round(r["weights_x_relu_activations"]) for r in d["roofline"]["mfma_stream_ref"]["per_rank"]], "keys", sorted(k for k in d if k not in ("config",)))
except Exception as e:
    print("  no line:", e)
PY
  [ $rc -ne 0 ] && grep -v "^\[Gloo\]\|hostname of the client\|amdgpu.ids" gpurun_out/r04/n${n}_err.txt | tail -30
done; done
