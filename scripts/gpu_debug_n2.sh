#!/bin/bash
# bisect a hang of the two-rank bench path on the one-GPU box: every variant under its own timeout, stacks dumped after 100 s
R=${GRAFT_REPO_ROOT:-.}; cd $R; mkdir -p gpurun_out/r04
export BENCH_BACKEND=gloo BENCH_DUMP_STACKS_AFTER=100
for v in "--no-train-rate --no-cpu-baseline --no-gemm-ref" "--no-train-rate --no-cpu-baseline" "--no-train-rate" ""; do
  echo "=== bench --gpus 2 $v"; t0=$(date +%s)
  timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --cpu-rays 2500 $v > gpurun_out/r04/n2_out.txt 2> gpurun_out/r04/n2_err.txt; rc=$?
  echo "rc=$rc  $(( $(date +%s) - t0 )) s"; tail -c 300 gpurun_out/r04/n2_out.txt; echo
  if [ $rc -ne 0 ]; then grep -v "^\[Gloo\]\|hostname of the client" gpurun_out/r04/n2_err.txt | tail -60; break; fi
done
