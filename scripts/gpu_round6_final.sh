#!/bin/bash
# Round 6 measurement call: -m gpu suite, the bench lines, rocprofv3 kernel stats + PMC passes of the render bench and of BOTH 2^14-ray
# training steps (MipNeRF, Ref-NeRF with prop_normal).   gpurun --timeout 3300 -- 'bash scripts/gpu_round6_final.sh'   -> gpurun_out/r06_final/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TAG=r06_final; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
rm -f gpurun_out/measured_gates.log
TESTS_TIMEOUT=1500 bash scripts/gpu_job.sh tests -s
cp gpurun_out/measured_gates.log $OUT/measured_gates.log 2>/dev/null
bash scripts/gpu_job.sh benchall
bash scripts/gpu_job.sh profile
rm -rf $OUT/round; cp -r gpurun_out/round $OUT/round
for cfg in 16384_bf16 ref_16384_bf16; do
  rm -rf gpurun_out/trainprof
  CFG_LIST=$cfg PMC=1 ROUND_TAG=r06 bash scripts/gpu_train_profile.sh > $OUT/trainprof_$cfg.log 2>&1
  mkdir -p $OUT/trainprof_$cfg; cp gpurun_out/trainprof/* $OUT/trainprof_$cfg/ 2>/dev/null
  tail -25 $OUT/trainprof_$cfg.log
done
