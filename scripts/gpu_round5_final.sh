#!/bin/bash
# Round 5 measurement call: -m gpu suite, the bench lines, rocprofv3 kernel stats + PMC passes of the render bench and of BOTH 2^14-ray
# training steps (MipNeRF, Ref-NeRF with prop_normal).   gpurun --timeout 3000 -- 'bash scripts/gpu_round5_final.sh'   -> gpurun_out/r05_final/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TAG=r05_final; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
bash scripts/gpu_job.sh tests -s
bash scripts/gpu_job.sh benchall
bash scripts/gpu_job.sh profile
cp -r gpurun_out/round $OUT/round
for cfg in 16384_bf16 ref_16384_bf16; do
  rm -rf gpurun_out/trainprof
  CFG_LIST=$cfg PMC=1 bash scripts/gpu_train_profile.sh > $OUT/trainprof_$cfg.log 2>&1
  mkdir -p $OUT/trainprof_$cfg; cp gpurun_out/trainprof/* $OUT/trainprof_$cfg/ 2>/dev/null
  tail -25 $OUT/trainprof_$cfg.log
done
