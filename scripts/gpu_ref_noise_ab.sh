#!/bin/bash
# Ref-NeRF 2^14-ray training step: bottle-neck perturbation drawn in-kernel (philox, default) against the torch.normal tensor (round 4),
# same box, alternated; rocprofv3 kernel averages.   gpurun -- 'bash scripts/gpu_ref_noise_ab.sh'  -> gpurun_out/ref_noise_ab.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in philox torch; do
  rm -rf /tmp/rn_$v
  REF_NOISE_RNG=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rn_$v -o tp -- python $R/scripts/gpu_train_rate.py ref 16384 bf16 > /tmp/rn_$v.log 2>&1
  echo "== $v: $(grep 'train step' /tmp/rn_$v.log | tail -1)"
  python $R/scripts/kstats.py $(find /tmp/rn_$v -name 'tp_kernel_stats.csv' | head -1) 14
done; done
