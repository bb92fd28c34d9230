#!/bin/bash
# Cost probe for the Ref-NeRF TRAINING FORWARD (probe first, build second).  What would a forward whose activation-dump
# stores cost nothing be worth?  Four builds of the same kernel, same box, alternated twice, rocprofv3 kernel averages of the 2^14-ray step:
#   BASE           8 waves x 32 samples (two waves per SIMD at 256 registers; shipped)
#   REFNODUMP      the same without the 512-byte activation stores (wrong results, right cost: the upper bound of ANY store-hiding scheme on this tile)
#   REFWIDE        4 waves x 64 samples (one wave per SIMD at 512 registers: every A fragment feeds two MFMAs)
#   REFWIDENODUMP  the same without the stores (= what a "store-only second wave" could reach at best -- which the register file rules out:
#                  a 512-register wave fills its SIMD, MI355X_MICROARCH.md "Register files")
#   gpurun --timeout 900 -- 'bash scripts/gpu_ref_fwd_probe.sh'   -> gpurun_out/ref_fwd_probe.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ref_fwd_probe.log; : > $OUT
for rep in 1 2; do for v in BASE REFNODUMP REFWIDE REFWIDENODUMP; do
  if [ $v = BASE ]; then unset NERF_AMD_LIB; else export NERF_AMD_LIB=$R/nerf_amd/ablate/libnerf_amd_$v.so; fi
  rm -rf /tmp/rp_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$v -o tp -- python $R/scripts/gpu_train_rate.py ref 16384 bf16 > /tmp/rp_$v.log 2>&1
  echo "== $v: $(grep 'train step' /tmp/rp_$v.log | tail -1)" | tee -a $OUT
  python $R/scripts/kstats.py $(find /tmp/rp_$v -name 'tp_kernel_stats.csv' | head -1) 6 | tee -a $OUT
done; done
