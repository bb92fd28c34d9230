#!/bin/bash
# rocprofv3 kernel statistics of the training step (scripts/gpu_train_rate.py) for the shipped library and, optionally,
# variants under nerf_amd/ablate (AB_LIST="BASE SAFE ...").  Output: gpurun_out/trainprof/<variant>_<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trainprof; mkdir -p $OUT
for v in ${AB_LIST:-BASE}; do
  if [ $v = BASE ]; then unset NERF_AMD_LIB; else export NERF_AMD_LIB=$R/nerf_amd/ablate/libnerf_amd_$v.so; fi
  for cfg in ${CFG_LIST:-"16384_bf16"}; do
    args=$(echo $cfg | tr _ ' ')
    rm -rf /tmp/tp_$v_$cfg
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp_${v}_$cfg -o tp -- python $R/scripts/gpu_train_rate.py $args > $OUT/${v}_$cfg.log 2>&1
    cp $(find /tmp/tp_${v}_$cfg -name 'tp_kernel_stats.csv' | head -1) $OUT/${v}_${cfg}_kernel_stats.csv
    echo "== $v $cfg: $(tail -1 $OUT/${v}_$cfg.log)"; python $R/scripts/kstats.py $OUT/${v}_${cfg}_kernel_stats.csv 12
  done
done
# optional HBM-traffic passes of the first configuration with the shipped library (PMC=1): separate --pmc runs, as the guide prescribes
if [ -n "$PMC" ]; then
  unset NERF_AMD_LIB
  cfg=$(echo ${CFG_LIST:-"16384_bf16"} | cut -d' ' -f1); args=$(echo $cfg | tr _ ' ')
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tpm_$c
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tpm_$c -o tp -- python $R/scripts/gpu_train_rate.py $args > $OUT/pmc_$c.log 2>&1
    cp $(find /tmp/tpm_$c -name 'tp_counter_collection.csv' | head -1) $OUT/pmc_${cfg}_$c.csv
  done
  python $R/scripts/summarize_train_profile.py $OUT $cfg
fi
