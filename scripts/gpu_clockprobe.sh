#!/bin/bash
# Runs bench.py on diagnostic library variants built with -DMLP_CLOCKPROBE (make variant NAME=CP_x FLAGS="-DMLP_CLOCKPROBE ...")
# and prints the fine kernel's duration together with the shader clock it ran at: the chip is power-limited on real data,
# so a variant's time has to be read together with its clock.  CP_LIST = variant names.
R=$GRAFT_REPO_ROOT
export NERF_AMD_CLOCKPROBE=1
for v in ${CP_LIST:-CP_N}; do
  export NERF_AMD_LIB=$R/nerf_amd/ablate/libnerf_amd_$v.so
  echo "== $v"; (timeout 120 python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | grep -E "clockprobe|ms_per_step" | sed -E 's/.*"ms_per_step": ([0-9.]+).*/step \1/' | tail -3 | tr '\n' ';'); echo
done
