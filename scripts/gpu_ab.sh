#!/bin/bash
# A/B of library variants on the bench workload: AB_LIST = names under nerf_amd/ablate (BASE = the shipped library),
# alternated AB_REPS times; BENCH_ARGS are passed to bench.py.
R=$GRAFT_REPO_ROOT
for rep in $(seq 1 ${AB_REPS:-2}); do
for v in ${AB_LIST:-BASE}; do
  if [ $v = BASE ]; then unset NERF_AMD_LIB; else export NERF_AMD_LIB=$R/nerf_amd/ablate/libnerf_amd_$v.so; fi
  echo -n "$v: "; python $R/bench.py --steps ${AB_STEPS:-12} --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step %.2f ms  dominant kernel %.2f ms  proposal kernel %.3f ms'%(d['ms_per_step'], d['roofline']['ms_per_launch'], d.get('roofline_proposal',{}).get('ms_per_launch', float('nan'))))"
done; done
