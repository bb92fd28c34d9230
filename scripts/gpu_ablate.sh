#!/bin/bash
R=$GRAFT_REPO_ROOT
for v in ${ABL_LIST:-BASE NOEPI NOPE NOBAR NOLDSA NOGLDS HALFLDS DUMMYLDS}; do
  if [ $v = BASE ]; then unset NERF_AMD_LIB; else export NERF_AMD_LIB=$R/nerf_amd/ablate/libnerf_amd_$v.so; fi
  echo -n "$v: "; (timeout 120 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step %.2f ms  fine %.2f ms'%(d['ms_per_step'], d['roofline']['ms_per_launch']))")
done
