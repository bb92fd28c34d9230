#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of the bench workload for library variants:
# AB_LIST = names under nerf_amd/ablate (BASE = the shipped library).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${AB_LIST:-BASE}; do
  if [ $v = BASE ]; then unset NERF_AMD_LIB; else export NERF_AMD_LIB=$R/nerf_amd/ablate/libnerf_amd_$v.so; fi
  rm -rf /tmp/kt_$v; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-ref --no-train-rate > /tmp/kt_$v.log 2>&1
  echo "== $v"; python - <<PY
import csv
for r in csv.DictReader(open('/tmp/kt_$v/kt_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('composite','resample','proposal','mip_kernel')): print('%-28s %9.3f ms' % (r['Name'].split('(')[0][-28:], float(r['AverageNs'])/1e6))
PY
done
