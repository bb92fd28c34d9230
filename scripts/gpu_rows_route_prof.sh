#!/bin/bash
# rocprofv3 kernel stats of render_image on the bf16-rows route: Ref-NeRF --ide_level 5 (200 x 200) and MipNeRF / proposal at width 512 (400 x 400)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/rows_route_prof; mkdir -p $OUT
for which in ref5 mip512; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $which -- python $R/scripts/gpu_rows_route_ab.py --only $which > $OUT/$which.log 2>&1
  python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/${which}_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('== $which: total kernel time %.1f ms' % (tot/1e6))
for r in rows[:22]:
    print('%-86s calls %5s avg %9.1f us %5.1f%%' % (r['Name'][:86], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
done
