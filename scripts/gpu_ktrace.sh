#!/bin/bash
# kernel-trace of the bench; prints per-kernel average durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/kt_$1; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline $2 > $OUT/log.txt 2>&1
python - <<PY
import csv, collections
d=collections.defaultdict(list)
for r in csv.DictReader(open('$OUT/kt_kernel_trace.csv')):
    d[r['Kernel_Name'][:60]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:5]:
    print('%-62s n=%3d avg=%8.3f ms'%(k,len(v),sum(v)/len(v)/1e6))
PY
tail -1 $OUT/log.txt | cut -c1-200
