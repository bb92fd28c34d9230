#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_b; mkdir -p $OUT
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $1"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p1 -- $B > $OUT/p1.log 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open('$OUT/p1_counter_collection.csv')))
kt={r['Dispatch_Id']:(int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in csv.DictReader(open('$OUT/p1_kernel_trace.csv'))}
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'][:42]; agg[k][r['Counter_Name']].append(float(r['Counter_Value'])); agg[k]['ns'].append(kt.get(r['Dispatch_Id'],0))
for k,v in agg.items():
    if 'mip_kernel' in k or 'proposal' in k:
        a={c:sum(x)/len(x) for c,x in v.items()}
        ns=a['ns']; clk=a['GRBM_GUI_ACTIVE']/8/ns
        print(k, 'ms=%.2f clk=%.3f GHz mfma_busy=%.1f%% wait_any=%.1f%% wait_inst=%.1f%% active=%.1f%%'%(ns/1e6, clk, 100*a['SQ_VALU_MFMA_BUSY_CYCLES']/1024/(a['GRBM_GUI_ACTIVE']/8), 100*a['SQ_WAIT_ANY']/a['SQ_WAVE_CYCLES'],100*a['SQ_WAIT_INST_ANY']/a['SQ_WAVE_CYCLES'],100*a['SQ_ACTIVE_INST_ANY']/a['SQ_WAVE_CYCLES']))
PY
