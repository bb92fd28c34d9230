#!/bin/bash
# Instruction-mix PMC passes for the MLP kernels (separate --pmc passes with --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_mix; mkdir -p $OUT
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $1"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT -o m1 -- $B > $OUT/m1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d $OUT -o m2 -- $B > $OUT/m2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH GRBM_GUI_ACTIVE --output-format csv -d $OUT -o m3 -- $B > $OUT/m3.log 2>&1
python - <<PY
import csv, collections
for f in ('m1','m2','m3'):
    try: rows=list(csv.DictReader(open('$OUT/%s_counter_collection.csv'%f)))
    except Exception as e: print(f, 'missing', e); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows: agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'mip_kernel' in k or 'proposal' in k:
            print(f, k, ' '.join('%s=%.4g'%(c,sum(x)/len(x)) for c,x in sorted(v.items())))
PY
