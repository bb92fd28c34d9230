#!/bin/bash
# PMC passes over nerf_amd_rows_gemm at 262 144 x 512 x 512 (separate --pmc passes with --kernel-trace only): where its cycles go.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/rows_gemm_pmc; mkdir -p $OUT
B="python $R/scripts/gpu_rows_gemm_rate.py --pmc"
run() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $n -- $B > $OUT/$n.log 2>&1; }
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- $B > $OUT/ks.log 2>&1
run m1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run m2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
run m3 FETCH_SIZE
run m5 WRITE_SIZE
run m6 TCC_HIT_sum TCC_MISS_sum
run m4 SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_BUSY_avr
python - <<PY
import csv, collections, glob
for f in ('ks_kernel_stats',):
    for p in glob.glob('$OUT/%s.csv' % f):
        for r in list(csv.DictReader(open(p)))[:6]: print(r.get('Name','')[:60], r.get('Calls'), r.get('AverageNs'))
for f in ('m1','m2','m3','m4','m5','m6'):
    try: rows=list(csv.DictReader(open('$OUT/%s_counter_collection.csv'%f)))
    except Exception as e: print(f, 'missing', e); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows: agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'rows_gemm' in k or 'gemm_kernel' in k:
            print(f, k, ' '.join('%s=%.4g'%(c,sum(x)/len(x)) for c,x in sorted(v.items())))
PY
