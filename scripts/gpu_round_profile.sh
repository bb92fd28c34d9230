#!/bin/bash
# Round profile: kernel-trace stats + PMC traffic passes of the default bench command; outputs under gpurun_out/round/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round; mkdir -p $OUT
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-ref --no-train-rate ${BENCH_EXTRA}"     # BENCH_EXTRA="--model ref": the Ref-NeRF render
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- $B > $OUT/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p1 -- $B > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o p2 -- $B > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o p3 -- $B > $OUT/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT -o p4 -- $B > $OUT/p4.log 2>&1
rm -f $OUT/*_agent_info.csv $OUT/p?_kernel_trace.csv $OUT/kt_kernel_trace.csv     # (gpurun copies back at most 64 MiB)
tail -1 $OUT/kt.log | cut -c1-300
ls $OUT | head -30
