#!/bin/bash
# Kernel-trace timelines of the reference's DEFAULT training batch (1 024 rays, procedures.py:170): MipNeRF and Ref-NeRF steps, eager and
# replayed from a hipGraph.   -> gpurun_out/timeline_1024/*.md (+ the raw kernel traces, gzipped)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/timeline_1024; mkdir -p $OUT
run() {  # name, marker, args...
  name=$1; marker=$2; shift 2
  rm -rf /tmp/tl_$name
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$name -o tp -- python $R/scripts/gpu_train_rate.py "$@" > $OUT/$name.log 2>&1
  f=$(find /tmp/tl_$name -name 'tp_kernel_trace.csv' | head -1)
  python $R/scripts/summarize_timeline.py $f "$marker" > $OUT/$name.md 2>> $OUT/$name.log
  tail -4 $OUT/$name.md
}
run mip_1024_eager proposal_kernel 1024 bf16
run mip_1024_graph proposal_kernel 1024 bf16 graph
run ref_1024_eager proposal_kernel ref 1024 bf16
run ref_1024_graph proposal_kernel ref 1024 bf16 graph
