#!/bin/bash
# Generic-path GEMM A/B on one box, alternated: library variants under nerf_amd/ablate (BASE = the shipped library) x the host-side
# re-layout of small row-major B operands (NERF_AMD_GEMM_RELAYOUT_B, ops.gemm); the three stride forms at 262 144 x W x W, then a
# render_image call and a training step (scripts/gpu_generic_rate.py).
#   AB_LIST="GBK32 BASE" RELAYOUT="0 1" WIDTHS="512 320" bash scripts/gpu_generic_ab.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=$R/gpurun_out/${TAG:-r04}; mkdir -p $OUT; cd $R
for rep in 1 2; do
  for v in ${AB_LIST:-GBK32 BASE}; do
    if [ $v = BASE ]; then unset NERF_AMD_LIB; else export NERF_AMD_LIB=$R/nerf_amd/ablate/libnerf_amd_$v.so; fi
    for rl in ${RELAYOUT:-1}; do
      for w in ${WIDTHS:-512 320}; do
        echo -n "$v relayout_b=$rl width $w: "; NERF_AMD_GEMM_RELAYOUT_B=$rl python scripts/gpu_generic_rate.py $w 2>/dev/null | grep -E "^gemm bf16|^gemm fp32|render_image|training step" | tr '\n' '|'; echo
      done
    done
  done
done | tee $OUT/generic_gemm_ab_${AB_NAME:-run}.log
