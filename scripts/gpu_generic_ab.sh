#!/bin/bash
# Generic-path GEMM: XCD-aware tile order (shipped) against the former plain 3-D grid order (nerf_amd/ablate/libnerf_amd_PLAINGRID.so, built with
# -DGK_PLAIN_GRID), same box, alternated; the three stride forms at 262 144 x W x W (scripts/gpu_generic_rate.py prints them first).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=$R/gpurun_out/${TAG:-r04}; mkdir -p $OUT; cd $R
for rep in 1 2; do
  for v in PLAINGRID BASE; do
    if [ $v = BASE ]; then unset NERF_AMD_LIB; else export NERF_AMD_LIB=$R/nerf_amd/ablate/libnerf_amd_$v.so; fi
    for w in ${WIDTHS:-512 320}; do
      echo -n "$v width $w: "; python scripts/gpu_generic_rate.py $w 2>/dev/null | grep -E "^gemm bf16|^gemm fp32|render_image|training step" | tr '\n' '|'; echo
    done
  done
done | tee $OUT/generic_gemm_xcd_ab.log
