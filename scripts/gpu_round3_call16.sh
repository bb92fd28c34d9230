#!/bin/bash
# round 3, call 16: software-pipelined wgrad256 (parity, then A/B against the un-pipelined body: bf16 and fp8 dumps, one box)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "training or train_step or backward or g14 or G14 or g17 or G17 or refnerf" 2>&1 | tail -3
python -m pytest tests/test_gpu_configs_train.py tests/test_gpu_fp8_dumps.py -x -q 2>&1 | tail -3
for v in BASE NOPIPE; do
  if [ $v = BASE ]; then unset NERF_AMD_LIB; else export NERF_AMD_LIB=$PWD/nerf_amd/ablate/libnerf_amd_$v.so; fi
  for d in bf16 fp8; do
    echo "== $v dumps $d: $(python bench.py --mode train-ddp --train-dumps $d --no-cpu-baseline --no-gemm-ref 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["value"])')"
  done
done 2>&1 | tee gpurun_out/r03_wgrad_pipelined_ab.log
unset NERF_AMD_LIB
AB_LIST="BASE NOPIPE" CFG_LIST="16384_bf16" bash scripts/gpu_train_profile.sh 2>&1 | grep "==\|wgrad256\|mip_kernel\|mip_bwd"
NERF_AMD_TRAIN_DUMPS=fp8 AB_LIST="BASE NOPIPE" CFG_LIST="16384_bf16" OUTTAG=fp8 bash scripts/gpu_train_profile.sh 2>&1 | grep "==\|wgrad256\|mip_kernel\|mip_bwd"
