#!/bin/bash
# round 3, call 13: the fused Ref-NeRF / density-gradient chains (parity first, then A/B against the single-layer launches on one box)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "refnerf or ref_ or density_grad or get_grad or training or train_step or g17 or G17 or normal" 2>&1 | tail -4
python -m pytest tests/test_gpu_multiprocess.py tests/test_gpu_configs_train.py tests/test_gpu_ddp.py tests/test_gpu_fp8_dumps.py -x -q 2>&1 | tail -3
for v in BASE UNFUSED; do
  if [ $v = BASE ]; then unset NERF_AMD_LIB; else export NERF_AMD_LIB=$PWD/nerf_amd/ablate/libnerf_amd_$v.so; fi
  for n in 512 2048; do
    echo "== $v eager: $(python scripts/gpu_train_rate.py ref $n bf16 2>&1 | tail -1)"
    echo "== $v graph: $(python scripts/gpu_train_rate.py ref $n bf16 graph 2>&1 | tail -1)"
  done
  echo "== $v fp32: $(python scripts/gpu_train_rate.py ref 512 fp32 2>&1 | tail -1)"
done 2>&1 | tee gpurun_out/r03_refnerf_chains_ab.log
unset NERF_AMD_LIB
AB_LIST="BASE UNFUSED" CFG_LIST="ref_512_bf16" bash scripts/gpu_train_profile.sh 2>&1 | tail -40
