#!/bin/bash
# round 3, call 18: lean weight-ring bookkeeping (M0 not saved/restored, power-of-two slot wrap) -- parity, then A/B against the previous library
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "mlps or render_rays or render_image or refnerf_forward or training or train_step" 2>&1 | tail -3
AB_LIST="BASE PREV" AB_REPS=3 BENCH_ARGS="--no-gemm-ref --no-train-rate" bash scripts/gpu_ab.sh 2>&1 | tee gpurun_out/r03_lean_ring_ab.log
for v in BASE PREV; do
  if [ $v = BASE ]; then unset NERF_AMD_LIB; else export NERF_AMD_LIB=$PWD/nerf_amd/ablate/libnerf_amd_$v.so; fi
  echo "== $v train: $(python bench.py --mode train-ddp --no-cpu-baseline --no-gemm-ref 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"])')"
  echo "== $v ref: $(python bench.py --model ref --steps 5 --no-cpu-baseline --no-gemm-ref --no-train-rate 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["roofline"]["ms_per_launch"])')"
done 2>&1 | tee -a gpurun_out/r03_lean_ring_ab.log
