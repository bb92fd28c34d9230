#!/bin/bash
# round 3, call 15: ReLU bit masks of the training forwards through v_pk_min_u16 (parity, then the 16384-ray step and its kernel stats)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "training or train_step or backward or g14 or G14 or g17 or G17" 2>&1 | tail -3
python -m pytest tests/test_gpu_configs_train.py tests/test_gpu_fp8_dumps.py tests/test_gpu_ddp.py -x -q 2>&1 | tail -3
python bench.py --mode train-ddp --no-cpu-baseline --no-gemm-ref 2>/dev/null | tail -1 | tee gpurun_out/r03_bench_train_ddp_16384_pkmin.json
python bench.py --mode train-ddp --train-dumps fp8 --no-cpu-baseline --no-gemm-ref 2>/dev/null | tail -1 | tee gpurun_out/r03_bench_train_ddp_16384_pkmin_fp8.json
CFG_LIST="16384_bf16" bash scripts/gpu_train_profile.sh 2>&1 | tail -16
