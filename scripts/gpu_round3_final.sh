#!/bin/bash
# Round-3 evidence run: render bench under rocprofv3 (kernel stats + four PMC passes), training-step profile with HBM traffic passes,
# Ref-NeRF step, and the unprofiled bench lines.  Outputs under gpurun_out/ (summarised into profiles/ by the two summarize_* scripts).
mkdir -p gpurun_out
bash scripts/gpu_round_profile.sh > gpurun_out/round_profile.log 2>&1
PMC=1 bash scripts/gpu_train_profile.sh > gpurun_out/train_profile.log 2>&1
cp gpurun_out/trainprof/train_pmc_summary.md gpurun_out/trainprof/train_pmc_summary_bf16dumps.md
NERF_AMD_TRAIN_DUMPS=fp8 bash scripts/gpu_train_profile.sh > gpurun_out/train_profile_fp8.log 2>&1
cp gpurun_out/trainprof/BASE_16384_bf16_kernel_stats.csv gpurun_out/trainprof/fp8dumps_final_kernel_stats.csv
CFG_LIST=ref_512_bf16 bash scripts/gpu_train_profile.sh > gpurun_out/train_profile_ref.log 2>&1
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
python bench.py --model ref --no-cpu-baseline --no-gemm-ref > gpurun_out/r03_bench_refnerf.json 2>/dev/null
python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-ref --no-train-rate > gpurun_out/r03_bench_fp32.json 2>/dev/null
python bench.py --mode render-strong --steps 10 --warmup 3 > gpurun_out/r03_bench_render_strong.json 2>/dev/null
for f in "" "--ipe" "--contract" "--hipgraph" "--train-dumps fp8"; do python bench.py --mode train-ddp --steps 20 --warmup 5 $f 2>/dev/null | tail -1; done > gpurun_out/r03_bench_train_variants.jsonl
python bench.py --mode train-ddp --steps 50 --warmup 5 --train-rays 512 --hipgraph 2>/dev/null | tail -1 >> gpurun_out/r03_bench_train_variants.jsonl
tail -c 600 gpurun_out/r03_bench_default.json
