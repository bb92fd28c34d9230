#!/bin/bash
# ONE parameterised entry for every GPU-box call of a round (replaces the per-call gpu_round3_call*.sh scripts):
#   gpurun --timeout T -- 'bash scripts/gpu_job.sh JOB [JOB ...]'      outputs under gpurun_out/$TAG (TAG defaults to r04)
# JOBs:  tests [pytest args]   -m gpu suite (log: tests.log)           bench [args]   default bench line -> bench_default.json
#        benchall              the round's bench lines (default, ref, fp32, strong, train variants)
#        profile               rocprofv3 kernel stats + PMC passes of the bench (scripts/gpu_round_profile.sh)
#        trainprof             rocprofv3 of the 16 384-ray training step (+ PMC=1 traffic passes)
#        ab                    AB_LIST / AB_REPS / BENCH_ARGS: scripts/gpu_ab.sh             py FILE [args]   run a script
#        psnr                  PSNR_SEEDS: the converging 20 000-iteration recipe, HIP fp32 + bf16 (scripts/psnr_seeds.py)
# A job's arguments end at the next job name; everything is logged, nothing aborts the rest.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${TAG:-r04}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
JOBS="tests bench benchall profile trainprof ab py psnr"
is_job() { for j in $JOBS; do [ "$1" = "$j" ] && return 0; done; return 1; }
while [ $# -gt 0 ]; do
  job=$1; shift; args=()
  while [ $# -gt 0 ] && ! is_job "$1"; do args+=("$1"); shift; done
  echo "=== $job ${args[*]} ($(date +%T))"
  case $job in
    tests)    timeout ${TESTS_TIMEOUT:-3000} python -m pytest tests -m gpu -q -x --durations=15 "${args[@]}" > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; tail -5 $OUT/tests.log ;;
    bench)    python bench.py "${args[@]}" > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json ;;
    benchall) python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
              python bench.py --model ref --no-cpu-baseline --no-gemm-ref > $OUT/bench_refnerf.json 2>/dev/null
              python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-ref --no-train-rate > $OUT/bench_fp32.json 2>/dev/null
              python bench.py --mode render-strong --steps 10 --warmup 3 > $OUT/bench_render_strong.json 2>/dev/null
              for f in "" "--ipe" "--contract" "--hipgraph" "--train-dumps fp8"; do python bench.py --mode train-ddp --steps 20 --warmup 5 --no-cpu-baseline $f 2>/dev/null | tail -1; done > $OUT/bench_train_variants.jsonl
              python bench.py --mode train-ddp --steps 50 --warmup 5 --train-rays 512 --hipgraph --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/bench_train_variants.jsonl
              python bench.py --mode train-ddp --model ref --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/bench_train_variants.jsonl
              tail -c 600 $OUT/bench_default.json ;;
    profile)  bash scripts/gpu_round_profile.sh > $OUT/round_profile.log 2>&1; tail -3 $OUT/round_profile.log ;;
    trainprof) PMC=${PMC:-1} bash scripts/gpu_train_profile.sh "${args[@]}" > $OUT/train_profile.log 2>&1; tail -3 $OUT/train_profile.log ;;
    ab)       bash scripts/gpu_ab.sh 2>&1 | tee $OUT/ab_${AB_NAME:-run}.log ;;
    py)       python "${args[@]}" 2>&1 | tee -a $OUT/py.log | tail -40 ;;
    psnr)     # the converging recipe of profiles/r04_psnr (40 x 40 views x 25, 512 rays, 32+64 samples, 20 000 iterations, lr x 3, hold 0.6): HIP fp32 + bf16
              python scripts/psnr_seeds.py --modes fp32,bf16 --seeds ${PSNR_SEEDS:-1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16} --size 40 --views 25 --held 1 --rays 512 \
                     --coarse 32 --fine 64 --iters 20000 --lr-mult 3 --hold 0.6 --ckpts 4 > $OUT/psnr_hip.log 2>&1; grep SUMMARY $OUT/psnr_hip.log ;;
  esac
done
