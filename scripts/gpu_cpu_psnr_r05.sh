#!/bin/bash
# CPU-oracle seeds of round 5's 10 000-iteration recipe on the GPU BOX's host cores (the build container's 8 cores carry seeds 1-8):
# SEEDS processes side by side, THREADS each; PROGRESS lines every 500 iterations, so a run cut off by the call's limit still pairs with
# the HIP runs at every iteration it reached.   gpurun --timeout 9000 -- 'bash scripts/gpu_cpu_psnr_r05.sh'  -> gpurun_out/psnr_cpu_r05/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/psnr_cpu_r05; mkdir -p $OUT; cd $R
SEEDS=${SEEDS:-"9 10 11 12"}; THREADS=${THREADS:-8}; LIMIT=${LIMIT:-8400}
python -c "import torch" > /dev/null 2>&1
for s in $SEEDS; do
  OMP_NUM_THREADS=$THREADS MKL_NUM_THREADS=$THREADS PSNR_PROGRESS_EVERY=500 timeout $LIMIT python scripts/psnr_seeds.py --modes cpu --seeds $s --threads $THREADS \
      --size 40 --views 25 --held 1 --rays 512 --coarse 32 --fine 64 --iters 10000 --lr-mult 3 --hold 0.6 --ckpts 4 > $OUT/cpu_seed$s.log 2>&1 &
done
wait
grep -h "RESULT\|PROGRESS" $OUT/cpu_seed*.log | tail -12
