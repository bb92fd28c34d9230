#!/bin/bash
# profiles/r04_psnr/short6k/summary.md: the 6 000-iteration version of round 4's converging recipe, where CPU-oracle seeds fit a session
# (build container: one 2-thread process per seed, ~45 iterations a minute) -- CPU oracle against HIP fp32 / bf16 on the same seeds.
cd "$(dirname "$0")/.."
D=profiles/r04_psnr/short6k
{
  echo "# PSNR at equal iterations, CPU oracle beside the HIP paths (round 4, 6 000-iteration recipe)"
  echo
  echo "Recipe: \`scripts/psnr_seeds.py --size 40 --views 25 --held 1 --rays 512 --coarse 32 --fine 64 --iters 6000 --lr-mult 3 --hold 0.6 --ckpts 4\`"
  echo "(round 4's converging recipe cut to 6 000 iterations so that CPU-oracle seeds finish inside a session; a run's figure = mean of the held-out renders at"
  echo "iterations 5 700 / 5 800 / 5 900 / 6 000; identical initial weights, batches and uniforms per seed in every path).  CPU oracle: torch autograd fp32, one 2-thread"
  echo "process per seed in the build container; HIP: the GPU box."
  echo
  python scripts/psnr_summary.py $D/*.log
} > $D/summary.md
cat $D/summary.md
