#!/bin/bash
# profiles/r04_psnr/summary.md from the RESULT lines of the round-4 converging recipe (CPU oracle seeds + HIP fp32 / bf16 seeds)
cd "$(dirname "$0")/.."
{
  echo "# PSNR on a recipe that converges (round 4)"
  echo
  echo "Recipe: \`scripts/psnr_seeds.py --size 40 --views 25 --held 1 --rays 512 --coarse 32 --fine 64 --iters 20000 --lr-mult 3 --hold 0.6 --ckpts 4\`"
  echo "(24 training views of 40 x 40 + 1 held-out view of the analytic scene, 512 rays per batch, 32 + 64 samples, the reference's learning-rate rule x 3 held for"
  echo "60 % of the run, then its 100x decay; a run's figure = mean of the held-out renders at iterations 19 700 / 19 800 / 19 900 / 20 000; identical initial weights,"
  echo "batches and uniforms per seed in every path).  CPU oracle: one 2-thread process per seed in the build container (resumable); HIP: \`bash scripts/gpu_job.sh psnr\`."
  echo
  python scripts/psnr_summary.py profiles/r04_psnr/*.log
  echo
  echo "Round 2's single CPU-oracle run of the same recipe (seed 7, \`profiles/r02_psnr_20k_cpu_seed7.log\`, renders every 500 iterations): 31.87 dB (mean of the last four)."
} > profiles/r04_psnr/summary.md
cat profiles/r04_psnr/summary.md
