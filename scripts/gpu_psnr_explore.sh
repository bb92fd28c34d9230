#!/bin/bash
# recipe exploration for scripts/psnr_seeds.py: which (iterations, learning rate) gives a small seed-to-seed spread?
mkdir -p gpurun_out
{
python scripts/psnr_seeds.py --modes fp32 --seeds 1,2,3,4,5,6,7,8 --iters 6000 --lr-mult 1.5 --hold 0.4 --held 4 --ckpts 5
python scripts/psnr_seeds.py --modes fp32 --seeds 1,2,3,4,5,6,7,8 --iters 6000 --lr-mult 2 --hold 0.3 --held 4 --ckpts 5
python scripts/psnr_seeds.py --modes fp32 --seeds 1,2,3,4,5,6,7,8 --iters 8000 --lr-mult 1 --hold 0.5 --held 4 --ckpts 5
python scripts/psnr_seeds.py --modes fp32 --seeds 1,2,3,4,5,6,7,8 --iters 6000 --lr-mult 2 --hold 0.3 --held 4 --ckpts 5 --views 100
} > gpurun_out/psnr_explore2.log 2>&1
grep -E "recipe|SUMMARY" gpurun_out/psnr_explore2.log
