mkdir -p gpurun_out
python scripts/psnr_seeds.py --modes fp32,bf16 --seeds 65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112 --iters 8000 --lr-mult 1 --hold 0.5 --held 2 --ckpts 3 > gpurun_out/psnr_hip_fp32_bf16_seeds65-112.log 2>&1
grep SUMMARY gpurun_out/psnr_hip_fp32_bf16_seeds65-112.log
