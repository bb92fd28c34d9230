mkdir -p gpurun_out
python scripts/psnr_seeds.py --modes bf16-fp8dumps --seeds 49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112 --iters 8000 --lr-mult 1 --hold 0.5 --held 2 --ckpts 3 > gpurun_out/psnr_hip_fp8dumps_seeds49-112.log 2>&1
grep SUMMARY gpurun_out/psnr_hip_fp8dumps_seeds49-112.log
