mkdir -p gpurun_out
python scripts/psnr_seeds.py --modes bf16-fp8dumps --seeds 1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16 --iters 8000 --lr-mult 1 --hold 0.5 --held 2 --ckpts 3 > gpurun_out/psnr_hip_fp8dumps_seeds1-16.log 2>&1
python scripts/psnr_seeds.py --modes fp32,bf16 --seeds 17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32 --iters 8000 --lr-mult 1 --hold 0.5 --held 2 --ckpts 3 > gpurun_out/psnr_hip_fp32_bf16_seeds17-32.log 2>&1
grep SUMMARY gpurun_out/psnr_hip_fp8dumps_seeds1-16.log gpurun_out/psnr_hip_fp32_bf16_seeds17-32.log
