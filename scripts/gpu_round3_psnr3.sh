mkdir -p gpurun_out
python scripts/psnr_seeds.py --modes fp32,bf16 --seeds 33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64 --iters 8000 --lr-mult 1 --hold 0.5 --held 2 --ckpts 3 > gpurun_out/psnr_hip_fp32_bf16_seeds33-64.log 2>&1
python scripts/psnr_seeds.py --modes bf16-fp8dumps --seeds 17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48 --iters 8000 --lr-mult 1 --hold 0.5 --held 2 --ckpts 3 > gpurun_out/psnr_hip_fp8dumps_seeds17-48.log 2>&1
grep SUMMARY gpurun_out/psnr_hip_fp32_bf16_seeds33-64.log gpurun_out/psnr_hip_fp8dumps_seeds17-48.log
