mkdir -p gpurun_out
python -m pytest tests/test_gpu_configs_train.py -x -q 2>&1 | tail -40 > gpurun_out/t_configs.log
python -m pytest tests/test_gpu_parity.py -x -q -k "ipe or contraction or train_step or training or refnerf_train" 2>&1 | tail -15 >> gpurun_out/t_configs.log
python -m pytest tests/test_gpu_training_psnr.py -x -q 2>&1 | tail -5 >> gpurun_out/t_configs.log
cat gpurun_out/t_configs.log
python scripts/psnr_seeds.py --modes fp32,bf16 --seeds 1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16 --iters 8000 --lr-mult 1 --hold 0.5 --held 2 --ckpts 3 > gpurun_out/psnr_hip_16seeds.log 2>&1
grep SUMMARY gpurun_out/psnr_hip_16seeds.log
