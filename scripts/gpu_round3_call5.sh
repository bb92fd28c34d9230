mkdir -p gpurun_out
python -m pytest tests/test_gpu_fp8_dumps.py -q -s 2>&1 | tail -90 > gpurun_out/t_fp8.log
tail -75 gpurun_out/t_fp8.log
bash scripts/gpu_train_profile.sh 2>&1 | tail -14
mv gpurun_out/trainprof/BASE_16384_bf16_kernel_stats.csv gpurun_out/trainprof/bf16dumps_kernel_stats.csv
NERF_AMD_TRAIN_DUMPS=fp8 bash scripts/gpu_train_profile.sh 2>&1 | tail -14
mv gpurun_out/trainprof/BASE_16384_bf16_kernel_stats.csv gpurun_out/trainprof/fp8dumps_kernel_stats.csv
