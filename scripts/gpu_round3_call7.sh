mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "training or train_step or narrower" 2>&1 | tail -3
python -m pytest tests/test_gpu_fp8_dumps.py tests/test_gpu_configs_train.py tests/test_gpu_ddp.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_multiprocess.py -x -q -k "oracles_step" 2>&1 | tail -3
bash scripts/gpu_train_profile.sh 2>&1 | tail -13
cp gpurun_out/trainprof/BASE_16384_bf16_kernel_stats.csv gpurun_out/trainprof/bf16dumps_8w_kernel_stats.csv
NERF_AMD_TRAIN_DUMPS=fp8 bash scripts/gpu_train_profile.sh 2>&1 | tail -6
cp gpurun_out/trainprof/BASE_16384_bf16_kernel_stats.csv gpurun_out/trainprof/fp8dumps_8w_kernel_stats.csv
AB_LIST="NODUMPST NOMASK" bash scripts/gpu_train_profile.sh 2>&1 | grep -E "==|mip_kernel"
