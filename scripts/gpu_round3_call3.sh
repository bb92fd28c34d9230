mkdir -p gpurun_out
./scripts/fp8_probe.bin > gpurun_out/fp8_probe.log 2>&1
cat gpurun_out/fp8_probe.log
python -m pytest tests/test_gpu_configs_train.py -x -q -k "config4_full" 2>&1 | tail -5
python -m pytest tests/test_gpu_ddp.py -q -k param_com 2>&1 | tail -5
