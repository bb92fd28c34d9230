mkdir -p gpurun_out
AB_LIST="BASE NARROWF NARROWB" bash scripts/gpu_train_profile.sh 2>&1 | grep -E "==|mip_kernel|mip_bwd|proposal_kernel|prop_bwd" 
python scripts/gpu_train_rate.py 16384 bf16
NERF_AMD_LIB=$PWD/nerf_amd/ablate/libnerf_amd_NARROWF.so python scripts/gpu_train_rate.py 16384 bf16
NERF_AMD_LIB=$PWD/nerf_amd/ablate/libnerf_amd_NARROWB.so python scripts/gpu_train_rate.py 16384 bf16
NERF_AMD_LIB=$PWD/nerf_amd/ablate/libnerf_amd_NARROWF.so python -m pytest tests/test_gpu_parity.py -x -q -k "mlp_training or train_step_gradients" 2>&1 | tail -3
NERF_AMD_LIB=$PWD/nerf_amd/ablate/libnerf_amd_NARROWB.so python -m pytest tests/test_gpu_parity.py -x -q -k "mlp_training or train_step_gradients" 2>&1 | tail -3
