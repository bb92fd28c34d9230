mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "training or train_step or mlps_fp32 or render_rays_fp32 or refnerf_forward" 2>&1 | tail -2
AB_LIST="PREV BASE" bash scripts/gpu_train_profile.sh 2>&1 | grep -E "==|mip_kernel|mip_bwd|proposal_kernel|prop_bwd|wgrad256"
for i in 1 2; do
NERF_AMD_LIB=$PWD/nerf_amd/ablate/libnerf_amd_PREV.so python bench.py --steps 10 --warmup 3 --no-train-rate --no-cpu-baseline --no-gemm-ref 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PREV', d['ms_per_step'], d['roofline']['ms_per_launch'])"
python bench.py --steps 10 --warmup 3 --no-train-rate --no-cpu-baseline --no-gemm-ref 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NEW ', d['ms_per_step'], d['roofline']['ms_per_launch'])"
done
