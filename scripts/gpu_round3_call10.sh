mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -k "fused_compositing" 2>&1 | tail -4
for i in 1 2 3; do
python bench.py --steps 10 --warmup 3 --no-train-rate --no-cpu-baseline --no-gemm-ref 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('two-launch', d['ms_per_step'], d['roofline']['ms_per_launch'])"
python bench.py --steps 10 --warmup 3 --no-train-rate --no-cpu-baseline --no-gemm-ref --fused 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused     ', d['ms_per_step'], d['roofline']['ms_per_launch'])"
done
