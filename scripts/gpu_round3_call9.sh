mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "training or train_step or narrower" 2>&1 | tail -2
python bench.py --mode train-ddp --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench_train_ddp.json; python -c "import json; d=json.load(open('gpurun_out/bench_train_ddp.json')); print('train-ddp', d['ms_per_step'], d['value'])"
python bench.py --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_launch']); t=d['train_step']; print({k:v for k,v in t.items() if k.startswith('rays')}); print(t['roofline']['frac'], t['iteration_512'], t['refnerf_rays_512'])"
