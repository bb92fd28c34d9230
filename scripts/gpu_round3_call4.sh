mkdir -p gpurun_out
python -m pytest tests/test_gpu_fp8_dumps.py -x -q -s 2>&1 | tail -90 > gpurun_out/t_fp8.log
cat gpurun_out/t_fp8.log
for f in "" "--train-dumps fp8"; do
  echo "== train-ddp $f" ; python bench.py --mode train-ddp --steps 20 --warmup 5 $f 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done 2>&1 | tee gpurun_out/bench_fp8.log
