mkdir -p gpurun_out
python -m pytest tests/test_gpu_configs_train.py -x -q -k "config4" 2>&1 | tail -15 > gpurun_out/t_call2.log
python -m pytest tests/test_gpu_ddp.py -q 2>&1 | tail -60 >> gpurun_out/t_call2.log
python -m pytest tests/test_gpu_multiprocess.py -q 2>&1 | tail -15 >> gpurun_out/t_call2.log
cat gpurun_out/t_call2.log
for f in "" "--ipe" "--contract" "--ipe --contract" "--hipgraph" "--ipe --hipgraph"; do
  echo "== train-ddp $f" ; python bench.py --mode train-ddp --steps 10 --warmup 3 $f 2>&1 | tail -2
done > gpurun_out/bench_train_variants.log 2>&1
python bench.py --mode train-ddp --steps 20 --warmup 3 --train-rays 512 --hipgraph 2>&1 | tail -1 >> gpurun_out/bench_train_variants.log
python bench.py --mode render-strong --steps 5 --warmup 2 2>&1 | tail -1 >> gpurun_out/bench_train_variants.log
cat gpurun_out/bench_train_variants.log
