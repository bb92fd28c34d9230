#!/bin/bash
# round 3, call 17: fused Ref-NeRF normal losses, ragged Ref-NeRF backward; the Ref-NeRF step again
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "normal_losses or backward_ragged or refnerf or g17 or G17 or ref_ or get_grad" 2>&1 | tail -4
python -m pytest tests/test_gpu_multiprocess.py -x -q 2>&1 | tail -3
for n in 512 2048; do
  echo "== eager: $(python scripts/gpu_train_rate.py ref $n bf16 2>&1 | tail -1)"
  echo "== graph: $(python scripts/gpu_train_rate.py ref $n bf16 graph 2>&1 | tail -1)"
done 2>&1 | tee gpurun_out/r03_refnerf_step_call17.log
