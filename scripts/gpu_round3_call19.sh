#!/bin/bash
# round 3, call 19: the default bench line with the measured MFMA-stream ceiling (roofline.mfma_stream_ref)
mkdir -p gpurun_out
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
tail -c 1500 gpurun_out/r03_bench_default.json
