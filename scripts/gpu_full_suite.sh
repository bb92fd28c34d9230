mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu -s 2>&1 | grep -v "^\[Gloo\]\|amdgpu.ids\|^\[W9" | tail -150 > gpurun_out/full_suite.log
tail -60 gpurun_out/full_suite.log
