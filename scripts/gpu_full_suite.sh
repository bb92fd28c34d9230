mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu -s 2>&1 | grep -v "^\[Gloo\]\|amdgpu.ids\|^\[W9" | tail -150 > gpurun_out/full_suite.log
tail -60 gpurun_out/full_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
