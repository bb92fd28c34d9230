python -m pytest tests/test_gpu_parity.py -q -k "other_ide_levels" 2>&1 | grep -E "AssertionError|passed|failed" | head
