#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -s -k "mfma_stream or normal_losses" 2>&1 | tail -5
