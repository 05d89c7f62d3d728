#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_pv.py tests/test_gpu_stft.py tests/test_gpu_guard.py -m gpu -q -x 2>&1 | tail -3
