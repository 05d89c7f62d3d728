#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_pv.py -m gpu -q -x -k "random_lengths" 2>&1 | tail -5
