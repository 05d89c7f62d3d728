#!/bin/bash
set -u
timeout 600 python -m pytest tests/test_pv.py -m gpu -q -x -k arena 2>&1 | tail -3
