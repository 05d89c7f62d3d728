#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pv.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3; do timeout 300 python tools/pv_ab.py 60 3 sweep 2>&1 | tail -1 | cut -c1-130; done
timeout 600 python tests/tools/pv8h_check.py 8 2 2>&1 | tail -3
