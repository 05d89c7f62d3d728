#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python tests/tools/pv8h_check.py 8 2 2>&1 | tail -8
timeout 300 python tools/pv_ab.py 60 3 sweep 2>&1 | tail -1
timeout 300 python tools/pv_ab.py 60 3 rich 2>&1 | tail -1
timeout 300 python tools/pv_ab.py 60 24 sweep 2>&1 | tail -1
timeout 300 python tools/pv_ab.py 60 -12 sweep 2>&1 | tail -1
