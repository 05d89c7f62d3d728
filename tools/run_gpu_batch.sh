#!/bin/bash
set -u
for rep in 1 2; do
timeout 300 python tools/pv_ab.py 60 3 sweep 2>&1 | tail -1 | cut -c1-110
for M in 256 128 128:2 64 64:4 32 32:8 16:16; do
  MELONIX_PV_SIDE_CUS=$M timeout 300 python tools/pv_ab.py 60 3 sweep 2>&1 | tail -1 | cut -c1-110 | sed "s/^/cus=$M /"
done; done
