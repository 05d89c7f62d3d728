#!/bin/bash
set -u
export TMPDIR=/tmp
for rep in 1 2; do
timeout 300 python tools/pv_ab.py 60 3 sweep 2>&1 | tail -1 | cut -c1-130
for V in lock64 noprio lock64noprio; do
  MX_AB_LIB=melonix_amd/lib/variants/$V.so timeout 300 python tools/pv_ab.py 60 3 sweep 2>&1 | tail -1 | cut -c1-150
  MX_AB_LIB=melonix_amd/lib/variants/$V.so MELONIX_PV_SIDE_PRIO=0 timeout 300 python tools/pv_ab.py 60 3 sweep 2>&1 | tail -1 | cut -c1-150 | sed "s/^/streamprio0 /"
done
MELONIX_PV_SIDE_PRIO=0 timeout 300 python tools/pv_ab.py 60 3 sweep 2>&1 | tail -1 | cut -c1-130 | sed "s/^/streamprio0 /"
done
