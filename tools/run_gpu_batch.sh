#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/tl
timeout 600 python -m pytest tests/test_pv.py -m gpu -q -x 2>&1 | tail -3
for C in 32768 65536; do for R in 4 6 8 12; do
  MELONIX_PV_ANALYSIS_RUN=$R MELONIX_PV_CHUNK_FRAMES=$C timeout 300 python tools/pv_ab.py 60 3 sweep 2>&1 | tail -1 | sed "s/^/C=$C run=$R /"
done; done
MELONIX_PV_ANALYSIS_RUN=8 timeout 300 python tools/pv_ab.py 60 3 rich 2>&1 | tail -1
C=32768
  MELONIX_PV_ANALYSIS_RUN=8 MELONIX_PV_CHUNK_FRAMES=$C timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/c$C -o pv -- python tools/pv_ab.py 60 3 sweep > gpurun_out/tl/c$C.log 2>&1
  f=$(find gpurun_out/tl/c$C -name "*kernel_trace.csv" | head -1)
  python tools/pv_timeline.py $f 60 > gpurun_out/tl/timeline_c$C.txt 2>&1
  tail -1 gpurun_out/tl/c$C.log
  cat gpurun_out/tl/timeline_c$C.txt
find gpurun_out/tl -name "*.db" -delete; find gpurun_out/tl -name "*kernel_trace.csv" -delete
