#!/bin/bash
# tools/profile_pv.sh — rocprofv3 kernel stats of the phase-vocoder path (tests/tools/pv_check.py) on the GPU box
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_pv
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o pv -- python tests/tools/pv_check.py > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cut -c1-170 "$f" | head -14; else echo "no stats file"; tail -5 $OUT/run.log; fi
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete; find $OUT -name "*kernel_trace.csv" -delete
