#!/bin/bash
# tools/profile_ranges.sh <tag> — rocprofv3 kernel-trace stats + separate PMC passes of tools/ranges_screen.py (the ranges-mode
# N = 32768 kernel with the fused colormap on 1280-column batches); run on the MI355X box from the repo root.
set -u
TAG=${1:-r03_32768_ranges}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
python tools/ranges_screen.py > $OUT/${TAG}_timing.json 2>$OUT/${TAG}_timing.err
cat $OUT/${TAG}_timing.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o rng -- python tools/ranges_screen.py > $OUT/prof_${TAG}.log 2>&1
i=0
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $P -d $OUT/pmc_${TAG}_$i -o pmc -- python tools/ranges_screen.py 1280 375 12 > $OUT/pmc_${TAG}_$i.log 2>&1 || echo "pass $i failed"
done
PROF_KERNEL="Plan<32768, 32>, 2," python tools/summarize_prof.py $TAG > $OUT/prof_${TAG}_summary.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
cat $OUT/prof_${TAG}_summary.txt
