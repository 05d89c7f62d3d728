#!/bin/bash
# tools/profile_gpu.sh <tag> [bench args...] — run on the MI355X box (via gpurun) from the repo root.
# 1. rocprofv3 --kernel-trace --stats of the bench command  -> gpurun_out/prof_<tag>/
# 2. PMC passes (each in its own run, kernel-trace only)      -> gpurun_out/pmc_<tag>_<n>/
# Summaries are condensed by tools/summarize_prof.py into gpurun_out/prof_<tag>_summary.txt / .json
set -u
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out
# The profiled command is the default bench step (STFT+pitch launch + resynthesis launch) without the CPU baseline and the
# supplementary extras.  PROF_KERNEL=<substring> picks the kernel whose PMC rows are summarised (default stft_kernel;
# PROF_RESYNTH=1 = resynth_kernel_v).
NORES="--no-supplementary --no-noise-secondary --no-limiter-probe"; [ "${PROF_RESYNTH:-0}" = "1" ] && export PROF_KERNEL=${PROF_KERNEL:-resynth_kernel_v}
BENCH="python bench.py --steps 50 --warmup 10 --no-cpu-baseline $NORES $*"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o stft -- $BENCH > $OUT/prof_${TAG}_bench.log 2>&1
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $P -d $OUT/pmc_${TAG}_$i -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline $NORES $* > $OUT/pmc_${TAG}_$i.log 2>&1 || echo "pass $i failed" >> $OUT/pmc_${TAG}_fail.log
done
python tools/summarize_prof.py $TAG > $OUT/prof_${TAG}_summary.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
cat $OUT/prof_${TAG}_summary.txt
