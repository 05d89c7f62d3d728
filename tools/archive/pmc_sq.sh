#!/bin/bash
# SQ counters of the STFT launch for one (fft, hop): pmc_sq.sh <tag> <fft> <hop>   (separate --pmc passes, kernel trace only)
# prints, per counter, the value of the largest stft_kernel dispatch; results under gpurun_out/pmc_sq_<tag>/
export TMPDIR=/tmp
tag=$1; fft=$2; hop=$3
out=gpurun_out/pmc_sq_$tag
mkdir -p $out
groups=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
        "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"
        "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LEVEL_WAVES")
i=0
for g in "${groups[@]}"; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $g -d $out/g$i -o pmc -- python bench.py --fft $fft --hop $hop --steps 2 --warmup 1 --no-cpu-baseline --no-resynth --no-supplementary > $out/g$i.log 2>&1
  i=$((i+1))
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
best = {}
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "stft_kernel" not in row["Kernel_Name"]:
            continue
        k = row["Counter_Name"]
        v = float(row["Counter_Value"])
        if v > best.get(k, (0, ""))[0]:
            best[k] = (v, row["Kernel_Name"][:90])
for k in sorted(best):
    print(f"{k:28s} {best[k][0]:16.0f}")
print("kernel:", next(iter(best.values()))[1] if best else "none")
PY
find $out -name "*.db" -delete; find $out -name "*.csv" -size +2M -delete
