#!/bin/bash
# tools/pmc_pv.sh — HBM traffic counters of the phase-vocoder kernels (separate --pmc passes, kernel trace only)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/pmc_pv
mkdir -p $OUT
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc $P -d $OUT/$P -o pmc -- python tests/tools/pv_check.py > $OUT/$P.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_pv/{name}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(name, "no counter file"); continue
    best = collections.defaultdict(float)
    for row in csv.DictReader(open(f[0])):
        if row.get("Counter_Name") != name: continue
        k = row["Kernel_Name"].split("(")[0].replace("mx::(anonymous namespace)::", "").replace("void ", "")
        best[k] = max(best[k], float(row["Counter_Value"]))
    for k, v in sorted(best.items()):
        if k.startswith("pv_"):
            print(f"{name} {k}: raw {v:.4g} KiB units -> {v*1024/1e9:.2f} GB" + (f" (x2 gfx950 correction: {2*v*1024/1e9:.2f} GB)" if name == "FETCH_SIZE" else ""))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete; find $OUT -name "*kernel_trace*" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
