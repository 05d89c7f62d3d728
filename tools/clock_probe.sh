#!/bin/bash
# effective shader clock per kernel variant: GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / duration
export TMPDIR=/tmp
for b in stft_variants sv_nogstore; do
  rm -rf gpurun_out/clk_$b
  rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE -d gpurun_out/clk_$b -o clk -- tools/bin/$b 60 > /dev/null 2>&1
  python - "$b" <<'PY'
import csv, glob, sys, collections
b = sys.argv[1]
dur = {}
for f in glob.glob(f"gpurun_out/clk_{b}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/clk_{b}/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    for d, v in per.items():
        if d in dur and "stft_kernel" in dur[d][0]:
            name, ns = dur[d]
            acc[name[-70:]].append((v / 8 / ns, ns / 1e6))
for k, v in acc.items():
    print(b, k, "GHz %.3f  ms %.3f  (n=%d)" % (sum(x for x, _ in v) / len(v), sum(y for _, y in v) / len(v), len(v)))
PY
  find gpurun_out/clk_$b -name "*.db" -delete
done
