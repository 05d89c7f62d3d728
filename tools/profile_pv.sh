#!/bin/bash
# tools/profile_pv.sh <tag> [signal] — the phase-vocoder call (tools/pv_ab.py, 60 min, +3 st) under rocprofv3 on the GPU box:
# kernel-trace stats, then separate --pmc passes (instruction counts, wait states, HBM bytes), one line per kernel.
# STATS_ONLY=1: the kernel-trace stats alone; MX_AB_LIB=<variant .so>: that build (tools/ab_variant.sh).
set -u
export TMPDIR=/tmp
TAG=${1:-r04}; SIG=${2:-sweep}
OUT=gpurun_out/prof_pv_$TAG
rm -rf $OUT; mkdir -p $OUT
python tools/pv_ab.py 60 3 $SIG > $OUT/plain.log 2>&1; grep "^pv" $OUT/plain.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o pv -- python tools/pv_ab.py 60 3 $SIG > $OUT/stats.log 2>&1
i=0
[ "${STATS_ONLY:-0}" = "1" ] || for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc $P -d $OUT/pmc$i -o pmc -- python tools/pv_ab.py 60 3 $SIG > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
out = sys.argv[1]
short = lambda k: k.replace("mx::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
print("== kernel stats (4 calls per run)")
if st:
    for row in csv.DictReader(open(st[0])):
        if "pv_" in row["Name"]:
            print(f"{short(row['Name']):28s} calls {row['Calls']:>4s}  avg {float(row['AverageNs'])/1e6:8.3f} ms  total {float(row['TotalDurationNs'])/1e6:8.2f} ms  {row['Percentage']} %")
print("== counters per launch (max over the launches of a kernel; FETCH doubled per MI355X_MICROARCH.md)")
vals = collections.defaultdict(dict)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if not k.startswith("pv_"):
            continue
        c = row["Counter_Name"]
        vals[k][c] = max(vals[k].get(c, 0.0), float(row["Counter_Value"]))
for k, d in sorted(vals.items()):
    w = d.get("SQ_WAVES", 0) or 1
    print(f"{k:28s} waves {w:9.0f}  VALU/wave {d.get('SQ_INSTS_VALU',0)/w:8.1f}  SALU/wave {d.get('SQ_INSTS_SALU',0)/w:7.1f}  LDS/wave {d.get('SQ_INSTS_LDS',0)/w:7.1f}  "
          f"VMEM rd/wr per wave {d.get('SQ_INSTS_VMEM_RD',0)/w:6.1f}/{d.get('SQ_INSTS_VMEM_WR',0)/w:6.1f}  wave-cycles {d.get('SQ_WAVE_CYCLES',0):.3g}  "
          f"wait_any {d.get('SQ_WAIT_ANY',0)/max(d.get('SQ_WAVE_CYCLES',1),1):.2f}  wait_inst {d.get('SQ_WAIT_INST_ANY',0)/max(d.get('SQ_WAVE_CYCLES',1),1):.2f}  "
          f"active {d.get('SQ_ACTIVE_INST_ANY',0)/max(d.get('SQ_WAVE_CYCLES',1),1):.2f}  "
          f"FETCH {2*d.get('FETCH_SIZE',0)*1024/1e9:6.2f} GB  WRITE {d.get('WRITE_SIZE',0)*1024/1e9:6.2f} GB  gui_active {d.get('GRBM_GUI_ACTIVE',0):.3g}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
