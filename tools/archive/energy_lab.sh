#!/bin/bash
# tools/energy_lab.sh (GPU box) — energy per operation of the building blocks the STFT kernel is made of: each stream of
# tools/ubench_issue.hip runs alone at full occupancy for 3 s while rocm-smi samples the package power.
OUT=gpurun_out/r2; mkdir -p $OUT; LOG=$OUT/energy_lab.log; : > $LOG
for k in idle fma pkfma ldsw64 ldsr64 ldsw32 ldsr128 gstore; do
  ( while true; do rocm-smi --showpower --showclocks --csv 2>/dev/null | grep -E "^(device|card0)"; sleep 0.05; done ) > $OUT/en_$k.smi &
  SP=$!
  tools/bin/ubench_issue power $k 3 >> $LOG 2>&1
  kill $SP; wait $SP 2>/dev/null
  python3 - "$k" $OUT/en_$k.smi >> $LOG <<'PY'
import sys, re
name, path = sys.argv[1], sys.argv[2]
rows, hdr = [], None
for l in open(path):
    f = l.strip().split(",")
    if f[0] == "device": hdr = f
    elif f[0] == "card0" and hdr and len(f) == len(hdr):
        d = dict(zip(hdr, f))
        pw = [v for k, v in d.items() if "Power" in k]; ck = [v for k, v in d.items() if k.startswith("sclk clock speed")]
        m = re.search(r"(\d+)", ck[0]) if ck else None
        if pw and m: rows.append((float(pw[0]), float(m.group(1))))
k = len(rows) // 4
mid = sorted(p for p, _ in rows[k:len(rows) - k]) if len(rows) >= 8 else sorted(p for p, _ in rows)
if mid: print("POWER  %-8s median %7.1f W (max %.0f), sclk median %.0f MHz, %d samples" % (name, mid[len(mid)//2], max(p for p, _ in rows), sorted(c for _, c in rows)[len(rows)//2], len(rows)))
PY
  sleep 1
done
cat $LOG
