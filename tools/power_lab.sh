#!/bin/bash
# tools/power_lab.sh [build|run] — energy attribution of the headline STFT kernel (see tools/power_lab.hip).
#   build (here, CPU container): one binary per variant under tools/bin/
#   run (GPU box): each variant for 3 s with rocm-smi power/sclk sampled every ~60 ms -> gpurun_out/r2/power_lab.log
VARIANTS="${LAB_VARIANTS:-base:-DDUMMY notw2read:-DMX_ABL_NOTW2READ nolds:-DMX_ABL_NOLDS nogstore:-DMX_ABL_NOGSTORE pitchonly:-DLAB_PITCH_ONLY novalu:-DMX_ABL_NOVALU noldsnovalu:-DMX_ABL_NOLDS,-DMX_ABL_NOVALU idle:-DLAB_IDLE}"
if [ "$1" = "build" ]; then
  mkdir -p tools/bin
  for v in $VARIANTS; do
    name=${v%%:*}; flags=$(echo ${v#*:} | tr ',' ' ')
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffp-contract=off -Wno-unused-value -I melonix_amd/csrc -I tools "-DABLNAME=\"$name\"" $flags tools/power_lab.hip -o tools/bin/plab_$name 2>&1 | grep -E "error" &
  done
  wait; ls tools/bin/plab_*
  exit 0
fi
OUT=gpurun_out/r2; mkdir -p $OUT
LOG=$OUT/power_lab${2:+_$2}.log; : > $LOG
for v in $VARIANTS; do
  name=${v%%:*}
  ( while true; do rocm-smi --showpower --showclocks --csv 2>/dev/null | grep -E "^(device|card0)"; sleep 0.05; done ) > $OUT/pl_$name.smi &
  SP=$!
  tools/bin/plab_$name 3 >> $LOG 2>&1
  kill $SP; wait $SP 2>/dev/null
  # drop the first and last quarter of the samples (ramp), average the rest
  python3 - "$name" $OUT/pl_$name.smi >> $LOG <<'PY'
import sys, re
name, path = sys.argv[1], sys.argv[2]
rows, hdr = [], None
for l in open(path):
    f = l.strip().split(",")
    if f[0] == "device":
        hdr = f
    elif f[0] == "card0" and hdr and len(f) == len(hdr):
        d = dict(zip(hdr, f))
        pw = [v for k, v in d.items() if "Power" in k]
        ck = [v for k, v in d.items() if k.startswith("sclk clock speed")]
        m = re.search(r"(\d+)", ck[0]) if ck else None
        if pw and m:
            rows.append((float(pw[0]), float(m.group(1))))
k = len(rows) // 4
mid = rows[k:len(rows) - k] if len(rows) >= 8 else rows
if mid:
    print("POWER %-12s %7.1f W  sclk %6.0f MHz  (%d samples, max %.0f W)" % (name, sum(p for p, _ in mid) / len(mid), sum(c for _, c in mid) / len(mid), len(mid), max(p for p, _ in rows)))
PY
  sleep 1
done
cat $LOG
