#!/bin/bash
# tools/pmc_variant.sh "<sizes>" <lib|shipped> ... — HBM write/fetch bytes per STFT launch (rocprofv3 --pmc, one counter per
# pass) of the shipped library or of variant builds (tools/ab_variant.sh), via tools/stft_sizes.py.  Run on the GPU box.
export TMPDIR=/tmp
sizes="$1"; shift
mkdir -p gpurun_out/pmc_var
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  for P in WRITE_SIZE FETCH_SIZE; do
    d=gpurun_out/pmc_var/${tag}_$P
    rm -rf "$d"
    if [ "$lib" = shipped ]; then unset MX_AB_LIB; else export MX_AB_LIB="$lib"; fi
    MX_WARM=2 MX_REPS=3 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $P -d "$d" -o pmc -- python tools/stft_sizes.py $sizes > "$d.log" 2>&1
  done
  python - "$tag" <<'PY'
import csv, collections, glob, sys
tag = sys.argv[1]
for name in ("WRITE_SIZE", "FETCH_SIZE"):
    best = collections.defaultdict(float)
    for f in glob.glob(f"gpurun_out/pmc_var/{tag}_{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != name or "stft_kernel" not in row["Kernel_Name"]:
                continue
            k = row["Kernel_Name"].split("Plan<")[1].split(">")[0] + " " + ("ovl" if "_ovl" in row["Kernel_Name"] else "")
            best[k] = max(best[k], float(row["Counter_Value"]))
    # WRITE_SIZE / FETCH_SIZE count KiB; FETCH_SIZE on gfx950 counts 2x units (MI355X_MICROARCH.md, HBM section)
    print(tag, name, {k: round(v * 1024 / 1e9 * (2 if name == "FETCH_SIZE" else 1), 3) for k, v in best.items()}, "GB per launch")
PY
done
find gpurun_out/pmc_var -name "*.db" -delete; find gpurun_out/pmc_var -name "*.csv" -size +5M -delete
