export TMPDIR=/tmp
for P in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $P -d gpurun_out/pmc_big/$P -o pmc -- python bench.py --fft 32768 --hop 512 --steps 2 --warmup 1 --no-cpu-baseline --no-resynth > gpurun_out/pmc_big_$P.log 2>&1
done
python - <<'PY'
import csv, collections, re
for name in ("FETCH_SIZE","WRITE_SIZE"):
    best=collections.defaultdict(float)
    try:
        for row in csv.DictReader(open(f"gpurun_out/pmc_big/{name}/pmc_counter_collection.csv")):
            if row.get("Counter_Name")!=name: continue
            if "stft_kernel" in row["Kernel_Name"]: best["stft"]=max(best["stft"],float(row["Counter_Value"]))
    except Exception as e: print(e)
    print(name, {k: round(v*1024/1e9*(2 if name=="FETCH_SIZE" else 1),2) for k,v in best.items()})
PY
find gpurun_out/pmc_big -name "*.db" -delete; find gpurun_out/pmc_big -name "*.csv" -size +5M -delete
