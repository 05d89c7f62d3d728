#!/bin/bash
# tools/run_gpu_batch.sh — what one gpurun call runs (rewritten per experiment):
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/run_gpu_batch.sh'
set -u
mkdir -p gpurun_out
echo "== pv tests"; timeout 1500 python -m pytest tests/test_pv.py -m gpu -x -q 2>&1 | tail -15
echo "== hour: compact vs full records"
python tools/pv_ab.py 60 3 sweep 2>&1 | grep "^pv " | tee gpurun_out/variants_r06_pv_records.log
MELONIX_PV_FULL_RECORDS=1 python tools/pv_ab.py 60 3 sweep 2>&1 | grep "^pv " | tee -a gpurun_out/variants_r06_pv_records.log
python tools/pv_ab.py 60 3 rich 2>&1 | grep "^pv " | tee -a gpurun_out/variants_r06_pv_records.log
MELONIX_PV_FULL_RECORDS=1 python tools/pv_ab.py 60 3 rich 2>&1 | grep "^pv " | tee -a gpurun_out/variants_r06_pv_records.log
echo "== all gpu tests"; MX_FAULT_LOG=gpurun_out/fault_sweep_r06_device.log MX_RCCL_LOG=gpurun_out/rccl_r06.log timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8
python tests/tools/pv8h_check.py 8 8 > gpurun_out/pv8h_r06.log 2>&1; tail -12 gpurun_out/pv8h_r06.log
echo "== bench"; python bench.py > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_check.json"))
pv = d["phase_vocoder_supplementary"]
print({k: d[k] for k in ("value", "ms_per_step", "outputs_ok", "library_src_sha")}, d["roofline"]["frac"], pv["call_ms"], pv["call_ms_runs"], pv["arena_bytes"], pv["chunks"], d.get("gpu_over_cpu_step"))
PY
