#!/bin/bash
# tools/run_gpu_batch.sh — what one gpurun call runs (rewritten per experiment):
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/run_gpu_batch.sh'
set -u
mkdir -p gpurun_out
echo "== all gpu tests"; MX_FAULT_LOG=gpurun_out/fault_sweep_r06_device.log MX_RCCL_LOG=gpurun_out/rccl_r06.log timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -25
python tests/tools/pv8h_check.py 8 8 > gpurun_out/pv8h_r06.log 2>&1; tail -12 gpurun_out/pv8h_r06.log
echo "== hour in K chunks"
for C in 0 401376 267584 200704 131072; do
  if [ $C = 0 ]; then python tools/pv_ab.py 60 3 sweep 2>&1 | grep "^pv "; else MELONIX_PV_CHUNK_FRAMES=$C python tools/pv_ab.py 60 3 sweep 2>&1 | grep "^pv "; fi
done | tee gpurun_out/variants_r06_pv_hour_chunks.log
echo "== bench"; python bench.py > gpurun_out/bench_r06_pv.json 2> gpurun_out/bench_r06_pv.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r06_pv.json"))
print({k: d[k] for k in ("value", "ms_per_step", "outputs_ok", "library_src_sha")}, d["roofline"]["frac"], d["phase_vocoder_supplementary"]["call_ms"], d.get("gpu_over_cpu_step"))
PY
