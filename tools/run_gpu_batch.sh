#!/bin/bash
# tools/run_gpu_batch.sh — what one gpurun call runs (rewritten per experiment; this is the round's closing check):
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/run_gpu_batch.sh'
set -u
mkdir -p gpurun_out
MX_FAULT_LOG=gpurun_out/fault_sweep_r06_device.log MX_RCCL_LOG=gpurun_out/rccl_r06.log timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5
python bench.py > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_check.json"))
pv = d["phase_vocoder_supplementary"]
print({k: d[k] for k in ("value", "ms_per_step", "outputs_ok", "library_src_sha")}, d["roofline"]["frac"], d["roofline"]["traffic"], pv["call_ms"], pv["arena_bytes"], pv["chunks"], d.get("gpu_over_cpu_step"))
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
