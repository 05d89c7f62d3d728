#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_guard.py -q -x 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_rccl.py -q -x -k "visible_device or start_up" 2>&1 | tail -15
python bench.py > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_check.json"))
print({k: d[k] for k in ("value", "ms_per_step", "outputs_ok", "library_src_sha")}, d["roofline"]["frac"], d["roofline"]["traffic_source"])
print(d["phase_vocoder_supplementary"])
cb = d["cpu_baseline"]
print({k: cb.get(k) for k in ("value", "cores", "resynth_samples_per_s", "step_seconds")}, cb.get("resynth"), d.get("gpu_over_cpu_step"), d.get("gpu_over_cpu_stft_pitch"))
PY
tail -3 gpurun_out/bench_check.err
