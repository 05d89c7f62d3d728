#!/bin/bash
# tools/run_gpu_batch.sh — what one gpurun call runs (rewritten per experiment):
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/run_gpu_batch.sh'
set -u
mkdir -p gpurun_out
echo "== pv tests"; timeout 1500 python -m pytest tests/test_pv.py -m gpu -x -q 2>&1 | tail -15
echo "== pv full size"; timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "pv" 2>&1 | tail -15
python tests/tools/pv8h_check.py 8 8 > gpurun_out/pv8h_r06.log 2>&1; tail -14 gpurun_out/pv8h_r06.log
echo "== rccl"; MX_RCCL_LOG=gpurun_out/rccl_r06.log timeout 1500 python -m pytest tests/test_gpu_rccl.py -q -k "world1 or eight_ranks" 2>&1 | tail -15
echo "== bench"; python bench.py > gpurun_out/bench_r06_pv.json 2> gpurun_out/bench_r06_pv.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r06_pv.json"))
print({k: d[k] for k in ("value", "ms_per_step", "outputs_ok", "library_src_sha")}, d["roofline"]["frac"], d["phase_vocoder_supplementary"], d.get("gpu_over_cpu_step"))
PY
echo "== evidence"; bash tools/gpu_batch_r06_evidence.sh 2>&1 | tail -30
