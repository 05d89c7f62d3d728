#!/bin/bash
# tools/run_gpu_batch.sh — what one gpurun call runs (rewritten per experiment; this is the round's closing check):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/run_gpu_batch.sh'
set -u
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3
python bench.py > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_check.json"))
print({k: d[k] for k in ("value", "ms_per_step", "outputs_ok")}, d["roofline"]["frac"], d["phase_vocoder_supplementary"]["call_ms"])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
