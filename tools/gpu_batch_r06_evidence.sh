#!/bin/bash
# tools/gpu_batch_r06_evidence.sh — round 6, VERDICT item 2: configs[4] (N = 16384 / hop 512) and resynthesis evidence at HEAD,
# one box: rocprofv3 stats + PMC passes, then the bench line (cpu_baseline included) with roofline.traffic from that PMC set.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_batch_r06_evidence.sh'
set -u
mkdir -p gpurun_out
bash tools/profile_gpu.sh r06_16384x512 --fft 16384 --hop 512 > gpurun_out/profile_r06_16384x512.log 2>&1
cp gpurun_out/pmc_latest_r06_16384x512.json profiles/pmc_latest_16384x512.json
python bench.py --fft 16384 --hop 512 > gpurun_out/bench_r06_c4_16384x512.json 2> gpurun_out/bench_r06_c4_16384x512.err
PROF_RESYNTH=1 bash tools/profile_gpu.sh r06_resynth > gpurun_out/profile_r06_resynth.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r06_c4_16384x512.json"))
print({k: d[k] for k in ("value", "ms_per_step", "outputs_ok", "library_src_sha")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
PY
tail -12 gpurun_out/prof_r06_16384x512_summary.txt
tail -8 gpurun_out/prof_r06_resynth_summary.txt
rocm-smi --showpower --showclocks 2>/dev/null | head -20 > gpurun_out/r06_evidence_smi.txt
