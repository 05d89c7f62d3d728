"""tests/tools/export_e2e.py — BASELINE configs[2] end to end on the GPU box: mx_export_wav of the 60-minute sweep at +3 st
(host f32 in -> upload, grain scan, schedule, gather-lerp + int16 kernel, D2H, WAV file on /tmp), with
MELONIX_TIMING=1 stage traces, next to the CPU oracle's export of the first minutes (App::exportWav's loop)."""
import os, sys, time
import numpy as np
import torch  # before the library: one HIP runtime per process
torch.cuda.init()
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import melonix_amd as mx
from oracle import pyoracle as O

SR = 48000
minutes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
w = O.sweep(minutes * 60 * SR)
n = len(w)
mk = [(1, 0, 0, 3.0), (n - 1, 0, 0, 3.0)]
ctx = mx.Context(0)
path = "/tmp/melonix_export.wav"
for rep in range(3):
    t0 = time.perf_counter()
    ctx.export_wav(w, SR, mk, path)
    dt = time.perf_counter() - t0
    print(f"mx_export_wav {minutes} min +3 st: {dt*1e3:.1f} ms, file {os.path.getsize(path)} bytes", flush=True)
os.remove(path)
# the oracle (CPU restatement of preproc + the exportWav loop + saveWav) on the first cpu_min minutes
cpu_min = min(minutes, 5)
wc = w[: cpu_min * 60 * SR]
mkc = [(1, 0, 0, 3.0), (len(wc) - 1, 0, 0, 3.0)]
t0 = time.perf_counter()
ex = O.export_run(wc, SR, mkc)
dt = time.perf_counter() - t0
print(f"oracle export {cpu_min} min: {dt:.2f} s -> {dt * minutes / cpu_min:.1f} s per {minutes} min (1 thread)")
