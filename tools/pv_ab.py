#!/usr/bin/env python3
"""tools/pv_ab.py — the build-defined phase vocoder (+3 st, 60 min) under one build of the library: sha1 of the f32 and
int16 outputs (bit-identity across builds: run it with MX_AB_LIB=<other .so> as well) and the time of the whole call."""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import melonix_amd as mx  # noqa: E402

if os.environ.get("MX_AB_LIB"):
    mx._capi.LIB_PATH = os.environ["MX_AB_LIB"]
from bench import SR, gen_shard  # noqa: E402

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
st = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
dev = torch.device("cuda", 0)
n = int(minutes * 60 * SR)
audio_t = gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
ctx = mx.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
audio = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
f32 = torch.empty(n, dtype=torch.float32, device=dev)
i16 = torch.empty(n, dtype=torch.int16, device=dev)
ts = []
for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.pv_pitch_shift_dev(audio, st, f32.data_ptr(), i16.data_ptr())
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
h32 = hashlib.sha1(f32.cpu().numpy().tobytes()).hexdigest()
h16 = hashlib.sha1(i16.cpu().numpy().tobytes()).hexdigest()
print(f"pv {minutes:g} min {st:+g} st [{os.environ.get('MX_AB_LIB', 'shipped')}]: call ms {min(ts[1:]):.2f} (runs {', '.join(f'{x:.2f}' for x in ts)}); "
      f"sha1 f32 {h32[:16]} i16 {h16[:16]}", flush=True)
