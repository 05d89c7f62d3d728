#!/usr/bin/env python3
"""tools/pv_ab.py [minutes] [semitones] [sweep|rich] — the build-defined phase vocoder (+3 st, 60 min) under one build of
the library: sha1 of the f32 and int16 outputs (bit-identity across builds: run it with MX_AB_LIB=<other .so> as well) and
the time of the whole call.  `rich`: the workload sweep plus eleven harmonics at 1/h and 1e-3 * U(-1,1) noise — a spectrum
with many peaks per frame (the sweep alone has a handful), what the phase-locking sweeps cost on music-like input."""
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
signal = sys.argv[3] if len(sys.argv) > 3 else "sweep"
dev = torch.device("cuda", 0)
n = int(minutes * 60 * SR)
audio_t = gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
if signal == "rich":
    import numpy as np

    from bench import add_noise

    pad = mx.MX_AUDIO_PAD
    T = n / SR
    chunk = 1 << 24
    for c in range(0, n, chunk):
        m = min(chunk, n - c)
        t = torch.arange(c, c + m, dtype=torch.float64, device=dev) / SR
        ph = 110.0 * t + (1760.0 - 110.0) * t * t / (2 * T)
        acc = torch.zeros(m, dtype=torch.float64, device=dev)
        for h in range(2, 13):
            acc += (0.25 / h) * torch.sin(2 * np.pi * h * ph)
        audio_t[pad + c:pad + c + m] = (audio_t[pad + c:pad + c + m].to(torch.float64) * 0.5 + acc).to(torch.float32)
    add_noise(torch, dev, audio_t, 0, 1, n, pad)
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
print(f"pv {signal} {minutes:g} min {st:+g} st [{os.environ.get('MX_AB_LIB', 'shipped')}]: call ms {min(ts[1:]):.2f} (runs {', '.join(f'{x:.2f}' for x in ts)}); "
      f"sha1 f32 {h32[:16]} i16 {h16[:16]}; arena {ctx.pv_arena_bytes() / 1e9:.2f} GB of a budget of {ctx.pv_arena_budget() / 1e9:.1f} GB, "
      f"{ctx.pv_last_chunks()} chunk(s)", flush=True)
