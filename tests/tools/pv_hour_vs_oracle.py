#!/usr/bin/env python3
"""tests/tools/pv_hour_vs_oracle.py [minutes] [semitones] — the build-defined phase vocoder over BASELINE configs[2]'s hour
against its definition (oracle/pv_oracle.py, numpy, binary64) at full size: 802 716 frames, 172.8 M samples.  The suite compares
with the oracle up to ten minutes (tests/test_pv.py) and checks the hour by properties, shape independence and a sha1 pinned to
the build's own earlier output; this tool closes the loop once per round — the hour's GPU output (its sha1 printed: the pinned
one) within 2e-5 of the oracle's.  Needs ~45 GB of host memory and a few minutes of one CPU core; kept out of the suite for that.
PARITY UNPINNED all the same: the oracle is this build's own definition, the reference has no phase vocoder."""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import melonix_amd as mx  # noqa: E402
from bench import SR, gen_shard  # noqa: E402
from oracle import pv_oracle as pv  # noqa: E402

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
ctx.pv_pitch_shift_dev(audio, st, f32.data_ptr(), i16.data_ptr())
torch.cuda.synchronize()
g = f32.cpu().numpy()
sha32 = hashlib.sha1(g.tobytes()).hexdigest()[:16]
sha16 = hashlib.sha1(i16.cpu().numpy().tobytes()).hexdigest()[:16]
x = audio_t[mx.MX_AUDIO_PAD:mx.MX_AUDIO_PAD + n].cpu().numpy().astype(np.float64)
ctx.release_scratch()
print(f"pv {minutes:g} min {st:+g} st on the GPU: sha1 f32 {sha32} i16 {sha16}; oracle running ...", flush=True)
t0 = time.perf_counter()
ref = pv.pitch_shift(x, st)
t_or = time.perf_counter() - t0
err = np.abs(g.astype(np.float64) - ref)
worst = int(err.argmax())
print(f"oracle: {t_or:.0f} s on one core; max |gpu - oracle| = {err.max():.3e} at sample {worst} (t = {worst / SR:.2f} s), rms {np.sqrt((err ** 2).mean()):.3e}; "
      f"tolerance 2e-5 of full scale: {'within' if err.max() <= 2e-5 else 'EXCEEDED'}", flush=True)
assert g.shape == ref.shape and err.max() <= 2e-5
print("pv_hour_vs_oracle ok")
