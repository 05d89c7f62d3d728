#!/usr/bin/env python3
"""What would running the step's two launches (STFT + pitch, resynthesis) CONCURRENTLY on two streams give?  They are
independent (same audio in, different outputs), so a pipeline may overlap them; bench.py keeps them back to back so that
each launch's time is its own.  Prints ms per step both ways."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import melonix_amd as mx  # noqa: E402
from bench import SR, gen_shard  # noqa: E402

dev = torch.device("cuda", 0)
N, hop = 4096, 256
n = 60 * 60 * SR
F = mx.frame_count(n, hop)
audio_t = gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
c1, c2 = mx.Context(0), mx.Context(0)
c1.set_stream(s1.cuda_stream)
c2.set_stream(s2.cuda_stream)
a1 = c1.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
a2 = c2.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
band = mx.pitch_band(N, SR)
mags = torch.empty((F, N // 2), dtype=torch.float32, device=dev)
pitch = torch.empty((F, 2), dtype=torch.int32, device=dev)
gs, gl, gf = c2.grain_table_dev(a2)
steps_arr, total, _ = mx.schedule_build_table(n, SR, gs, gl, gf, [(1, 0, 0, 3.0), (n - 1, 0, 0, 3.0)])
d_steps = torch.from_numpy(steps_arr.view(np.uint8).copy()).to(dev)
pcm = torch.empty(total, dtype=torch.int16, device=dev)
torch.cuda.synchronize()


def stft(ctx, a):
    ctx.stft_hop_dev(a, N, hop, 0, F, mags.data_ptr(), pitch.data_ptr(), band=band)


def resynth(ctx, a):
    ctx.resynth_dev(a, d_steps.data_ptr(), len(steps_arr), total, None, pcm.data_ptr())


def run(overlap, steps=100, warm=20):
    import time
    for k in range(warm + steps):
        if k == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if overlap:
            stft(c1, a1)
            resynth(c2, a2)
        else:
            stft(c1, a1)
            resynth(c1, a1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(2):
    print("back to back  %.3f ms per step" % run(False), flush=True)
    print("two streams   %.3f ms per step" % run(True), flush=True)
