#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench line): resynthesis +3 st on 60 min (BASELINE
configs[2]), STFT at N=16384/hop 512 (configs[4]) and N=32768/hop 375 (the reference's own size),
grain scan.  Prints one JSON object per measurement."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import melonix_amd as mx  # noqa: E402
from bench import SR, b_alg, gen_shard  # noqa: E402

dev = torch.device("cuda", 0)
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n = int(minutes * 60 * SR)
n -= n % 512
audio_t = gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
ctx = mx.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
audio = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)


def timed(fn, reps=5, warm=4):  # (fresh multi-GB output buffers: the first launches also fault their pages in)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for N, hop in ((4096, 256), (4096, 512), (16384, 512), (16384, 1024), (32768, 1024), (32768, 375), (32768, 512), (4096, 375)):
    F = mx.frame_count(n, hop)
    if F * (N // 2) * 4 > 60e9:
        continue
    mags = torch.empty((F, N // 2), dtype=torch.float32, device=dev)
    pitch = torch.empty((F, 2), dtype=torch.int32, device=dev)
    band = mx.pitch_band(N, SR)
    ms = timed(lambda: ctx.stft_hop_dev(audio, N, hop, 0, F, mags.data_ptr(), pitch.data_ptr(), band=band))
    ms_p = timed(lambda: ctx.stft_hop_dev(audio, N, hop, 0, F, None, pitch.data_ptr(), band=band))
    ba = b_alg(N, hop)
    print(json.dumps({"what": "stft", "fft": N, "hop": hop, "frames": F, "ms": ms, "frames_per_s": F / ms * 1e3,
                      "alg_GBps": ba * F / ms / 1e6, "frac_of_8TBps": ba * F / ms / 1e6 / 8000, "pitch_only_ms": ms_p}))
    del mags, pitch
    torch.cuda.empty_cache()

# grain scan (zero-crossing bitmaps on the GPU + host chain walk) and schedule build
host = audio_t[mx.MX_AUDIO_PAD:mx.MX_AUDIO_PAD + n].cpu().numpy()
t0 = time.perf_counter()
gs, gl = ctx.grains_dev(audio)
t_gr = time.perf_counter() - t0
mk = [(1, 0, 0, 3.0), (n - 1, 0, 0, 3.0)]
t0 = time.perf_counter()
steps, total = mx.schedule_build(host, SR, gs, gl, mk)
t_sc = time.perf_counter() - t0
print(json.dumps({"what": "grains+schedule", "n": n, "grains": int(len(gs)), "grains_s": t_gr, "steps": int(len(steps)),
                  "schedule_s": t_sc, "pcm_samples": int(total)}))
d_steps = torch.from_numpy(steps.view(np.uint8)).to(dev)
pcm_f = torch.empty(total, dtype=torch.float32, device=dev)
pcm_i = torch.empty(total, dtype=torch.int16, device=dev)
for name, pf, pi in (("f32+i16", pcm_f, pcm_i), ("i16 only", None, pcm_i)):
    ms = timed(lambda: ctx.resynth_dev(audio, d_steps.data_ptr(), len(steps), total, pf.data_ptr() if pf is not None else None,
                                       pi.data_ptr()))
    bytes_per_sample = 4 * 1.189207 + 2 + (4 if pf is not None else 0)
    print(json.dumps({"what": "resynth +3st " + name, "samples": int(total), "ms": ms, "Msamples_per_s": total / ms / 1e3,
                      "hop256_frames_per_s": total / 256 / ms * 1e3, "alg_GBps": bytes_per_sample * total / ms / 1e6}))
