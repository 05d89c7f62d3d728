#!/usr/bin/env python3
"""STFT launch time per (N, hop) on 60 min of synthetic audio: `stft_sizes.py 32768x375 16384x512 ...`
(one line per size: N hop ms frac-of-8TB/s).  A/B helper for kernel variants; bench_extra.py is the full table."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import melonix_amd as mx  # noqa: E402

if os.environ.get("MX_AB_LIB"):  # A/B against another build of the library (tools/ab_prev.sh)
    import ctypes

    mx._capi.LIB_PATH = os.environ["MX_AB_LIB"]
    _other = ctypes.CDLL(mx._capi.LIB_PATH)  # an older revision may lack entry points this tool does not use
    mx._capi.SIGNATURES = {k: v for k, v in mx._capi.SIGNATURES.items() if hasattr(_other, k)}
from bench import SR, PowerSampler, b_alg, gen_shard  # noqa: E402

dev = torch.device("cuda", 0)
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(32768, 375), (32768, 1024), (16384, 512)]
n = 60 * 60 * SR
audio_t = gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
ctx = mx.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
audio = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
for N, hop in sizes:
    F = mx.frame_count(n, hop)
    mags = torch.empty((F, N // 2), dtype=torch.float32, device=dev)
    pitch = torch.empty((F, 2), dtype=torch.int32, device=dev)
    band = mx.pitch_band(N, SR)
    pp = 0 if os.environ.get("MX_NO_PITCH") else pitch.data_ptr()  # MX_NO_PITCH=1: rows only (what the pitch pick costs)
    fn = lambda: ctx.stft_hop_dev(audio, N, hop, 0, F, mags.data_ptr(), pp, band=band)  # noqa: E731
    # (fresh output buffers fault their pages in, and a fresh box ramps for ~25 launches before the power manager settles)
    WARM, REPS = int(os.environ.get("MX_WARM", 40)), int(os.environ.get("MX_REPS", 150))
    for _ in range(WARM):
        fn()
    torch.cuda.synchronize()
    with PowerSampler(0) as ps:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REPS):
            fn()
        b.record()
        torch.cuda.synchronize()
    ms = a.elapsed_time(b) / REPS
    sm = ps.summary() or {}
    w, mhz = sm.get("watts"), sm.get("sclk_mhz")
    print(N, hop, round(ms, 4), round(b_alg(N, hop) * F / ms / 1e6 / 8000, 4),
          f"{w:.0f} W {mhz:.0f} MHz {w * ms * 1e-3 / F * 1e6:.3f} uJ/frame {ms * 1e-3 * mhz:.1f} Mcycles" if w else "", flush=True)
    del mags, pitch
    torch.cuda.empty_cache()
