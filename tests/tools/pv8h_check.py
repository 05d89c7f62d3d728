#!/usr/bin/env python3
"""tests/tools/pv8h_check.py [hours] [world] — the phase vocoder over BASELINE configs[3]'s signal (8 h of 48 kHz audio, 6.4 M
analysis frames at +3 st) on ONE GPU: the bounded arena (<= 2.5 GB whatever the length), properties of the output, and the
multi-GPU path on the same signal — `world` ranks played by contexts on this device, each walking a range many chunks long
(stage 1 keeps the maps only, stage 2 analyses again with the carry, the rank's edges wait for the seams) — equal to the
single call bit for bit.  Run by tests/test_gpu_fullsize.py in a process of its own (it holds ~25 GB of device memory)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import melonix_amd as mx  # noqa: E402
from bench import SR, gen_shard  # noqa: E402
from melonix_amd import shard as sh  # noqa: E402

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
st = 3.0
dev = torch.device("cuda", 0)
n = int(hours * 3600 * SR)
free0, _ = torch.cuda.mem_get_info()
audio_t = gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
ctx = mx.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
audio = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
f32 = torch.empty(n, dtype=torch.float32, device=dev)
i16 = torch.empty(n, dtype=torch.int16, device=dev)
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info()
ts = []
for _ in range(2):
    t0 = time.perf_counter()
    ctx.pv_pitch_shift_dev(audio, st, f32.data_ptr(), i16.data_ptr())
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
arena = ctx.pv_arena_bytes()
free2, _ = torch.cuda.mem_get_info()
frames = int(np.ceil(n * 2.0 ** (st / 12.0) / 256)) + 1
print(f"pv {hours:g} h {st:+g} st: {frames} frames, call ms {ts[0]:.1f} (first: arena built) / {ts[1]:.1f}; arena {arena / 1e9:.3f} GB; "
      f"device memory taken by the call {(free1 - free2) / 1e9:.3f} GB", flush=True)
assert 0 < arena <= 2.5e9, arena
assert free1 - free2 <= 2.6e9, (free1, free2)  # nothing else was allocated behind the caller's back
# properties: int16 = the f32 clamped and scaled; the level of a sweep is kept; deterministic
ref16 = (f32.clamp(-1.0, 1.0).to(torch.float64) * 32767.0).to(torch.int16)
assert torch.equal(ref16, i16)
del ref16
seg = slice(n // 4, n // 4 + (1 << 24))
rms_in = float(audio_t[mx.MX_AUDIO_PAD:][seg].to(torch.float64).pow(2).mean().sqrt())
rms_out = float(f32[seg].to(torch.float64).pow(2).mean().sqrt())
assert abs(rms_out / rms_in - 1.0) < 0.01, (rms_in, rms_out)
again = torch.empty_like(f32)
ctx.pv_pitch_shift_dev(audio, st, again.data_ptr(), None)
torch.cuda.synchronize()
assert torch.equal(again.view(torch.int32), f32.view(torch.int32))
del again
# the output's pitch track, by this build's own STFT + pitch pick: the input's sweep times 2^(3/12)
N, HOP = 4096, 256
out_a = torch.zeros(n + 2 * mx.MX_AUDIO_PAD, dtype=torch.float32, device=dev)
out_a[mx.MX_AUDIO_PAD:mx.MX_AUDIO_PAD + n] = f32
oa = ctx.wrap_device(out_a.data_ptr(), n, keepalive=out_a)
F = mx.frame_count(n, HOP)
pitch = torch.empty((F, 2), dtype=torch.int32, device=dev)
band = (5, 200)
ctx.stft_hop_dev(oa, N, HOP, 0, F, None, pitch.data_ptr(), band=band)
torch.cuda.synchronize()
bins = pitch[:, 0].cpu().numpy().astype(np.float64)
h = np.arange(F)
f_in = 110.0 + (1760.0 - 110.0) * ((h + 1) * HOP / SR) / (n / SR)
expect = f_in * 2.0 ** (st / 12.0) * N / SR
sel = (expect > band[0] + 3) & (expect < band[1] - 3) & (h > 64) & (h < F - 64)
assert sel.sum() > 400000 * hours / 8
assert np.abs(bins[sel] - expect[sel]).max() <= 2.0, float(np.abs(bins[sel] - expect[sel]).max())
del out_a, pitch
# the same signal as `world` ranks (contexts on this device; the two exchanges by hand)
ctxs = [mx.Context(0) for _ in range(world)]
auds = [c.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t) for c in ctxs]
t0 = time.perf_counter()
tots = [c.pv_shard_analyze(x, st, r, world) for r, (c, x) in enumerate(zip(ctxs, auds))]
all_sums = np.stack([t[0] for t in tots])
all_org = np.stack([t[1] for t in tots])
seams = [c.pv_shard_synthesize(sh.pv_fold_carry(all_sums, all_org, r) if r else None) for r, c in enumerate(ctxs)]
ok = True
for r, c in enumerate(ctxs):
    _, _, lo, hi = mx.pv_shard_frames(n, st, r, world)
    pf, pi = c.pv_shard_finish(hi - lo, seams[r - 1][1] if r else None, seams[r + 1][0] if r < world - 1 else None)
    assert c.pv_arena_bytes() <= 2.5e9
    ok &= bool(torch.equal(torch.from_numpy(pf.view(np.int32)).to(dev), f32[lo:hi].view(torch.int32)))
    ok &= bool(torch.equal(torch.from_numpy(pi).to(dev), i16[lo:hi]))
    del pf, pi
    c.close()
print(f"{world} ranks on one device: {time.perf_counter() - t0:.1f} s incl. host copies; slices {'equal' if ok else 'DIFFER from'} the single call", flush=True)
assert ok
print("pv8h_check ok")
