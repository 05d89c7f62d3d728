#!/usr/bin/env python3
"""BASELINE configs[3] on ONE device (driven by tests/test_gpu_fullsize.py; TEST INFRASTRUCTURE).

8 h of the closed-form sweep (1 382 400 000 samples, 5 400 000 frames at N = 4096 / hop 256) is cut into the 8 frame
shards `melonix_amd.shard.shard_frames` gives 8 ranks.  Each shard is laid out as its own padded device image with the
true neighbour samples in the pads (the N - hop input halo; `padded_shard` semantics) and run through mx_stft_hop_dev.
Checks: every magnitude row of every shard — the 7 seams' first frames included — and the concatenated pitch track are
bit-identical to the unsharded 8 h run; rows of the unsharded run (all seam rows among them) agree with the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402  (first: torch must bring up the HIP runtime before libmelonix_amd.so does)

torch.cuda.init()
import bench as B  # noqa: E402
import melonix_amd as mx  # noqa: E402
from melonix_amd import shard as sh  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

SR = 48000


def main(hours: float = 8.0, N: int = 4096, HOP: int = 256, world: int = 8):
    dev = torch.device("cuda", 0)
    n = int(hours * 60 * 60 * SR)
    pad = mx.MX_AUDIO_PAD
    F = mx.frame_count(n, HOP)
    whole_t = B.gen_shard(torch, dev, 0, 1, n, pad)  # [pad zeros][n samples][pad zeros]
    ctx = mx.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    band = mx.pitch_band(N, SR)
    whole = ctx.wrap_device(whole_t.data_ptr(), n, keepalive=whole_t)
    mags = torch.empty((F, N // 2), dtype=torch.float32, device=dev)  # 44 GB at 8 h
    pitch = torch.empty((F, 2), dtype=torch.int32, device=dev)
    ctx.stft_hop_dev(whole, N, HOP, 0, F, mags.data_ptr(), pitch.data_ptr(), band=band)
    torch.cuda.synchronize()
    parts = [sh.shard_frames(n, N, HOP, r, world) for r in range(world)]
    g = sh.pin_run_length(ctx, N, HOP, F)  # what every rank of a sharded job does: the whole signal's run length
    assert parts[0].lo == 0 and parts[-1].hi == F and all(p.lo % sh.frame_align(N, HOP) == 0 or p.lo == F for p in parts)
    track = []
    smags = torch.empty((max(p.frames for p in parts), N // 2), dtype=torch.float32, device=dev)
    for p in parts:
        own = p.sample_hi - p.sample_lo
        img = torch.zeros(own + 2 * pad, dtype=torch.float32, device=dev)  # the rank's own image
        g0 = p.sample_lo - pad
        lo_s, hi_s = max(g0, 0), min(p.sample_hi + pad, n)
        img[lo_s - g0: hi_s - g0] = whole_t[pad + lo_s: pad + hi_s]       # neighbours in the pads, zeros beyond the signal
        if p.frames == 0:
            continue
        assert p.rank == 0 or p.halo_left == N - HOP
        a = ctx.wrap_device(img.data_ptr(), own, keepalive=img)
        sp = torch.empty((p.frames, 2), dtype=torch.int32, device=dev)
        smags.fill_(-1.0)
        ctx.stft_hop_dev(a, N, HOP, 0, p.frames, smags.data_ptr(), sp.data_ptr(), band=band)
        torch.cuda.synchronize()
        assert torch.equal(smags[: p.frames], mags[p.lo: p.hi]), f"shard {p.rank}: rows differ from the unsharded run"
        track.append(sp)
        a.free()
        del img
    assert torch.equal(torch.cat(track), pitch), "concatenated pitch track differs"
    pick = sorted({0, 1, 17, F - 1, F // 3} | {p.lo for p in parts[1:] if p.frames} | {p.lo - 1 for p in parts[1:] if p.frames})
    got = mags[torch.tensor(pick, device=dev)].cpu().numpy()
    worst = 0.0
    for i, f in enumerate(pick):
        s, e = f * HOP, (f + 1) * HOP
        s0 = max(0, e - N)
        seg = whole_t[pad + s0: pad + min(n, e)].cpu().numpy()
        ref = O.spec_frame(seg, N, s - s0, e - s0)
        tol = 2e-5 * ref.max() + 1e-9
        err = float(np.abs(got[i] - ref).max())
        assert err <= tol, (f, err, tol)
        worst = max(worst, err / tol)
    extra = ""
    if hours >= 8.0 and N == 4096:
        # configs[3] in full against the oracle: all 5.4 M pitch records of the 8 h signal, and the magnitude rows of 64
        # chunks of 1024 frames spread over the 8 hours (the seams are among the rows checked above)
        import time

        t0 = time.time()
        host = whole_t[pad: pad + n].cpu().numpy()  # the very samples the device transformed (5.5 GB)
        T = len(os.sched_getaffinity(0))
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                T = min(T, -(-int(q) // int(per)))
        except Exception:
            pass
        _, ob, om = O.stft_hop(host, N, HOP, band=band, want_mags=False, nthreads=T)
        pt = pitch.cpu().numpy()
        gb, gm = pt[:, 0], pt[:, 1].view(np.float32)
        diff = np.nonzero(gb != ob)[0]
        assert len(diff) <= 500, len(diff)
        for f in diff:
            ref = O.spec_frame(host, N, int(f) * HOP, (int(f) + 1) * HOP)
            assert ref[ob[f]] - ref[gb[f]] <= 2 * (2e-5 * ref.max() + 1e-9), int(f)
        same = gb == ob
        assert (np.abs(gm[same] - om[same]) <= 2e-5 * om[same] + 1e-9).mean() > 0.9999
        rows = 0
        for c in np.linspace(0, F - 1024, 64).astype(np.int64) // 32 * 32:
            rm, _, _ = O.stft_hop(host, N, HOP, first=int(c), count=1024, band=band, nthreads=T)
            gmr = mags[int(c): int(c) + 1024].cpu().numpy()
            tolr = 2e-5 * rm.max(axis=1) + 1e-9
            assert (np.abs(gmr - rm).max(axis=1) <= tolr).all(), int(c)
            rows += 1024
        extra = f"; vs the oracle: {F} of {F} pitch records ({len(diff)} near-ties), {rows} magnitude rows, {time.time() - t0:.0f} s on {T} threads"
        del host
    whole.free()
    ctx.close()
    print(f"shard8_check ok: {F} frames, {world} shards of {parts[0].frames} frames (run length {g}), {len(pick)} rows vs the oracle "
          f"(worst {worst:.3f} of the tolerance){extra}")


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 8.0, *(int(v) for v in sys.argv[2:5]))
