#!/usr/bin/env python3
"""tools/power_wall.py — is the headline STFT kernel limited by the package power budget?  (not part of the product)

The same launch (60 min, N = 4096, hop 256, magnitudes + pitch) runs back to back for ~1.5 s per case while a host thread
samples the package power and the shader clock (bench.PowerSampler: amdgpu hwmon of the device under test):

  inputs   silence | the workload sweep | sweep + 1e-3 * PCG32 noise | full-scale PCG32 noise
           (same instruction stream, same memory traffic: only the number of wires that toggle differs)
  outputs  magnitudes + pitch | pitch only (no 8 KiB row per frame: the store energy is gone, the arithmetic stays)
  clocks   the default, then the shader clock capped at a few levels (`rocm-smi --setperfdeterminism <MHz>`) if the
           lease lets this process set them — a kernel at the power wall loses LESS than the clock ratio when capped
           (the cap removes the voltage the power manager would have had to give back anyway), an issue-bound kernel
           loses exactly the ratio

usage: power_wall.py [seconds per case] [N hop]"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench as B  # noqa: E402
import melonix_amd as mx  # noqa: E402

SR = 48000


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    hop = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n = 60 * 60 * SR
    pad = mx.MX_AUDIO_PAD
    F = mx.frame_count(n, hop)
    ctx = mx.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    band = mx.pitch_band(N, SR)
    mags = torch.empty((F, N // 2), dtype=torch.float32, device=dev)
    pitch = torch.empty((F, 2), dtype=torch.int32, device=dev)

    def inputs():
        z = torch.zeros(n + 2 * pad, dtype=torch.float32, device=dev)
        yield "silence", z
        s = B.gen_shard(torch, dev, 0, 1, n, pad)
        yield "sweep (the workload)", s
        s2 = s.clone()
        B.add_noise(torch, dev, s2, 0, 1, n, pad, level=1e-3)
        yield "sweep + 1e-3 PCG32 noise", s2
        del s2
        u = torch.zeros(n + 2 * pad, dtype=torch.float32, device=dev)
        u[pad:pad + n] = (0.5 * B.pcg32_uniform(torch, dev, 0, n)).to(torch.float32)
        yield "PCG32 noise, amplitude 0.5", u

    def run_case(audio_t, with_rows):
        audio = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
        launch = lambda: ctx.stft_hop_dev(audio, N, hop, 0, F, mags.data_ptr() if with_rows else None, pitch.data_ptr(), band=band)
        for _ in range(30):
            launch()
        torch.cuda.synchronize()
        with B.PowerSampler(0) as ps:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.time()
            k = 0
            a.record()
            while time.time() - t0 < secs:
                for _ in range(50):
                    launch()
                torch.cuda.synchronize()
                k += 50
            b.record()
            torch.cuda.synchronize()
        ms = a.elapsed_time(b) / k
        sm = ps.summary() or {}
        audio.free()
        return ms, sm

    def table(tag):
        rows = []
        for name, t in inputs():
            for with_rows in (True, False):
                ms, sm = run_case(t, with_rows)
                w, mhz = sm.get("watts"), sm.get("sclk_mhz")
                uj = (w * ms * 1e-3 / F * 1e6) if w else None
                rows.append({"clock": tag, "input": name, "outputs": "magnitudes + pitch" if with_rows else "pitch only",
                             "ms_per_launch": round(ms, 4), "watts": w, "sclk_mhz": mhz,
                             "uJ_per_frame": round(uj, 3) if uj else None,
                             "frac_hbm": round(B.b_alg(N, hop, with_rows) * F / (ms * 1e-3) / 1e9 / B.HBM_PEAK_GBS, 4),
                             "shader_Mcycles_per_launch": round(ms * 1e-3 * mhz, 2) if mhz else None, "samples": sm.get("samples")})
                print(json.dumps(rows[-1]), flush=True)
            del t
            torch.cuda.empty_cache()
        return rows

    print(f"# power_wall: N={N} hop={hop}, {F} frames per launch, {secs} s per case; sampler: {B.PowerSampler(0).source}", flush=True)
    table("default")
    # shader-clock caps, if this process may set them
    for mhz in (2100, 1800, 1500, 1200):
        r = subprocess.run(["rocm-smi", "--setperfdeterminism", str(mhz)], capture_output=True, text=True)
        ok = r.returncode == 0 and "rror" not in (r.stdout + r.stderr) and "denied" not in (r.stdout + r.stderr).lower()
        print(f"# rocm-smi --setperfdeterminism {mhz}: rc={r.returncode} {'ok' if ok else 'REFUSED'} "
              f"{(r.stdout + r.stderr).strip().splitlines()[-1][:160] if (r.stdout + r.stderr).strip() else ''}", flush=True)
        if not ok:
            break
        try:
            table(f"sclk <= {mhz} MHz")
        finally:
            subprocess.run(["rocm-smi", "--resetperfdeterminism"], capture_output=True, text=True)
    ctx.close()


main()
