#!/usr/bin/env python3
"""Regenerates tests/golden/*: data fixtures for the 10 s known-answer signal.

The reference has no tests or golden vectors, and spec.cpp/app.cpp cannot be built here without
stand-in headers, so these fixtures are produced by the ORACLE after it has been
pinned against the facts the survey recorded from the compiled reference (BASELINE.md §2:
319 grains; 320/379/254/491 process() calls; 480407/478903/479781/479189 samples; WAV size and
header quirk).  They guard the oracle (and through it every parity test) against regressions;
they are data only: hashes, a few PCM samples and eight magnitude rows.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from conftest import SR, accum_sweep  # noqa: E402
from oracle import pyoracle as O  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    w = accum_sweep(10 * SR)
    n = len(w)
    s, l = O.grains(w)
    assert len(s) == 319
    g = {"signal": "phase-accumulated sweep 110->1760 Hz, 10 s, 48 kHz, amp 0.5 (tests/conftest.py accum_sweep)",
         "signal_sha256": sha(w), "grains": int(len(s)), "grain_starts_sha256": sha(s), "grain_lens_sha256": sha(l),
         "export": {}}
    facts = {"p0": (0.0, 320, 480407), "p3": (3.0, 379, 478903), "m4": (-4.0, 254, 479781), "p7_5": (7.5, 491, 479189)}
    for key, (pb, calls, samples) in facts.items():
        steps, pcm = O.export_run(w, SR, [(1, 0, 0, pb), (n - 1, 0, 0, pb)])
        assert len(steps) + 1 == calls and len(pcm) == samples, "oracle no longer reproduces BASELINE.md §2"
        g["export"][key] = {"pitch_bend": pb, "process_calls": calls, "samples": samples,
                            "sz_sha256": sha(steps["sz"]), "pcm_f32_sha256": sha(pcm),
                            "pcm_i16_sha256": sha(O.pcm_to_i16(pcm)), "pcm_head": pcm[:8].tolist(),
                            "rate0": float(steps["rate"][0])}
    with open(os.path.join(HERE, "c1_sweep10.json"), "w") as f:
        json.dump(g, f, indent=1)
    ranges = {}
    for N in (4096, 32768):
        rr = np.array([(48000, 48375), (0, 256), (-500, -100), (239744, 240000), (479900, 480300), (100000, 100001),
                       (5000, 4000), (1000, 60000)], np.int32)
        ranges[f"ranges_{N}"] = rr
        m = np.stack([O.spec_frame(w, N, int(a), int(b)) for a, b in rr])
        if N == 32768:  # keep the fixture small: first 2048 bins would lose coverage; store float16-free full rows
            pass
        ranges[f"mags_{N}"] = m
    np.savez_compressed(os.path.join(HERE, "c1_mag_rows.npz"), **ranges)
    # build-defined phase vocoder (no reference counterpart): pins oracle/pv_oracle.py — the definition — against
    # accidental change; 2 s of the same kind of sweep, +3 / -4 semitones, every 997th sample + level
    from oracle import pv_oracle as pv
    w2 = accum_sweep(2 * SR).astype(np.float64)
    pvg = {"signal": "accum_sweep(2 s)", "stride": 997}
    for st in (3.0, -4.0):
        y = pv.pitch_shift(w2, st)
        pvg[f"st_{st:+.0f}"] = {"samples": y[::997].tolist(), "rms": float(np.sqrt((y ** 2).mean())), "len": len(y)}
    with open(os.path.join(HERE, "pv_sweep2.json"), "w") as f:
        json.dump(pvg, f)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
