"""Runs the gfx950 kernel's per-thread templates (melonix_amd/csrc/stft_core.h) thread by thread
on the CPU (tests/emu/stft_emu.cpp) and checks them against the oracle: index maps, LDS swizzle
closed forms (bijective, equal to t1_index), twiddles, the real-FFT split, the sliding window."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import SR, accum_sweep, mag_tol, noisy

HERE = os.path.dirname(os.path.abspath(__file__))


RANGES = [(48000, 48375), (0, 256), (-500, -100), (239744, 240000), (479900, 480300), (100000, 100001),
          (5000, 4000), (1000, 60000)]


@pytest.mark.parametrize("N,E", [(4096, 16), (16384, 32), (32768, 32)])
def test_single_frames(emu, oracle, N, E):
    w = noisy(accum_sweep(10 * SR))
    fp = C.POINTER(C.c_float)
    for (s, e) in RANGES:
        ref = oracle.spec_frame(w, N, s, e)
        for hop_mode in (0, 1):
            if hop_mode and e <= s:
                continue
            out = np.empty(N // 2, np.float32)
            rc = emu.emu_stft_frame(N, E, w.ctypes.data_as(fp), len(w), s, e, hop_mode, out.ctypes.data_as(fp))
            assert rc == 0, "LDS closed-form address differs from the swizzle, or is not a bijection"
            assert (np.abs(out - ref) <= mag_tol(ref[None])[0]).all()
            assert np.abs(out - ref).max() <= 1e-6 * max(ref.max(), 1e-30) + 1e-12  # in fact ~1e-7 of the peak


@pytest.mark.parametrize("N,E,hop,first,count", [(4096, 16, 256, 520, 43), (4096, 16, 256, 0, 40),
                                                  (16384, 32, 512, 0, 40), (16384, 32, 512, 250, 32),
                                                  (4096, 16, 512, 0, 24), (4096, 16, 512, 100, 20), (16384, 32, 1024, 3, 20),
                                                  (32768, 32, 1024, 0, 20), (32768, 32, 1024, 40, 18)])
def test_sliding_window(emu, oracle, N, E, hop, first, count):
    """The register image slid frame to frame stays within 1e-6 of the frame peak of the oracle
    (tolerance 2e-5): at most N/hop - 2 decays per point."""
    w = noisy(accum_sweep(3 * SR))
    fp = C.POINTER(C.c_float)
    out = np.empty((count, N // 2), np.float32)
    rc = emu.emu_stft_slide(N, E, hop, w.ctypes.data_as(fp), len(w), first, count, out.ctypes.data_as(fp))
    assert rc == 0
    ref = np.stack([oracle.spec_frame(w, N, (first + f) * hop, (first + f + 1) * hop) for f in range(count)])
    err = np.abs(out - ref)
    assert (err <= mag_tol(ref)).all()
    assert (err.max(axis=1) <= 1e-6 * ref.max(axis=1) + 1e-12).all()


@pytest.mark.parametrize("N,E,hop,first,count", [(4096, 16, 375, 0, 40), (4096, 16, 375, 360, 25), (4096, 16, 100, 3, 50),
                                                  (4096, 16, 384, 0, 24), (4096, 16, 257, 11, 24), (4096, 16, 1, 4000, 24),
                                                  (16384, 32, 375, 0, 50), (16384, 32, 512 - 1, 250, 20),
                                                  (32768, 32, 375, 0, 24), (32768, 32, 512, 270, 12), (32768, 32, 500, 5, 10)])
def test_circular_window(emu, oracle, N, E, hop, first, count):
    """Uniform hops that do not slide by whole slots: the circular register image (only the newest 2*hop samples
    fetched per frame, everything older aged by one multiply) stays within 1e-6 of the frame peak of the oracle —
    including frames that start before the file and end after it."""
    w = noisy(accum_sweep(3 * SR))
    fp = C.POINTER(C.c_float)
    out = np.empty((count, N // 2), np.float32)
    rc = emu.emu_stft_circ(N, E, hop, w.ctypes.data_as(fp), len(w), first, count, out.ctypes.data_as(fp))
    assert rc == 0
    ref = np.stack([oracle.spec_frame(w, N, (first + f) * hop, (first + f + 1) * hop) for f in range(count)])
    err = np.abs(out - ref)
    assert (err <= mag_tol(ref)).all()
    assert (err.max(axis=1) <= 1e-6 * ref.max(axis=1) + 1e-12).all()
