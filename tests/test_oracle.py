"""Pins the oracle (oracle/melonix_oracle.c) before anything is checked against it.

 * grains / process / export / time maps: the known-answer facts the survey recorded from the
   compiled reference.
 * saveWav: byte-compared with the reference's own save-wav.cpp (oracle/_ref, built from
   /root/reference in the build container; travels prebuilt to the GPU box).
 * the double DFT that stands where FFTW stands: against an independent f64 FFT (numpy
   pocketfft) and analytic DFT pairs — the reference holds no golden vectors for it.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import SR, accum_sweep, noisy

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ---- known answers recorded from the compiled reference ----------------
def test_known_answer_grains(oracle, sweep10):
    s, l = oracle.grains(sweep10)
    assert len(s) == 319
    assert s[0] == 0 and (s[1:] == s[:-1] + l[:-1]).all()
    assert l.min() >= 751 and l.max() <= 2249  # "good" grains only on this signal


@pytest.mark.parametrize("pb,calls,samples", [(0, 320, 480407), (3, 379, 478903), (-4, 254, 479781), (7.5, 491, 479189)])
def test_known_answer_export(oracle, sweep10, pb, calls, samples):
    n = len(sweep10)
    steps, pcm = oracle.export_run(sweep10, SR, [(1, 0, 0, pb), (n - 1, 0, 0, pb)])
    assert len(steps) + 1 == calls  # process() calls incl. the terminating one
    assert len(pcm) == samples
    assert int(steps["sz"].sum()) + 1500 == samples
    assert not pcm[-1500:].any()


def test_known_answer_wav(oracle, sweep10, tmp_path):
    n = len(sweep10)
    _, pcm = oracle.export_run(sweep10, SR, [(1, 0, 0, 3), (n - 1, 0, 0, 3)])
    b = oracle.wav_bytes(oracle.pcm_to_i16(pcm), SR)
    assert len(b) == 957850
    assert int.from_bytes(b[40:44], "little") == 957822  # 2M + 16: the save-wav.cpp:43 quirk
    assert b[44:48] == b"\0\0\0\0"                         # samples 0 and 1 zeroed
    assert int.from_bytes(b[4:8], "little") == 957850 - 8
    p = tmp_path / "o.wav"
    from oracle import pyoracle
    assert pyoracle.lib().mxo_save_wav(str(p).encode(), oracle.pcm_to_i16(pcm).ctypes.data_as(
        __import__("ctypes").POINTER(__import__("ctypes").c_int16)), len(pcm), SR) == 0
    assert p.read_bytes() == b


def test_known_answer_column_range(oracle):
    tm = oracle.TimeMap([], SR, 480000)
    assert tm.column_range(1.0, 1280, 10.0) == (128, 48000, 48375)  # SURVEY §8 a-5 (verified there)


def test_memo_equals_pure_on_export(oracle, sweep10):
    n = len(sweep10)
    mk = [(1000, 0, 0.0, 2.0), (100000, 0, 0.5, -3.0), (300000, 0, -0.2, 5.0), (n - 1, 0, 0, 0)]
    a_steps, a_pcm = oracle.export_run(sweep10, SR, mk, memo=True)
    b_steps, b_pcm = oracle.export_run(sweep10, SR, mk, memo=False)
    for f in ("cursor", "grain_start", "grain_len", "rate", "next_first", "sz", "out_offset"):  # (_pad is struct padding)
        assert np.array_equal(a_steps[f], b_steps[f])
    assert np.array_equal(a_pcm.view(np.uint32), b_pcm.view(np.uint32))


# ---- saveWav against the reference's own code -----------------------------------------------
@pytest.mark.parametrize("m", [0, 1, 2, 3, 1000, 70001])
def test_savewav_matches_reference_build(oracle, tmp_path, m):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference here and no prebuilt .so)")
    rng = np.random.default_rng(m)
    pcm = rng.integers(-32768, 32767, m, dtype=np.int16)
    for sr in (48000, 44100, 8000):
        ref = oracle.ref_wav_bytes(pcm, sr, tmp_path / "r.wav")
        assert oracle.wav_bytes(pcm, sr) == ref


# ---- the DFT that stands where FFTW stands ---------------------------------------------------
@pytest.mark.parametrize("N", [2, 4, 8, 64, 4096, 16384, 32768])
def test_fft_vs_independent_f64(oracle, N):
    rng = np.random.default_rng(N)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    got = oracle.fft(x)
    ref = np.fft.fft(x)
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.sqrt(N) * np.max(np.abs(ref))


@pytest.mark.parametrize("N", [4096, 16384, 32768])
def test_fft_error_against_extended_precision(oracle, N):
    """The restatement's own rounding error, measured against an 80-bit (x87 long double) transform of the same input
    (scipy's pocketfft, a third implementation): it stays at the binary64 level any FFTW build works at, five orders
    of magnitude below the binary32 magnitudes spec.cpp:63 rounds to — which FFTW build the reference links against
    cannot move a result bit except at a rounding tie."""
    sfft = pytest.importorskip("scipy.fft")
    if np.finfo(np.longdouble).eps >= np.finfo(np.float64).eps:
        pytest.skip("long double is binary64 on this platform")
    rng = np.random.default_rng(7 * N)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    ref = sfft.fft(x.astype(np.clongdouble))
    err_oracle = float(np.max(np.abs(oracle.fft(x) - ref)) / np.max(np.abs(ref)))
    err_numpy = float(np.max(np.abs(np.fft.fft(x) - ref)) / np.max(np.abs(ref)))
    assert err_oracle <= 4e-15 and err_oracle <= 8 * err_numpy
    assert err_oracle < 1e-7 * float(np.finfo(np.float32).eps)  # far below one binary32 ulp of any bin


def test_fft_analytic_pairs(oracle):
    N = 4096
    d = np.zeros(N, complex); d[3] = 1.0
    k = np.arange(N)
    assert np.allclose(oracle.fft(d), np.exp(-2j * np.pi * 3 * k / N), atol=1e-13)
    tone = np.exp(2j * np.pi * 17 * k / N)
    X = oracle.fft(tone)
    assert abs(X[17] - N) < 1e-9 and np.max(np.abs(np.delete(X, 17))) < 1e-8


@pytest.mark.parametrize("N", [4096, 32768])
def test_spec_frame_vs_numpy_restatement(oracle, N):
    """spec.cpp:44-66 restated independently in numpy (f64 pocketfft)."""
    w = noisy(accum_sweep(10 * SR))
    n = len(w)
    for (s, e) in [(48000, 48375), (0, 256), (-500, -100), (479900, 480300), (5000, 4000)]:
        i = np.arange(e - N, e)
        inside = (i >= 0) & (i < n)
        x = np.zeros(N, np.float32)
        ii = i[inside]
        d = (s - ii).astype(np.float32)
        wt = np.where(ii >= s, np.float32(1), np.exp(np.float32(-2.5e-4) * d).astype(np.float32))
        x[inside] = wt * w[ii]
        ref = (np.abs(np.fft.fft(x.astype(np.float64)))[: N // 2] / N).astype(np.float32)
        got = oracle.spec_frame(w, N, s, e)
        # numpy's expf may differ from glibc's by an ulp: compare to 1e-6 of the frame peak
        assert np.max(np.abs(got - ref)) <= 1e-6 * max(ref.max(), 1e-30) + 1e-12


def test_pitch_pick_and_band(oracle):
    assert oracle.pitch_band(4096, SR) == (5, 150)
    m = np.zeros(2048, np.float32); m[[7, 9, 300]] = [2.0, 2.0, 5.0]
    assert oracle.pitch_pick(m, 5, 150) == (7, 2.0)        # ties -> lowest k; 300 is out of band
    assert oracle.pitch_pick(np.zeros(2048, np.float32), 5, 150) == (5, 0.0)


def test_colormap_segments(oracle):
    """spec-cache.cpp:77-96: integer thresholds 85 / 170, truncating casts."""
    k = 1.0
    m = np.array([0, 10.9, 84.99, 85, 100, 169.9, 170, 200, 255, 300, -5], np.float32)
    rgb = oracle.colormap(m, k)
    assert rgb[0].tolist() == [0, 0, 0] and rgb[1].tolist() == [10, 0, 0] and rgb[2].tolist() == [84, 0, 0]
    assert rgb[3].tolist() == [85, 0, 0]                      # a = 0 -> (tmp*cos0, tmp*sin0)
    a = (np.float32(100) - 85) / 85 * 3.141592 / 2
    assert rgb[4].tolist() == [int(100 * np.cos(a)), int(100 * np.sin(a)), 0]
    assert rgb[6].tolist() == [0, 170, 0] and rgb[7].tolist() == [90, 200, 90]
    assert rgb[8].tolist() == [255, 255, 255] and rgb[9].tolist() == [255, 255, 255] and rgb[10].tolist() == [0, 0, 0]


# ---- committed fixtures (tests/golden/, generated by tests/golden/make_golden.py) ---------------
def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_golden_fixture(oracle, sweep10):
    path = os.path.join(GOLD, "c1_sweep10.json")
    with open(path) as f:
        g = json.load(f)
    assert _sha(sweep10) == g["signal_sha256"], "the known-answer signal itself changed"
    s, l = oracle.grains(sweep10)
    assert _sha(s) == g["grain_starts_sha256"] and _sha(l) == g["grain_lens_sha256"]
    n = len(sweep10)
    for key, pb in (("p0", 0.0), ("p3", 3.0), ("m4", -4.0), ("p7_5", 7.5)):
        steps, pcm = oracle.export_run(sweep10, SR, [(1, 0, 0, pb), (n - 1, 0, 0, pb)])
        e = g["export"][key]
        assert len(pcm) == e["samples"] and len(steps) + 1 == e["process_calls"]
        assert _sha(steps["sz"]) == e["sz_sha256"] and _sha(pcm) == e["pcm_f32_sha256"]
        assert _sha(oracle.pcm_to_i16(pcm)) == e["pcm_i16_sha256"]
        assert pcm[:8].tolist() == e["pcm_head"] and steps["rate"][0].item() == e["rate0"]
    rows = np.load(os.path.join(GOLD, "c1_mag_rows.npz"))
    for N in (4096, 32768):
        rr = rows[f"ranges_{N}"]
        got = np.stack([oracle.spec_frame(sweep10, N, int(s_), int(e_)) for s_, e_ in rr])
        assert np.array_equal(got, rows[f"mags_{N}"])


@pytest.mark.parametrize("N", [4096, 16384, 32768])
def test_dft_matches_fftw_api_library(oracle, N):
    """spec.cpp's call sequence — fftw_plan_dft_1d(N, in, out, FFTW_FORWARD, FFTW_MEASURE), fftw_execute — run on a
    production implementation of the FFTW3 API found on this machine (a real libfftw3, else Intel MKL's FFTW3
    interface) gives the magnitudes of the built-in restatement to the last or next-to-last binary32 digit.  (The
    reference pins no FFTW version and ships no vectors, so this still is not a pin by the reference itself.)"""
    if oracle.fftw_api_name() == "none":
        pytest.skip("no library implementing the FFTW3 API on this machine")
    from conftest import accum_sweep, noisy
    w = noisy(accum_sweep(10 * 48000))
    n = len(w)
    for s, e in [(48000, 48375), (0, 256), (-500, -100), (239744, 240000), (479900, 480300), (100000, 100001),
                 (1000, 60000), (n - 375, n)]:
        a = oracle.spec_frame(w, N, s, e)
        b = oracle.spec_frame_fftw_api(w, N, s, e)
        assert b is not None
        assert np.abs(a - b).max() <= 2.0 * np.spacing(np.float32(max(a.max(), 1e-30)))
    m1, b1, p1 = oracle.stft_hop(w[:200000], N, 512, nthreads=4)
    m2, b2, p2 = oracle.stft_hop(w[:200000], N, 512, nthreads=4, fftw_api=True)
    assert np.abs(m1 - m2).max() <= 2.0 * np.spacing(np.float32(m1.max())) and (b1 == b2).mean() > 0.999
