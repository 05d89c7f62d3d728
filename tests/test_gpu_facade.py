"""The C++ drop-in facade (melonix_amd/cpp: Spec, SpecCache, saveWav, melonix::Resynth) driven the
way the reference's App drives the originals, checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import SR, accum_sweep, mag_tol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(mxlib, tmp_path_factory):
    cpp = os.path.join(ROOT, "melonix_amd", "cpp")
    subprocess.check_call(["make", "-s", "-C", cpp, "NO_GL=1"])
    exe = str(tmp_path_factory.mktemp("facade") / "facade_driver")
    lib = os.path.join(ROOT, "melonix_amd", "lib")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-DMELONIX_AMD_NO_GL", "-I", cpp, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "facade_driver.cpp"), "-o", exe, "-L", lib,
                           "-lmelonix_facade", "-lmelonix_amd", f"-Wl,-rpath,{lib}", "-lpthread"])
    return exe


@pytest.mark.parametrize("N", [32768, 4096])
def test_facade_end_to_end(driver, oracle, tmp_path, N):
    w = accum_sweep(10 * SR)
    n = len(w)
    w.tofile(tmp_path / "audio.f32")
    r = subprocess.run([driver, str(tmp_path / "audio.f32"), str(tmp_path), str(N)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout)  # (pytest -s shows the cold-screen timing)
    assert "cold_screen:" in r.stdout
    bins = N // 2
    # Spec::getSpec rows vs the oracle (N = 32768 is the reference's SpectrSize)
    keys = [(47000, 47375), (0, 256), (-500, -100), (100000, 100001)]
    rows = np.fromfile(tmp_path / "rows.f32", np.float32).reshape(len(keys), bins)
    ref = np.stack([oracle.spec_frame(w, N, s, e) for s, e in keys])
    assert (np.abs(rows - ref) <= mag_tol(ref)).all()
    # SpecCache: column indexing is bit-exact, the texture is the oracle colormap of the row it got
    tex = np.fromfile(tmp_path / "tex.u8", np.uint8).reshape(4, bins, 3)
    texrows = np.fromfile(tmp_path / "texrows.f32", np.float32).reshape(4, bins)
    tm = oracle.TimeMap([], SR, n)
    for i, t in enumerate((1.0, 2.5, 0.0, 7.123)):
        _, s, e = tm.column_range(t, 1280, 10.0)
        oref = oracle.spec_frame(w, N, s, e)
        assert (np.abs(texrows[i] - oref) <= mag_tol(oref[None])[0]).all()
        assert np.array_equal(tex[i], oracle.colormap(texrows[i], 512.0 * 64))
        # against the oracle's own magnitudes the texture may differ by one level where a bin sits on an edge
        d = np.abs(tex[i].astype(int) - oracle.colormap(oref, 512.0 * 64).astype(int))
        assert (d <= 1).mean() > 0.999
    # saveWav: byte-exact incl. the save-wav.cpp:43 quirk
    pcm = (np.arange(1000) * 37 - 12000).astype(np.int16)
    assert (tmp_path / "plain.wav").read_bytes() == oracle.wav_bytes(pcm, SR)
    # Resynth::exportWav == App::exportWav (oracle), byte for byte; PCM bit-exact
    mk = [(1, 0, 0, 3.0), (n - 1, 0, 0, 3.0)]
    _, opcm = oracle.export_run(w, SR, mk)
    got = np.fromfile(tmp_path / "pcm.f32", np.float32)
    assert np.array_equal(got.view(np.uint32), opcm.view(np.uint32))
    assert (tmp_path / "export.wav").read_bytes() == oracle.wav_bytes(oracle.pcm_to_i16(opcm), SR)
    assert np.array_equal(np.fromfile(tmp_path / "grains.i32", np.int32), oracle.grains(w)[0])
    # Resynth::phaseVocoder == the build's own oracle (the reference has no phase vocoder)
    from oracle import pv_oracle
    pvgot = np.fromfile(tmp_path / "pv.f32", np.float32)
    assert np.abs(pvgot - pv_oracle.pitch_shift(w.astype(np.float64), 3.0)).max() <= 2e-5
    ramp = [(1, 0, 0, -5.0), (n // 2, 0, 0.5, 7.0), (n - 1, 0, 0, 0.0)]
    pvm = np.fromfile(tmp_path / "pv_markers.f32", np.float32)
    assert np.abs(pvm - pv_oracle.render(w.astype(np.float64), SR, ramp)).max() <= 5e-5
    want16 = (np.clip(pvm, -1.0, 1.0).astype(np.float64) * 32767.0).astype(np.int16)
    assert (tmp_path / "export_pv.wav").read_bytes() == oracle.wav_bytes(want16, SR)  # through saveWav, quirk and all
    # Resynth::refill == App::playback's refill loop at t = 2.5 s (oracle), bit for bit, and its exit cursor
    _, rest, cend = oracle.playback_fill(w, SR, mk, 2.5, 1024 + 1500)
    got = np.fromfile(tmp_path / "refill.f32", np.float32)
    assert np.array_equal(got.view(np.uint32), rest.view(np.uint32))
    assert np.fromfile(tmp_path / "refill_cursor.f64", np.float64)[0] == cend


def test_facade_without_device_row_cache(driver, oracle, tmp_path):
    """MELONIX_SPEC_DEVICE_MB=0: no kept device rows, every batch through the staging-only calls — the path `Spec` falls
    back to when the row cache cannot allocate.  Same rows and texels; a late getSpec and a changed
    brightness cost transforms instead of copies / re-colourings (the driver checks the counts for this mode)."""
    import os
    N = 32768
    w = accum_sweep(10 * SR)
    w.tofile(tmp_path / "audio.f32")
    env = dict(os.environ, MELONIX_SPEC_DEVICE_MB="0")
    r = subprocess.run([driver, str(tmp_path / "audio.f32"), str(tmp_path), str(N)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cold_screen:" in r.stdout and "recolour_screen:" in r.stdout
    bins = N // 2
    keys = [(47000, 47375), (0, 256), (-500, -100), (100000, 100001)]
    rows = np.fromfile(tmp_path / "rows.f32", np.float32).reshape(len(keys), bins)
    ref = np.stack([oracle.spec_frame(w, N, s, e) for s, e in keys])
    assert (np.abs(rows - ref) <= mag_tol(ref)).all()
    tex = np.fromfile(tmp_path / "tex.u8", np.uint8).reshape(4, bins, 3)
    texrows = np.fromfile(tmp_path / "texrows.f32", np.float32).reshape(4, bins)
    for i in range(4):
        assert np.array_equal(tex[i], oracle.colormap(texrows[i], 512.0 * 64))
