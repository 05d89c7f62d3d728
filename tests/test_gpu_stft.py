"""GPU parity: HIP STFT + pitch pick (through the C-ABI) vs the oracle."""
import numpy as np
import pytest

from conftest import SR, accum_sweep, mag_tol, noisy

pytestmark = pytest.mark.gpu

SIZES = [4096, 16384, 32768]


def _check_pitch(pitch, ref_mags, band, tol_rows):
    kmin, kmax = band
    sub = ref_mags[:, kmin:kmax + 1]
    ref_bin = sub.argmax(axis=1) + kmin  # first maximum = lowest k on ties
    same = pitch["bin"] == ref_bin
    # where they differ, the oracle's value at the GPU's bin must be within tolerance of its maximum
    bad = np.nonzero(~same)[0]
    for f in bad:
        b = pitch["bin"][f]
        assert kmin <= b <= kmax
        assert ref_mags[f, ref_bin[f]] - ref_mags[f, b] <= 2 * tol_rows[f, 0], (f, b, ref_bin[f])
    # and the reported magnitude is the GPU's own value at that bin (checked by caller when mags are there)
    return len(bad)


@pytest.mark.parametrize("N,seed", [(4096, 1), (4096, 2), (16384, 3), (32768, 4)])
def test_random_ranges_vs_oracle(gpu_ctx, oracle, N, seed):
    """Randomised (start, end) pairs as a display would produce them and far beyond: columns of any width from 1
    sample to several frames, starts before the file, ends past it, reversed pairs, odd alignments."""
    rng = np.random.default_rng(seed)
    w = noisy(accum_sweep(5 * SR), level=0.01)
    n = len(w)
    cnt = 160 if N == 4096 else 48
    ends = rng.integers(-2 * N, n + 2 * N, cnt)
    widths = np.concatenate([rng.integers(1, 600, cnt // 2), rng.integers(-50, 4 * N, cnt - cnt // 2)])
    ranges = np.stack([ends - widths, ends], axis=1).astype(np.int32)
    a = gpu_ctx.upload(w)
    band = oracle.pitch_band(N, SR)
    mags, pitch = gpu_ctx.stft_ranges(a, N, ranges, band=band)
    ref = np.stack([oracle.spec_frame(w, N, int(s), int(e)) for s, e in ranges])
    tol = mag_tol(ref)
    err = np.abs(mags - ref)
    assert (err <= tol).all(), float((err / tol).max())
    _check_pitch(pitch, ref, band, tol)
    a.free()


@pytest.mark.parametrize("N", SIZES)
def test_ranges_vs_oracle(gpu_ctx, oracle, N):
    w = noisy(accum_sweep(10 * SR))
    n = len(w)
    ranges = [(48000, 48375), (0, 256), (-500, -100), (239744, 240000), (479900, 480300), (100000, 100001),
              (5000, 4000), (1000, 60000), (n - 1, n), (n, n + 375), (n + N, n + N + 10), (-10, 5), (0, 0),
              (12345, 12346), (n - 375, n)]
    a = gpu_ctx.upload(w)
    band = oracle.pitch_band(N, SR)
    mags, pitch = gpu_ctx.stft_ranges(a, N, ranges, band=band)
    ref = np.stack([oracle.spec_frame(w, N, s, e) for s, e in ranges])
    tol = mag_tol(ref)
    err = np.abs(mags - ref)
    assert (err <= tol).all(), float((err / tol).max())
    _check_pitch(pitch, ref, band, tol)
    assert np.array_equal(pitch["mag"], mags[np.arange(len(ranges)), pitch["bin"]])
    a.free()


SLIDING_HOPS = ((4096, 256), (4096, 512), (16384, 512), (16384, 1024), (32768, 1024))  # stft_kernels.hip Tune::slides


@pytest.mark.parametrize("N,hop", [(4096, 256), (16384, 512), (32768, 375), (4096, 375), (4096, 128), (4096, 512),
                                   (16384, 256), (32768, 512), (4096, 2), (4096, 4096), (4096, 6000),
                                   (16384, 1024), (32768, 1024), (4096, 1024)])
def test_hop_vs_oracle(gpu_ctx, oracle, N, hop):
    w = noisy(accum_sweep(4 * SR))
    n = len(w)
    F = (n + hop - 1) // hop
    a = gpu_ctx.upload(w)
    band = oracle.pitch_band(N, SR)
    mags, pitch = gpu_ctx.stft_hop(a, N, hop, band=band)
    assert mags.shape == (F, N // 2)
    # the oracle is slow at the big sizes: check a deterministic subset of frames incl. both ends
    rng = np.random.default_rng(N + hop)
    pick = np.unique(np.concatenate([np.arange(0, 40), np.arange(F - 40, F), rng.integers(0, F, 120)]))
    pick = pick[(pick >= 0) & (pick < F)]
    ref = np.stack([oracle.spec_frame(w, N, int(h) * hop, (int(h) + 1) * hop) for h in pick])
    tol = mag_tol(ref)
    err = np.abs(mags[pick] - ref)
    assert (err <= tol).all(), float((err / tol).max())
    _check_pitch(pitch[pick], ref, band, tol)
    assert np.array_equal(pitch["mag"], mags[np.arange(F), pitch["bin"]])
    # frame indexing: with one frame per workgroup every frame is loaded directly (no sliding-window
    # carry).  The sliding kernels then read the exact weight table, as ranges mode does, and the rows
    # are bit-identical; the other hops derive the weights from a per-thread seed (<= 2 ulp off the
    # table, load_frame_geo), so their rows agree with ranges mode to a tenth of the tolerance.
    gpu_ctx.set_frames_per_block(1)
    m1, p1 = gpu_ctx.stft_hop(a, N, hop, band=band)
    gpu_ctx.set_frames_per_block(0)
    rr = np.stack([pick * hop, (pick + 1) * hop], axis=1).astype(np.int32)
    m2, p2 = gpu_ctx.stft_ranges(a, N, rr, band=band)
    if (N, hop) in SLIDING_HOPS:
        assert np.array_equal(m2, m1[pick])
        assert np.array_equal(p2, p1[pick])
    else:
        assert (np.abs(m2 - m1[pick]) <= mag_tol(m2) / 10).all()
        _check_pitch(p1[pick], ref, band, tol)
        assert np.array_equal(p1["mag"], m1[np.arange(F), p1["bin"]])
    # the sliding register image (frames_per_block > 1) stays within tolerance of the direct load
    assert (np.abs(mags - m1) <= mag_tol(m1) / 10).all()
    a.free()


@pytest.mark.parametrize("N,hop", [(4096, 256), (32768, 375), (32768, 1024), (16384, 375), (32768, 3000)])
def test_pitch_only_and_mags_only(gpu_ctx, oracle, N, hop):
    """Either output alone is the same launch with the other store left out — for the sliding, circular-window,
    register-stored-row and direct-load kernels alike."""
    w = accum_sweep(2 * SR)
    a = gpu_ctx.upload(w)
    band = oracle.pitch_band(N, SR)
    m, p = gpu_ctx.stft_hop(a, N, hop, band=band)
    m2, none = gpu_ctx.stft_hop(a, N, hop, want_pitch=False, band=band)
    none2, p2 = gpu_ctx.stft_hop(a, N, hop, want_mags=False, band=band)
    assert none is None and none2 is None
    assert np.array_equal(m, m2) and np.array_equal(p, p2)
    a.free()


@pytest.mark.parametrize("N,hop", [(4096, 256), (32768, 375), (16384, 375), (32768, 1024)])
def test_linearity_and_silence(gpu_ctx, oracle, N, hop):
    """Size-independent properties: |STFT(c*x)| = c*|STFT(x)| for a power-of-two c (exact in fp32, also through the
    multiply chains of the sliding and circular-window kernels), silence -> zeros and pitch bin = kmin."""
    w = noisy(accum_sweep(3 * SR))
    a = gpu_ctx.upload(w)
    b = gpu_ctx.upload(w * 0.25)
    z = gpu_ctx.upload(np.zeros(3 * SR, np.float32))
    band = oracle.pitch_band(N, SR)
    ma, pa = gpu_ctx.stft_hop(a, N, hop, band=band)
    mb, pb = gpu_ctx.stft_hop(b, N, hop, band=band)
    mz, pz = gpu_ctx.stft_hop(z, N, hop, band=band)
    assert np.array_equal(ma * 0.25, mb)
    assert np.array_equal(pa["bin"], pb["bin"])
    assert not mz.any() and (pz["bin"] == band[0]).all() and not pz["mag"].any()
    for x in (a, b, z):
        x.free()


def test_frames_per_block_invariance(gpu_ctx):
    """hop 3000 loads every frame directly: any workgroup shape gives identical bits; hops 256 (slot-aligned sliding
    image) and 375 (circular image) carry products from frame to frame inside a workgroup, which may move results by a
    few 1e-7 of the frame peak, never more."""
    w = noisy(accum_sweep(2 * SR))
    a = gpu_ctx.upload(w)
    for hop, exact in ((3000, True), (375, False), (256, False)):
        ref = None
        for g in (1, 3, 16, 64, 1000):
            gpu_ctx.set_frames_per_block(g)
            m, p = gpu_ctx.stft_hop(a, 4096, hop)
            if ref is None:
                ref = (m, p)
            if exact:
                assert np.array_equal(m, ref[0]) and np.array_equal(p, ref[1])
            else:
                assert (np.abs(m - ref[0]) <= mag_tol(ref[0]) / 10).all()
                assert (np.abs(p["bin"] - ref[1]["bin"]) <= 1).all()
    gpu_ctx.set_frames_per_block(0)
    a.free()


def test_sinusoid_pitch(gpu_ctx):
    """A pure tone lands in the expected bin (known answer, independent of the oracle)."""
    n = 2 * SR
    t = np.arange(n) / SR
    for f in (110.0, 440.0, 1000.0):
        w = (0.5 * np.sin(2 * np.pi * f * t)).astype(np.float32)
        a = gpu_ctx.upload(w)
        _, p = gpu_ctx.stft_hop(a, 4096, 256, want_mags=False)
        assert (np.abs(p["bin"][40:-4] - f * 4096 / SR) <= 1.0).all()
        a.free()


def test_errors_newer_entry_points(gpu_ctx, mxlib):
    """Every failure is a negative status + message, never a crash: staged phase-vocoder calls out of order or with
    a missing neighbour seam, a shard request on too short a signal, misaligned device audio, bad colormap sizes."""
    import ctypes as C
    from melonix_amd import _capi
    L = _capi.lib()
    w = np.zeros(50000, np.float32)
    a = gpu_ctx.upload(w)
    c2 = mxlib.Context(0)
    with pytest.raises(mxlib.MxError):
        c2.pv_shard_synthesize(None)                      # no analyze before
    a2 = c2.upload(w)
    c2.pv_shard_analyze(a2, 3.0, 1, 2)
    with pytest.raises(mxlib.MxError):
        c2.pv_shard_synthesize(None)                      # rank 1 needs a carry
    c2.pv_shard_synthesize(np.zeros(2048, np.uint32))
    with pytest.raises(mxlib.MxError):
        c2.pv_shard_finish(10, None, None)                # rank 1 needs the previous rank's tail
    with pytest.raises(mxlib.MxError):
        mxlib.pv_shard_frames(5000, 0.0, 0, 8)            # 21 frames cannot feed 8 ranks
    with pytest.raises(mxlib.MxError):
        mxlib.pv_shard_frames(50000, 0.0, 3, 2)
    a2.free()
    c2.close()
    from conftest import loaded_hip
    hip = loaded_hip()  # the runtime the library itself uses (one HIP runtime per process)
    buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(buf), C.c_size_t(4 * (50000 + 2 * mxlib.MX_AUDIO_PAD) + 64)) == 0
    out = C.c_void_p()
    rc = L.mx_audio_wrap_device(gpu_ctx.handle, C.c_void_p(buf.value + 4), 50000, C.byref(out))
    assert rc == -1 and b"aligned" in L.mx_last_error()
    rc = L.mx_colormap_dev(gpu_ctx.handle, buf, 2049, 1.0, buf)
    assert rc == -1
    hip.hipFree(buf)
    a.free()


def test_errors(gpu_ctx, mxlib):
    a = gpu_ctx.upload(np.zeros(1000, np.float32))
    with pytest.raises(mxlib.MxError):
        gpu_ctx.stft_hop(a, 1024, 256)
    with pytest.raises(mxlib.MxError):
        gpu_ctx.stft_hop(a, 4096, 256, first=0, count=100)  # beyond ceil(n/hop)
    with pytest.raises(mxlib.MxError):
        gpu_ctx.stft_hop(a, 4096, 256, band=(100, 50))
    a.free()


@pytest.mark.parametrize("N", [4096, 16384, 32768])
def test_fused_colormap(gpu_ctx, oracle, N):
    """mx_stft_ranges_rgb = oracle colormap (spec-cache.cpp:77-96) of the GPU's own magnitude rows, byte for
    byte, for brightness settings that exercise all three segments; vs the oracle's own magnitudes the
    bytes may differ by one level where a bin sits on a truncation edge."""
    w = noisy(accum_sweep(10 * SR))
    ranges = [(48000, 48375), (0, 256), (-500, -100), (239744, 240000), (100000, 100001), (1000, 60000)]
    a = gpu_ctx.upload(w)
    mags, _ = gpu_ctx.stft_ranges(a, N, ranges)
    for k in (0.01, 512.0, 2.0 ** 15, 2.0 ** 19):  # k = 2^(brightness/10 + 9), app.cpp:75
        rgb = gpu_ctx.stft_ranges_rgb(a, N, ranges, k)
        want = np.stack([oracle.colormap(m, k) for m in mags])
        assert np.array_equal(rgb, want)
        ref = np.stack([oracle.colormap(oracle.spec_frame(w, N, s, e), k) for s, e in ranges])
        assert (np.abs(rgb.astype(int) - ref.astype(int)) <= 1).mean() > 0.999
    seg = np.stack([oracle.colormap(m, 2.0 ** 15) for m in mags])
    assert (seg[..., 1] > 0).any() and (seg[..., 2] > 0).any()  # the test really reaches segments 2 and 3
    # texels and magnitudes out of one launch: the rows are the plain ranges call's, bit for bit
    rgb2, mags2 = gpu_ctx.stft_ranges_rgb(a, N, ranges, 2.0 ** 15, want_mags=True)
    assert np.array_equal(mags2, mags) and np.array_equal(rgb2, seg)
    # a whole cold screen (1280 columns of 375 samples, spec-cache.cpp:63-65) in one call
    cols = np.stack([np.arange(1280) * 375, (np.arange(1280) + 1) * 375], axis=1).astype(np.int32)
    rgb3, mags3 = gpu_ctx.stft_ranges_rgb(a, N, cols, 2.0 ** 15, want_mags=True)
    pick = [0, 1, 63, 64, 640, 1278, 1279]
    assert np.array_equal(rgb3[pick], np.stack([oracle.colormap(mags3[i], 2.0 ** 15) for i in pick]))
    assert np.array_equal(mags3, gpu_ctx.stft_ranges(a, N, cols)[0])
    a.free()


@pytest.mark.parametrize("N,E", [(4096, 16), (16384, 32), (32768, 32)])
def test_device_equals_cpu_emulation(gpu_ctx, emu, N, E):
    """The kernels spell out every FMA and are built with -ffp-contract=off, so the CPU emulation of the
    same templates (tests/emu) reproduces the device arithmetic; only v_sqrt_f32 (1 ulp) vs sqrtf differs."""
    import ctypes as C
    w = noisy(accum_sweep(3 * SR))
    fp = C.POINTER(C.c_float)
    ranges = [(48000, 48375), (0, 256), (100000, 100001), (5000, 4000), (1000, 60000)]
    a = gpu_ctx.upload(w)
    mags, _ = gpu_ctx.stft_ranges(a, N, ranges)
    for i, (s_, e_) in enumerate(ranges):
        out = np.empty(N // 2, np.float32)
        assert emu.emu_stft_frame(N, E, w.ctypes.data_as(fp), len(w), s_, e_, 0, out.ctypes.data_as(fp)) == 0
        ulp = np.spacing(np.maximum(np.abs(out), np.float32(1e-30)))
        assert (np.abs(mags[i] - out) <= 2 * ulp).all(), float((np.abs(mags[i] - out) / ulp).max())
    a.free()


@pytest.mark.parametrize("N,E,hop", [(32768, 32, 375), (32768, 32, 512), (32768, 32, 300), (32768, 32, 77), (16384, 32, 375),
                                     (16384, 32, 500), (16384, 32, 129)])  # Tune::CIRC
def test_circular_window_equals_cpu_emulation(gpu_ctx, emu, N, E, hop):
    """Uniform hops that do not slide by whole slots run, at N = 16384 and 32768, the circular register image (stft_core.h): the device's
    frames — every workgroup restarts from a direct load — equal the CPU emulation of the same templates walked over the
    same workgroup, to the 2 ulp of v_sqrt_f32 vs sqrtf."""
    import ctypes as C
    w = noisy(accum_sweep(3 * SR))
    fp = C.POINTER(C.c_float)
    a = gpu_ctx.upload(w)
    g = 7
    gpu_ctx.set_frames_per_block(g)
    mags, _ = gpu_ctx.stft_hop(a, N, hop)
    gpu_ctx.set_frames_per_block(0)
    F = mags.shape[0]
    for blk in (0, 3, (F - 1) // g):  # first workgroups (frames that start before the file) and the last, ragged one
        first, count = blk * g, min(g, F - blk * g)
        out = np.empty((count, N // 2), np.float32)
        assert emu.emu_stft_circ(N, E, hop, w.ctypes.data_as(fp), len(w), first, count, out.ctypes.data_as(fp)) == 0
        ulp = np.spacing(np.maximum(np.abs(out), np.float32(1e-30)))
        d = np.abs(mags[first:first + count] - out)
        assert (d <= 2 * ulp).all(), float((d / ulp).max())
    a.free()


@pytest.mark.parametrize("N", [4096, 32768])
def test_kept_rows(gpu_ctx, oracle, mxlib, N):
    """mx_stft_ranges_keep / mx_rows_*: the rows that stay on the device are the plain ranges call's bit for bit, a span
    fetched later equals them, and re-colouring a span with another scale equals the oracle's colormap of those rows."""
    w = noisy(accum_sweep(3 * SR))
    a = gpu_ctx.upload(w)
    cols = np.array([(i * 375, (i + 1) * 375) for i in range(0, 300, 3)] + [(-500, -100), (5000, 4000)], dtype=np.int32)
    ref = gpu_ctx.stft_ranges(a, N, cols)[0]
    rows, rgb, mags = gpu_ctx.stft_ranges_keep(a, N, cols, k=2.0 ** 14, want_mags=True)
    assert len(rows) == len(cols)
    assert np.array_equal(mags, ref)
    assert np.array_equal(rgb, gpu_ctx.stft_ranges_rgb(a, N, cols, 2.0 ** 14))
    assert np.array_equal(rows.fetch(0, len(cols)), ref)
    assert np.array_equal(rows.fetch(17, 5), ref[17:22])
    again = rows.colormap(40, 9, 2.0 ** 16)
    assert np.array_equal(again, np.stack([oracle.colormap(ref[i], 2.0 ** 16) for i in range(40, 49)]))
    rows2, rgb2, mags2 = gpu_ctx.stft_ranges_keep(a, N, cols[:7])  # nothing comes back: rows only
    assert rgb2 is None and mags2 is None and np.array_equal(rows2.fetch(0, 7), ref[:7])
    with pytest.raises(mxlib.MxError):
        rows.fetch(len(cols) - 1, 2)
    with pytest.raises(mxlib.MxError):
        rows.colormap(-1, 1, 1.0)
    rows.free()
    rows2.free()
    a.free()


@pytest.mark.parametrize("N,hop", [(4096, 256), (4096, 375), (16384, 512), (32768, 375), (32768, 1024)])
def test_pitch_pick_is_the_argmax_of_its_own_row_for_any_band(gpu_ctx, N, hop):
    """The pick against the kernel's OWN magnitude rows, bit for bit: bin = lowest k in [kmin, kmax] holding the row's largest
    value there, mag = the row's value at that bin — for bands that start at bin 0, end at the last bin, consist of one bin,
    sit on the bins thread 0 owns (multiples of NS3 / 2) and for random ones; on silence (every in-band value equal: the pick
    is kmin), a sweep and noise.  Independent of the oracle: no tolerance."""
    M = N // 2
    rng = np.random.default_rng(N + hop)
    w = noisy(accum_sweep(3 * SR), level=0.02)
    w[:2 * N] = 0.0                      # whole frames of exact zeros: rows of zeros
    w[SR:SR + 3 * N] = 0.0
    a = gpu_ctx.upload(w)
    ns3 = M // (8 if N == 4096 else 16)
    bands = [(0, M - 1), (0, 0), (M - 1, M - 1), (1, 1), (ns3 // 2, ns3 // 2 + 3), (ns3, ns3), (0, ns3 // 2), (M // 2 - 1, M // 2 + 1),
             (M - ns3, M - 1)]
    for _ in range(6):
        lo = int(rng.integers(0, M))
        bands.append((lo, int(rng.integers(lo, M))))
    for kmin, kmax in bands:
        mags, pitch = gpu_ctx.stft_hop(a, N, hop, band=(kmin, kmax))
        sub = mags[:, kmin:kmax + 1]
        want = sub.argmax(axis=1) + kmin  # numpy: the first maximum = the lowest bin on ties
        assert np.array_equal(pitch["bin"], want), (kmin, kmax, int(np.nonzero(pitch["bin"] != want)[0][0]))
        assert np.array_equal(pitch["mag"], mags[np.arange(len(mags)), want])
        zero_rows = ~mags.any(axis=1)
        assert zero_rows.any() and (pitch["bin"][zero_rows] == kmin).all()
    a.free()
