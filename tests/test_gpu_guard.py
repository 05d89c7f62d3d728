"""Guard bands around every device output of the `_dev` entry points (round-4 verdict: the row stores are hand-addressed —
`global_store_dword ... offset:imm` on an SGPR base with a 32-bit lane offset, stft_kernel_impl.h — and the whole-config
comparisons would miss a stray write OUTSIDE mags / pitch / rgb / pcm).  Every output lives inside a larger allocation filled
with a sentinel byte, 64 KiB either side; each call runs twice, once with the output at the start of the payload area and once
shifted by an odd number of ELEMENTS (rows no longer 16-byte aligned): the sentinels must survive both, and the shifted output
must be the unshifted one bit for bit.  Covers, at all three transform sizes: the sliding, circular-window and direct bulk
kernels (aligned and unaligned sample loads), the deferred row stores of the two-wave plan, ranges mode, the texel kernels;
first frames other than 0, runs cut short by the end of the launch, one-frame launches; then the colormap, the resynthesis
(f32 and int16), the waveform pyramid.  (mx_grain_table_dev hands back host tables: its device buffers are the library's
own.  The phase vocoder's outputs have their guard test in tests/test_pv.py.)"""
import ctypes as C

import numpy as np
import pytest

from conftest import SR, accum_sweep, loaded_hip, noisy

pytestmark = pytest.mark.gpu

G = 64 * 1024      # bytes of sentinel either side
SENT = 0xA5


class Guarded:
    """`nbytes` of payload at `shift` bytes into the payload area of a sentinel-filled device allocation."""

    def __init__(self, hip, nbytes, shift=0):
        self.hip, self.nbytes, self.shift = hip, int(nbytes), int(shift)
        self.total = G + self.shift + self.nbytes + G
        self.total += (-self.total) % 256
        self.base = C.c_void_p()
        assert hip.hipMalloc(C.byref(self.base), C.c_size_t(self.total)) == 0
        assert hip.hipMemset(self.base, C.c_int(SENT), C.c_size_t(self.total)) == 0
        assert hip.hipDeviceSynchronize() == 0
        self.ptr = self.base.value + G + self.shift

    def fetch(self):
        """-> the payload's bytes; asserts that nothing outside it was touched."""
        host = np.empty(self.total, dtype=np.uint8)
        assert self.hip.hipDeviceSynchronize() == 0
        assert self.hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), self.base, C.c_size_t(self.total), C.c_int(2)) == 0  # D2H
        lo, hi = G + self.shift, G + self.shift + self.nbytes
        before, after = host[:lo], host[hi:]
        assert (before == SENT).all(), f"{int((before != SENT).sum())} bytes written BEFORE the output (first at {int(np.argmax(before != SENT)) - lo})"
        assert (after == SENT).all(), f"{int((after != SENT).sum())} bytes written BEHIND the output (first at +{int(np.argmax(after != SENT))})"
        return host[lo:hi].copy()

    def free(self):
        self.hip.hipFree(self.base)


def _to_device(hip, arr):
    arr = np.ascontiguousarray(arr)
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(max(arr.nbytes, 16))) == 0
    assert hip.hipMemcpy(p, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), C.c_int(1)) == 0  # H2D
    return p


@pytest.fixture(scope="module")
def hip(gpu_ctx):
    return loaded_hip()


@pytest.fixture(scope="module")
def audio10(gpu_ctx):
    w = noisy(accum_sweep(10 * SR))
    a = gpu_ctx.upload(w)
    yield w, a
    a.free()


def _twice(hip, sizes, shifts, call):
    """Runs `call(ptrs)` with every output unshifted and then shifted (bytes, per output); returns the unshifted payloads
    after checking guards both times and that the shifted payloads equal them."""
    ref = None
    for use_shift in (False, True):
        bufs = [Guarded(hip, n, s if use_shift else 0) for n, s in zip(sizes, shifts)]
        try:
            call([b.ptr for b in bufs])
            got = [b.fetch() for b in bufs]
        finally:
            for b in bufs:
                b.free()
        if ref is None:
            ref = got
        else:
            for i, (x, y) in enumerate(zip(ref, got)):
                assert np.array_equal(x, y), f"output {i}: shifted by {shifts[i]} bytes it differs from the unshifted run"
    return ref


# (N, hop): which bulk kernel a uniform hop reaches (stft_kernels.hip launch_plan)
BULK = [
    (4096, 256), (4096, 512),          # sliding window, deferred row stores
    (4096, 300), (4096, 375),          # direct loads: aligned, unaligned
    (16384, 512), (16384, 1024),       # sliding window, rows straight from registers
    (16384, 300), (16384, 375),        # circular window
    (16384, 2000), (16384, 2001),      # direct loads: aligned, unaligned
    (32768, 1024),                     # sliding window
    (32768, 375), (32768, 512),        # circular window
    (32768, 2000), (32768, 2001),      # direct loads
]


@pytest.mark.parametrize("N,hop", BULK)
def test_bulk_rows_and_pitch_stay_inside(gpu_ctx, hip, audio10, N, hop):
    w, a = audio10
    F = (len(w) + hop - 1) // hop
    M = N // 2
    # a first frame that is no run head, a count that ends inside a run; the signal's last frames (reads run into the pad); one frame
    for first, count in ((3, 77), (max(0, F - 41), min(41, F)), (5, 1)):
        def call(p, first=first, count=count):
            gpu_ctx.stft_hop_dev(a, N, hop, first, count, p[0], p[1])
        mags, pitch = _twice(hip, [count * M * 4, count * 8], [4 * 1, 8 * 1], call)
        m = mags.view(np.float32).reshape(count, M)
        assert np.isfinite(m).all() and (m >= 0).all()  # (every row was written: no sentinel floats left, 0xA5A5A5A5 is negative)
        pk = pitch.view(np.dtype([("bin", "<i4"), ("mag", "<f4")]))
        assert (pk["bin"] >= 0).all() and (pk["bin"] < M).all()

        # pitch only (no rows materialised) and rows only
        def call_p(p, first=first, count=count):
            gpu_ctx.stft_hop_dev(a, N, hop, first, count, None, p[0])
        (p_only,) = _twice(hip, [count * 8], [8 * 3], call_p)
        assert np.array_equal(p_only, pitch)

        def call_m(p, first=first, count=count):
            gpu_ctx.stft_hop_dev(a, N, hop, first, count, p[0], None)
        (m_only,) = _twice(hip, [count * M * 4], [4 * 3], call_m)
        assert np.array_equal(m_only, mags)


@pytest.mark.parametrize("N", [4096, 16384, 32768])
def test_ranges_rows_texels_and_pitch_stay_inside(gpu_ctx, hip, audio10, N):
    w, a = audio10
    M = N // 2
    ranges = np.array([(48000, 48375), (0, 256), (-500, -100), (len(w) - 100, len(w) + 300), (100000, 100001), (1000, 60000),
                       (479000, 470000)], dtype=np.int32)
    count = len(ranges)
    d_r = _to_device(hip, ranges)
    try:
        def call(p):
            gpu_ctx.stft_ranges_dev(a, N, d_r.value, count, p[0], p[1])
        mags, pitch = _twice(hip, [count * M * 4, count * 8], [4 * 1, 8 * 1], call)
        host_m, host_p = gpu_ctx.stft_ranges(a, N, ranges)
        assert np.array_equal(mags.view(np.float32).reshape(count, M), host_m)
        assert np.array_equal(pitch, host_p.view(np.uint8).reshape(-1))
        from melonix_amd import _capi
        L = _capi.lib()
        for k in (512.0, 2.0 ** 17):
            def call_t(p, k=k):
                _capi.check(L.mx_stft_ranges_rgb_dev(gpu_ctx.handle, a.handle, N, d_r, count, k, C.c_void_p(p[0]), C.c_void_p(p[1])))
            m2, rgb = _twice(hip, [count * M * 4, count * M * 3], [4 * 1, 1], call_t)
            assert np.array_equal(m2, mags)
            assert np.array_equal(rgb.reshape(count, M, 3), gpu_ctx.stft_ranges_rgb(a, N, ranges, k))

            def call_t0(p, k=k):  # texels alone
                _capi.check(L.mx_stft_ranges_rgb_dev(gpu_ctx.handle, a.handle, N, d_r, count, k, None, C.c_void_p(p[0])))
            (rgb0,) = _twice(hip, [count * M * 3], [3], call_t0)
            assert np.array_equal(rgb0, rgb)

            def call_c(p, k=k):  # the colormap alone, over the device rows of the first run
                d_m = _to_device(hip, mags)
                try:
                    _capi.check(L.mx_colormap_dev(gpu_ctx.handle, d_m, count * M, k, C.c_void_p(p[0])))
                    assert hip.hipDeviceSynchronize() == 0
                finally:
                    hip.hipFree(d_m)
            (rgb1,) = _twice(hip, [count * M * 3], [1], call_c)
            assert np.array_equal(rgb1, rgb)
    finally:
        hip.hipFree(d_r)


def test_resynth_pcm_stays_inside(gpu_ctx, hip, mxlib):
    n = 3 * SR
    w = accum_sweep(n)
    a = gpu_ctx.upload(w)
    try:
        s, l = gpu_ctx.grains_dev(a)
        for mk in ([(1, 0, 0, 3.0), (n - 1, 0, 0, 3.0)], [(1, 0, 0, -5.0), (n // 2, 0, 0.25, 7.0), (n - 1, 0, 0, 0.0)]):
            st, total = mxlib.schedule_build(w, SR, s, l, mk)
            d_st = _to_device(hip, st)
            try:
                def call(p):
                    gpu_ctx.resynth_dev(a, d_st.value, len(st), total, p[0], p[1])
                f32, i16 = _twice(hip, [total * 4, total * 2], [4 * 1, 2 * 1], call)
                hf, hi = gpu_ctx.resynth(a, st, total)
                assert np.array_equal(f32.view(np.float32).view(np.uint32), hf.view(np.uint32)) and np.array_equal(i16.view(np.int16), hi)

                def call_i(p):
                    gpu_ctx.resynth_dev(a, d_st.value, len(st), total, None, p[0])
                (only16,) = _twice(hip, [total * 2], [2 * 3], call_i)
                assert np.array_equal(only16, i16)
            finally:
                hip.hipFree(d_st)
    finally:
        a.free()


@pytest.mark.parametrize("n", [1 << 20, 3 * SR + 17, 4099])
def test_pyramid_stays_inside(gpu_ctx, hip, n):
    from melonix_amd import _capi
    L = _capi.lib()
    w = noisy(accum_sweep(n))
    a = gpu_ctx.upload(w)
    try:
        counts = np.zeros(64, dtype=np.int64)
        nl = C.c_int()

        def call(p):
            _capi.check(L.mx_minmax_pyramid_dev(gpu_ctx.handle, a.handle, C.c_void_p(p[0]), counts.ctypes.data_as(C.c_void_p), C.byref(nl)))
        # (the documented capacity is 2 n floats; what is written is 2 * sum(counts): the rest must keep the sentinel too)
        (raw,) = _twice(hip, [2 * n * 4], [4 * 1], call)
        used = 2 * int(counts[: nl.value].sum())
        assert (raw[4 * used:] == SENT).all(), "the pyramid wrote beyond its last level"
        want = gpu_ctx.minmax_pyramid(a)
        got = raw[: 4 * used].view(np.float32)
        assert np.array_equal(got.view(np.uint32), np.concatenate([x.reshape(-1) for x in want]).view(np.uint32))
    finally:
        a.free()


def test_the_guard_catches_an_overrun(gpu_ctx, hip, audio10):
    """The harness itself: four rows into a payload declared three rows long must trip the sentinel check behind it, and an
    output pointer handed over one row early the one in front."""
    w, a = audio10
    M = 2048
    g = Guarded(hip, 3 * M * 4)
    try:
        gpu_ctx.stft_hop_dev(a, 4096, 256, 0, 4, g.ptr, None)
        with pytest.raises(AssertionError, match="BEHIND"):
            g.fetch()
    finally:
        g.free()
    g = Guarded(hip, 3 * M * 4)
    try:
        gpu_ctx.stft_hop_dev(a, 4096, 256, 0, 3, g.ptr - M * 4, None)
        with pytest.raises(AssertionError, match="BEFORE"):
            g.fetch()
    finally:
        g.free()
