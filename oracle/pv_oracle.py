"""pv_oracle.py — CPU restatement (numpy, binary64) of the BUILD-DEFINED phase-vocoder pitch shifter.

TEST INFRASTRUCTURE ONLY (same rule as melonix_oracle.h): tests/ and bench.py's CPU leg may import this;
nothing under melonix_amd/ may.

PARITY UNPINNED: the reference has no phase vocoder (its pitch shift is the granular
resampler of app.cpp:294-345, which melonix_amd reproduces bit for bit).  BASELINE.json's north_star names a
phase-vocoder / overlap-add resynthesis, so this build defines one; there is no reference arithmetic to match
and this file is the only oracle it can have.  The definition (N = 4096, Hs = 256, ratio r = 2^(st/12)):

  frames f = 0..F-1, F = ceil(n*r/Hs) + 1
  analysis centre   a_f = floor(f*Hs/r)                     (input advances Hs/r per frame: stretch by r)
  frame             x_f[j] = w[j] * x[a_f - N/2 + j],  w = periodic Hann, zeros outside the file
  spectrum          X_f[k] = sum_j x_f[j] e^{-2 pi i jk/N} / N,  k = 0..N/2-1    (Nyquist bin dropped)
  phase in turns    P_f[k] = 2 * round(arg X_f[k] / 2pi * 2^31) mod 2^32      (uint32, even: the product keeps the
                    bin's activity flag in the spare low bit of the same word)
  hop               h_f = a_f - a_{f-1}
  deviation         d = int32( P_f - P_{f-1} - (k*h_f mod N) * 2^32/N )        (wraps to [-1/2, 1/2) turn)
  synthesis advance inc = (k*Hs mod N) * 2^32/N + trunc(float64(d) * (Hs / h_f))   (uint32; one binary64 product,
                    Hs/h_f itself a binary64 quotient: the same two roundings on every IEEE machine)
  active bin        act_f[k] = |X_f[k]| >= 1e-3 * max_k |X_f[k]|              (60 dB below the frame's peak)
  peak              act_f[k] and |X_f[k]| >= rho * |X_f[k+-1]|, rho * |X_f[k+-2]|,  rho = 1 - 2^-10
                    (bins outside 0..N/2-1 never stand in the way).  The margin makes near-ties peaks on both
                    sides instead of leaving them to rounding: the two bins straddling a partial are both peaks
                    (they are coherent anyway), and in a flat spectrum — an impulse, silence — every bin is its own
                    peak, i.e. plain per-bin propagation
  owner             p_f(k) = the peak nearest to k among those at most 32 bins away (a tie goes to the lower
                    bin); a bin with no such peak has no owner
  synthesis phase   IDENTITY PHASE LOCKING (Laroche & Dolson): only peaks propagate a phase, every other bin
                    rides on its peak.  With p = p_f(k):
                      Phi_f[k] = Phi_{f-1}[p] + inc_f[p] + (P_f[k] - P_f[p])   if act_f[p] and act_{f-1}[p]
                      Phi_f[k] = P_f[k]                                        otherwise (no owner; frame 0;
                                 h_f < 1; the peak's bin carried no signal in the previous frame)
                    (uint32 wrap = mod 1 turn).  Phi_{f-1}[p] is whatever bin p held in the previous frame, so a
                    peak that moves to a neighbouring bin continues from the phase that bin already had as part
                    of the same lobe: a tone, a sweep and a vibrato keep their level, where independent bins lose
                    a quarter of it on a sweep.  A bin that carries no signal leaves no trace.
  synthesis frame   y_f[j] = Re sum_k c_k |X_f[k]| e^{2 pi i (Phi_f[k]/2^32 + jk/N)},  c_0 = 1, c_k = 2
  overlap-add       s[f*Hs - N/2 + j] += w[j] * y_f[j];   s /= sum_f w^2 = 3N/(8 Hs) = 6
  resample          out[i] = (1-t) s[m] + t s[m+1],  m = floor(i*r), t = i*r - m,  i = 0..n-1

The phase bookkeeping is integer (the one binary64 product is rounded identically everywhere) and each frame's
update is a map k -> (source bin, delta) or a restart, so composing those maps in any grouping — a parallel
scan over frames — gives exactly the serial result.

How the product evaluates this definition (round 4; melonix_amd/csrc/pv_kernels.hip): with identity phase locking
Phi_f[k] - P_f[k] = C_f[p] is one number per PEAK (p = the owner of k), so the synthesis coefficient
|X_f[k]| e^{2 pi i Phi_f[k]/2^32} equals X_f[k] e^{2 pi i C_f[p]/2^32} up to the 2^-31-turn quantisation of P_f[k]
(3e-9 rad).  The product therefore keeps phases at the peaks only — the same uint32 arithmetic as above on the same
values — and rotates the complex analysis bins by their owner's phasor; bins without a continuing owner keep X_f[k]
as it is.  This file stays the definition: the per-bin form below is what the tests compare against.
"""
import numpy as np

N = 4096
HS = 256


def ratio(semitones: float) -> float:
    return float(2.0 ** (float(semitones) / 12.0))


def plan(n: int, r: float):
    F = int(np.ceil(n * r / HS)) + 1
    a = np.floor(np.arange(F, dtype=np.float64) * HS / r).astype(np.int64)
    return F, a


def window():
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(N) / N)


def analysis(x, a, chunk=256):
    """-> mags (F, N/2) f64 = |X|/N, phases (F, N/2) uint32 turns."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    w = window()
    F = len(a)
    mags = np.empty((F, N // 2))
    ph = np.empty((F, N // 2), dtype=np.uint32)
    xp = np.concatenate([np.zeros(N), x, np.zeros(N + HS)])  # zeros outside the file
    j = np.arange(N)
    for f0 in range(0, F, chunk):
        af = a[f0:f0 + chunk]
        idx = af[:, None] - N // 2 + j[None, :] + N  # a frame never leaves [-N/2, n + Hs + N/2)
        fr = xp[idx] * w[None, :]
        X = np.fft.rfft(fr, axis=1)[:, : N // 2] / N
        mags[f0:f0 + chunk] = np.abs(X)
        turns = np.angle(X) / (2.0 * np.pi)
        ph[f0:f0 + chunk] = ((np.rint(turns * 2147483648.0).astype(np.int64) << 1) & 0xFFFFFFFF).astype(np.uint32)
    return mags, ph


ACTIVE_REL = 1e-3


LOCK_REACH = 32  # bins
PEAK_MARGIN = 1.0 - 2.0 ** -10


def peaks(m, act):
    """Peak mask of one frame."""
    e = np.concatenate([np.full(2, -1.0), np.asarray(m, dtype=np.float64), np.full(2, -1.0)])
    c = e[2:-2]
    rho = PEAK_MARGIN
    return act & (c >= rho * e[1:-3]) & (c >= rho * e[0:-4]) & (c >= rho * e[3:-1]) & (c >= rho * e[4:])


def owners(m, act):
    """Owner peak of every bin of one frame (int64; -1 = none)."""
    M = len(m)
    idx = np.flatnonzero(peaks(m, act))
    own = np.full(M, -1, dtype=np.int64)
    if len(idx):
        k = np.arange(M)
        j = np.searchsorted(idx, k)  # first peak >= k
        right = np.where(j < len(idx), idx[np.minimum(j, len(idx) - 1)], 1 << 40)
        jl = np.searchsorted(idx, k, side="right") - 1  # last peak <= k
        left = np.where(jl >= 0, idx[np.maximum(jl, 0)], -(1 << 40))
        dl, dr = k - left, right - k
        own = np.where(dl <= dr, left, right)
        own = np.where(np.minimum(dl, dr) <= LOCK_REACH, own, -1)
    return own


def synthesis_phases(ph, a, mags, allow_stall=False):
    """Integer phase propagation with identity phase locking -> Phi (F, N/2) uint32.  allow_stall: frames whose
    analysis position did not advance (h < 1) restart every bin (marker-driven variant); the constant-ratio plan
    never has them."""
    F = len(a)
    act = mags >= np.float32(ACTIVE_REL) * mags.max(axis=1, keepdims=True)
    k = np.arange(N // 2, dtype=np.int64)
    unit = 4294967296 // N
    h = np.diff(a)
    assert allow_stall or (h >= 1).all(), "ratio too large: the analysis hop must stay >= 1 sample"
    Phi = np.zeros((F, N // 2), dtype=np.uint32)
    Phi[0] = ph[0]
    for f in range(1, F):
        if h[f - 1] < 1:
            Phi[f] = ph[f]
            continue
        expect = ((k * int(h[f - 1])) % N) * unit
        d = (ph[f].astype(np.int64) - ph[f - 1].astype(np.int64) - expect) & 0xFFFFFFFF
        d = np.where(d >= 2147483648, d - 4294967296, d)  # int32 reinterpretation
        q = np.trunc(d.astype(np.float64) * (np.float64(HS) / np.float64(int(h[f - 1])))).astype(np.int64)
        inc = (((k * HS) % N) * unit + q) & 0xFFFFFFFF
        cont = act[f] & act[f - 1]
        own = owners(mags[f], act[f])
        p = np.where(own >= 0, own, 0)
        cur = ph[f].astype(np.int64)
        locked = (Phi[f - 1].astype(np.int64)[p] + inc[p] + cur - cur[p]) & 0xFFFFFFFF
        Phi[f] = np.where((own >= 0) & cont[p], locked, cur).astype(np.uint32)
    return Phi


def synthesis(mags, Phi, chunk=256):
    """Overlap-added, normalised stretched signal s (index 0 = stretched time -N/2)."""
    F = mags.shape[0]
    w = window()
    s = np.zeros(F * HS + N)
    for f0 in range(0, F, chunk):
        m = mags[f0:f0 + chunk]
        Y = np.zeros((m.shape[0], N // 2 + 1), dtype=np.complex128)
        Y[:, : N // 2] = m * np.exp(2j * np.pi * (Phi[f0:f0 + chunk].astype(np.float64) / 4294967296.0))
        # irfft treats bin 0 (and the zeroed Nyquist) as real: Re of the one-sided sum with c_0 = 1, c_k = 2
        y = np.fft.irfft(Y, n=N, axis=1) * N
        for i in range(m.shape[0]):
            f = f0 + i
            s[f * HS: f * HS + N] += w * y[i]
    return s / (3.0 * N / (8.0 * HS))


def resample(s, n, r):
    pos = np.arange(n, dtype=np.float64) * r + N // 2  # s[0] is stretched time -N/2
    m = np.floor(pos).astype(np.int64)
    t = pos - m
    return (1.0 - t) * s[m] + t * s[m + 1]


def pitch_shift(x, semitones):
    """-> float64 PCM of len(x) samples."""
    r = ratio(semitones)
    F, a = plan(len(x), r)
    mags, ph = analysis(x, a)
    Phi = synthesis_phases(ph, a, mags)
    s = synthesis(mags, Phi)
    return resample(s, len(x), r)


# ---- marker-driven variant ------------------------------------------------------------------------------
# The same vocoder steered by the editor's markers the way App::exportWav is (app.cpp:1194-1207): the output runs
# over warped time t in [0, duration()); at warped time t the source is read around time2Sample(t) and shifted by
# 2^(time2PitchBend(t)/12) (app.cpp:296-301).  Per frame (the bend is taken constant over a frame's hop):
#   t_0 = 0;  r_f = 2^(pb(t_f)/12) (the C library's binary64 exp2 of the binary32 bend / 12);  a_f = time2Sample(t_f);
#   i0_f = ceil(t_f * sr)  (first output sample at or after t_f);  t_{f+1} = t_f + Hs / (r_f * sr)
#   frames until t_f >= duration(), that frame included
#   output sample i in [i0_f, i0_{f+1}):  u = f*Hs + (i/sr - t_f) * r_f * sr   (stretched position), lerp of s there
# A bin also restarts whenever the analysis position does not advance (h_f < 1: a time warp that stalls or runs
# backwards).  n_out = number of i with i/sr < duration().
def _libm_exp2(x):
    """exp2 of the C library (the product's host code calls std::exp2; numpy's own exp2 differs in the last bit)."""
    import ctypes
    import ctypes.util
    global _LIBM
    try:
        _LIBM
    except NameError:
        _LIBM = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        _LIBM.exp2.restype = ctypes.c_double
        _LIBM.exp2.argtypes = [ctypes.c_double]
    return float(_LIBM.exp2(float(x)))


def marker_plan(n, sr, markers):
    from oracle import pyoracle as O
    tm = O.TimeMap(list(markers), sr, n, memo=False)
    dur = tm.duration()
    n_out = int(np.ceil(dur * sr - 1e-12)) if dur > 0 else 0
    while n_out > 0 and (n_out - 1) / sr >= dur:
        n_out -= 1
    while n_out / sr < dur:
        n_out += 1
    t, a, rf, tf, i0 = 0.0, [], [], [], []
    while True:
        pb = float(np.float32(tm.time2pitchbend(t)))
        r = _libm_exp2(pb / 12.0)
        tf.append(t)
        rf.append(r)
        a.append(int(tm.time2sample(t)))
        i0.append(min(n_out, int(np.ceil(t * sr))))
        if t >= dur:
            break
        t = t + HS / (r * sr)
    i0.append(n_out)
    return n_out, np.array(a, np.int64), np.array(tf), np.array(rf), np.array(i0, np.int64)


def render(x, sr, markers):
    """-> float64 PCM of n_out samples (warped duration)."""
    x = np.asarray(x, dtype=np.float64)
    n_out, a, tf, rf, i0 = marker_plan(len(x), sr, markers)
    mags, ph = analysis(x, a)
    Phi = synthesis_phases(ph, a, mags, allow_stall=True)
    s = synthesis(mags, Phi)
    out = np.zeros(n_out)
    for f in range(len(a)):
        lo, hi = int(i0[f]), int(i0[f + 1])
        if hi <= lo:
            continue
        i = np.arange(lo, hi, dtype=np.float64)
        u = f * HS + (i / sr - tf[f]) * rf[f] * sr + N // 2
        m = np.floor(u).astype(np.int64)
        w = u - m
        out[lo:hi] = (1.0 - w) * s[m] + w * s[m + 1]
    return out
