// stft_core.h — per-thread building blocks of the LDS-resident real-input FFT
// behind Spec::internalGetSpec (reference spec.cpp:44-66).
//
// The same text compiles for gfx950 (hipcc) and for the host (g++), so that
// tests/emu can run one workgroup thread-by-thread on the CPU and check the
// index maps, swizzles and twiddles against the oracle before a GPU is
// involved.  Nothing here is a CPU fallback: the product only instantiates
// these templates inside __global__ kernels (stft_kernels.hip).
//
// Scheme (N real samples, M = N/2 packed complex points, 32 points per thread,
// T = M/32 threads per frame = one workgroup per hop):
//   pass 1  radix R1, straight from the windowed samples in HBM
//   -- transposition T1 through LDS (in place, XOR-swizzled) --
//   pass 2  radix R2 with twiddles exp(-2*pi*i*r*k/(R1*R2))
//   -- transposition T2 through LDS --
//   pass 3  radix 16 with twiddles exp(-2*pi*i*r*k0/M); thread t owns the
//           butterflies k0 = t and NS3 - t, i.e. both members of every
//           (k, M-k) pair the real-FFT split needs, so the split, the
//           magnitude and the pitch pick never leave registers.
//   N = 4096 : R = 8,16,16  T = 64  (one wavefront per frame, 16 KiB LDS)
//   N = 16384: R = 32,16,16 T = 256 (64 KiB LDS)
//   N = 32768: R = 32,32,16 T = 512 (128 KiB LDS; the reference's SpectrSize)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MX_HD __host__ __device__ __forceinline__
namespace mx {
using cpx = float2;
}
#else
#define MX_HD inline __attribute__((always_inline))
namespace mx {
struct alignas(8) cpx {
  float x, y;
};
}  // namespace mx
#endif

namespace mx {

#include "stft_consts.inc"

MX_HD cpx mk(float x, float y) {
  cpx r;
  r.x = x;
  r.y = y;
  return r;
}
MX_HD cpx cadd(cpx a, cpx b) { return mk(a.x + b.x, a.y + b.y); }
MX_HD cpx csub(cpx a, cpx b) { return mk(a.x - b.x, a.y - b.y); }
MX_HD cpx cmul(cpx a, cpx b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
MX_HD cpx cconj(cpx a) { return mk(a.x, -a.y); }
// by-value select (a conditional on two array lvalues would select addresses and
// push the register array into scratch)
MX_HD cpx csel(bool c, cpx a, cpx b) { return mk(c ? a.x : b.x, c ? a.y : b.y); }
MX_HD float fast_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);  // v_sqrt_f32, 1 ulp
#else
  return __builtin_sqrtf(x);
#endif
}

// a * exp(-2*pi*i*K/64), K a compile-time constant.
template <int K>
MX_HD cpx mulw64(cpx a) {
  constexpr int k = ((K % 64) + 64) % 64;
  constexpr float h = 0.707106781187f;
  if constexpr (k == 0) return a;
  else if constexpr (k == 16) return mk(a.y, -a.x);
  else if constexpr (k == 32) return mk(-a.x, -a.y);
  else if constexpr (k == 48) return mk(-a.y, a.x);
  else if constexpr (k == 8) return mk(h * (a.x + a.y), h * (a.y - a.x));
  else if constexpr (k == 24) return mk(h * (a.y - a.x), -h * (a.x + a.y));
  else if constexpr (k == 40) return mk(-h * (a.x + a.y), h * (a.x - a.y));
  else if constexpr (k == 56) return mk(h * (a.x - a.y), h * (a.x + a.y));
  else {
    constexpr float c = kCos64[k], s = kSin64[k];
    return mk(a.x * c + a.y * s, a.y * c - a.x * s);
  }
}

// ---- in-register DFT of size R (natural order in, natural order out) --------
template <int R, int Q>
struct Combine {
  static MX_HD void run(const cpx *E, const cpx *O, cpx *out) {
    const cpx t = mulw64<Q * 64 / R>(O[Q]);
    out[Q] = cadd(E[Q], t);
    out[Q + R / 2] = csub(E[Q], t);
    if constexpr (Q + 1 < R / 2) Combine<R, Q + 1>::run(E, O, out);
  }
};

template <int R>
struct Dft {
  static MX_HD void run(const cpx *in, cpx *out) {
    cpx e[R / 2], o[R / 2], E[R / 2], O[R / 2];
#pragma unroll
    for (int q = 0; q < R / 2; ++q) {
      e[q] = in[2 * q];
      o[q] = in[2 * q + 1];
    }
    Dft<R / 2>::run(e, E);
    Dft<R / 2>::run(o, O);
    Combine<R, 0>::run(E, O, out);
  }
};
template <>
struct Dft<2> {
  static MX_HD void run(const cpx *in, cpx *out) {
    out[0] = cadd(in[0], in[1]);
    out[1] = csub(in[0], in[1]);
  }
};
template <>
struct Dft<1> {
  static MX_HD void run(const cpx *in, cpx *out) { out[0] = in[0]; }
};

// ---- geometry -------------------------------------------------------------
template <int N_>
struct Cfg {
  static constexpr int N = N_;
  static constexpr int M = N / 2;   // packed complex points
  static constexpr int E = 32;      // points per thread
  static constexpr int T = M / E;   // threads per frame
  static constexpr int R1 = (N == 4096) ? 8 : 32;
  static constexpr int R2 = (N == 32768) ? 32 : 16;
  static constexpr int R3 = 16;
  static constexpr int NS3 = R1 * R2;  // finished sub-transform size entering pass 3 (= M/16)
  static constexpr int NB1 = E / R1;   // butterflies per thread in pass 1
  static constexpr int NB2 = E / R2;
  static constexpr int TW2 = (R2 - 1) * R1;   // entries of the pass-2 twiddle table
  static constexpr int TW3 = (R3 - 1) * NS3;  // entries of the pass-3 twiddle table
  static_assert(N == 4096 || N == 16384 || N == 32768, "supported FFT sizes");
  static_assert(R1 * R2 * R3 == M, "radix plan must cover M");
};

// XOR swizzles of the complex index (8-byte granules) inside the LDS image.
// T1 is written with lane stride R1 (pass-1 outputs) and read contiguously;
// T2 is written in runs of R1 and read contiguously.  Both keep every aligned
// block of 32 complex points a permutation of itself, so contiguous reads stay
// conflict-free while the strided writes spread over all banks
// (tools/lds_sim.py checks this against the gfx950 lane-group model).
template <int N>
MX_HD int swz1(int i) {
  if constexpr (Cfg<N>::R1 == 8) return i ^ ((i >> 3) & 15);
  else return i ^ ((i >> 5) & 15);
}
template <int N>
MX_HD int swz2(int i) {
  if constexpr (Cfg<N>::R1 == 8) return i ^ (((i >> 7) & 1) << 3);
  else return i;
}

// ---- pass 1: windowed samples -> radix-R1 butterflies ----------------------
// x points at the frame's first sample (file index end-N); w at the weight of
// that sample.  WSTEP = +1: w[p] (bulk table, forward); WSTEP = -1: w[-p]
// (the d-indexed table walked downwards, ranges mode).  ALIGNED8: x and w are
// 8-byte aligned so the pair (2m, 2m+1) is one 64-bit load.
struct alignas(4) f2u {  // 4-byte aligned pair for frames starting at odd samples
  float x, y;
};

// The windowed frame as the thread sees it: Y[e], e = b + NB1*r, is the packed complex
// point c = t + T*e  (samples 2c, 2c+1 of the frame), i.e. input r of pass-1 butterfly
// j = t + T*b.  The window tables carry the output scale 1/(2N) (a power of two, so
// x*(w*2^-k) == (x*w)*2^-k bit for bit): magnitudes come out of the split already
// scaled and the per-bin multiply disappears.
template <int N, int WSTEP, bool ALIGNED8>
MX_HD void load_frame(int t, cpx (&Y)[32], const float *x, const float *w) {
#pragma clang fp contract(off)  // the windowed sample is a rounded binary32 product (spec.cpp:58)
  using C = Cfg<N>;
#pragma unroll
  for (int e = 0; e < 32; ++e) {
    const int p = 2 * (t + C::T * e);
    float x0, x1, w0, w1;
    if constexpr (ALIGNED8 && WSTEP == 1) {
      const cpx xs = *reinterpret_cast<const cpx *>(x + p);
      const cpx ws = *reinterpret_cast<const cpx *>(w + p);
      x0 = xs.x; x1 = xs.y; w0 = ws.x; w1 = ws.y;
    } else {
      const f2u xs = *reinterpret_cast<const f2u *>(x + p);
      x0 = xs.x; x1 = xs.y;
      if constexpr (WSTEP == 1) {
        const f2u ws = *reinterpret_cast<const f2u *>(w + p);
        w0 = ws.x; w1 = ws.y;
      } else {
        const f2u ws = *reinterpret_cast<const f2u *>(w - p - 1);
        w0 = ws.y; w1 = ws.x;
      }
    }
#ifdef MX_ABL_NOW
    w0 = 0.75f; w1 = 0.5f;
#endif
#ifdef MX_ABL_NOX
    x0 = (float)p; x1 = 1.0f;
#endif
    Y[e] = mk(x0 * w0, x1 * w1);  // float product, as spec.cpp:58 (times the folded 2^-k)
  }
}

template <int N>
MX_HD void pass1(const cpx (&Y)[32], cpx (&v)[32]) {
  using C = Cfg<N>;
#pragma unroll
  for (int b = 0; b < C::NB1; ++b) {
    cpx in[C::R1], out[C::R1];
#pragma unroll
    for (int r = 0; r < C::R1; ++r) in[r] = Y[b + C::NB1 * r];
    Dft<C::R1>::run(in, out);
#pragma unroll
    for (int r = 0; r < C::R1; ++r) v[b * C::R1 + r] = out[r];
  }
}

// ---- sliding window (uniform hop) -------------------------------------------
// Consecutive frames overlap by N-hop samples and the one-sided exponential window
// is shift-invariant up to a constant: w(p - hop) = w(p) * exp(-2.5e-4*hop).  So the
// next frame's windowed points are this frame's, moved down by D = (hop/2)/T slots
// and decayed:  Y'[e] = Y[e+D] * g.  The register move and the decay are the same
// multiply.  Points that just left the flat (weight 1) tail get their exact table
// weight instead of g, the newest hop enters with weight 1: each sample is loaded
// from HBM once per workgroup instead of N/hop times.  A point is decayed at most
// N/hop - 2 times before it leaves the frame, so the weights stay within
// (N/hop)*2^-24 relative of the expf table (tests bound the end-to-end effect).
template <int N, int HOP>
struct Slide {
  using C = Cfg<N>;
  static constexpr int H = HOP / 2;            // packed points per hop
  static constexpr bool ok = (HOP % 2 == 0) && (H % C::T == 0) && (H / C::T >= 1) && (2 * (H / C::T) <= 32);
  static constexpr int D = ok ? H / C::T : 1;  // slots per hop
};

// edge[i] (i < D): table weights (times the folded scale) of the slots [32-2D, 32-D),
// i.e. of the hop that has just left the weight-1 tail.  sc = folded scale 1/(2N).
template <int N, int HOP>
MX_HD void slide_edge(int t, const float *wtab, float inv_sc, cpx (&edge)[Slide<N, HOP>::D]) {
  using S = Slide<N, HOP>;
#pragma unroll
  for (int i = 0; i < S::D; ++i) {
    const int p = 2 * (t + S::C::T * (32 - 2 * S::D + i));
    // the slot already carries sc; the table carries it too: take it out once (exact, power of two)
    edge[i] = mk(wtab[p] * inv_sc, wtab[p + 1] * inv_sc);
  }
}

// newest hop of the frame that ends at sample pointer xe (one past the frame's last sample)
template <int N, int HOP>
MX_HD void slide_fetch(int t, const float *xe, cpx (&nx)[Slide<N, HOP>::D]) {
  using S = Slide<N, HOP>;
#pragma unroll
  for (int i = 0; i < S::D; ++i) nx[i] = *reinterpret_cast<const cpx *>(xe - HOP + 2 * (t + S::C::T * i));
}

template <int N, int HOP>
MX_HD void slide_step(cpx (&Y)[32], const cpx (&nx)[Slide<N, HOP>::D], const cpx (&edge)[Slide<N, HOP>::D],
                      float g, float sc) {
#pragma clang fp contract(off)  // keep each windowed point a rounded product, whatever consumes it
  using S = Slide<N, HOP>;
#pragma unroll
  for (int e = 0; e < 32 - 2 * S::D; ++e) Y[e] = mk(Y[e + S::D].x * g, Y[e + S::D].y * g);
#pragma unroll
  for (int i = 0; i < S::D; ++i) {
    const cpx o = Y[32 - S::D + i];  // weight-1 slot (scaled by sc): becomes raw * edge
    Y[32 - 2 * S::D + i] = mk(o.x * edge[i].x, o.y * edge[i].y);
  }
#pragma unroll
  for (int i = 0; i < S::D; ++i) Y[32 - S::D + i] = mk(nx[i].x * sc, nx[i].y * sc);
}

// LDS addressing.  Every access below is "per-thread base + compile-time offset"
// (or base ^ constant for the T1 store), so a frame needs a dozen address registers
// instead of one per access; the closed forms are the swizzles above evaluated
// symbolically (tests/test_emu.py checks them against swz1/swz2 for every thread).
template <int N>
MX_HD void store_t1(int t, const cpx (&v)[32], cpx *lds) {
  using C = Cfg<N>;
  // swz1((t + T*b)*R1 + r) = (((t*R1) ^ (t & 15)) ^ r) + b*T*R1
  const int B = (t * C::R1) ^ (t & 15);
#pragma unroll
  for (int r = 0; r < C::R1; ++r) {
    cpx *p = lds + (B ^ r);
#pragma unroll
    for (int b = 0; b < C::NB1; ++b) p[b * C::T * C::R1] = v[b * C::R1 + r];
  }
}

template <int N>
MX_HD void load_t1(int t, cpx (&v)[32], const cpx *lds) {
  using C = Cfg<N>;
  // swz1(j + r*S) = swz1(j) + r*S: S = M/R2 only touches bits above the swizzle's source field
  constexpr int S = C::M / C::R2;
  static_assert((C::R1 == 8 && S % 128 == 0) || (C::R1 == 32 && S % 512 == 0), "T1 read is base+offset");
#pragma unroll
  for (int b = 0; b < C::NB2; ++b) {
    const cpx *p = lds + swz1<N>(t + C::T * b);
#pragma unroll
    for (int r = 0; r < C::R2; ++r) v[b * C::R2 + r] = p[r * S];
  }
}

// ---- pass 2 ----------------------------------------------------------------
// tw2[(r-1)*R1 + k] = exp(-2*pi*i*r*k/(R1*R2)), r = 1..R2-1, k = 0..R1-1
template <int N>
MX_HD void pass2(int t, cpx (&v)[32], const cpx *tw2) {
  using C = Cfg<N>;
#pragma unroll
  for (int b = 0; b < C::NB2; ++b) {
    const int j = t + C::T * b;
    const int k = j & (C::R1 - 1);
    cpx in[C::R2], out[C::R2];
    in[0] = v[b * C::R2];
#pragma unroll
    for (int r = 1; r < C::R2; ++r) {
#ifdef MX_ABL_NOTW
      in[r] = cmul(v[b * C::R2 + r], mk(0.5f + r, 0.25f * k));
#else
      in[r] = cmul(v[b * C::R2 + r], tw2[(r - 1) * C::R1 + k]);
#endif
    }
    Dft<C::R2>::run(in, out);
#pragma unroll
    for (int r = 0; r < C::R2; ++r) v[b * C::R2 + r] = out[r];
  }
}

template <int N>
MX_HD void store_t2(int t, const cpx (&v)[32], cpx *lds) {
  using C = Cfg<N>;
#pragma unroll
  for (int b = 0; b < C::NB2; ++b) {
    const int j = t + C::T * b;
    const int k = j & (C::R1 - 1);
    const int base = (j - k) * C::R2 + k;  // (j / R1) * R1 * R2 + k
    if constexpr (C::R1 == 8) {
      // swz2 flips bit 3 (= r & 1 here) by bit 7 (= (j >> 3) & 1): even r go to +8f, odd r to -8f
      const int f8 = ((j >> 3) & 1) << 3;
      cpx *pe = lds + base + f8, *po = lds + base - f8;
#pragma unroll
      for (int r = 0; r < C::R2; ++r) ((r & 1) ? po : pe)[r * C::R1] = v[b * C::R2 + r];
    } else {
      cpx *p = lds + base;
#pragma unroll
      for (int r = 0; r < C::R2; ++r) p[r * C::R1] = v[b * C::R2 + r];
    }
  }
}

// Butterfly indices of pass 3: P = k0p(t), Q = k0q(t); {P,Q} = {t, NS3-t},
// thread 0 takes the two self-paired ones {0, NS3/2}.
template <int N>
MX_HD int k0p(int t) { return t; }
template <int N>
MX_HD int k0q(int t) { return t ? Cfg<N>::NS3 - t : Cfg<N>::NS3 / 2; }

template <int N>
MX_HD void load_t2(int t, cpx (&v)[32], const cpx *lds) {
  using C = Cfg<N>;
  const int p = k0p<N>(t), q = k0q<N>(t);
  if constexpr (C::R1 == 8) {
    // NS3 = 128: bit 7 of (k0 + 128 r) is r & 1 (k0 < 128), so odd r read from k0 ^ 8
    const cpx *pe = lds + p, *po = lds + (p ^ 8), *qe = lds + q, *qo = lds + (q ^ 8);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[r] = ((r & 1) ? po : pe)[C::NS3 * r];
      v[16 + r] = ((r & 1) ? qo : qe)[C::NS3 * r];
    }
  } else {
    const cpx *pp = lds + p, *qq = lds + q;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[r] = pp[C::NS3 * r];
      v[16 + r] = qq[C::NS3 * r];
    }
  }
}

// ---- pass 3 ----------------------------------------------------------------
// tw3[(r-1)*NS3 + k0] = exp(-2*pi*i*r*k0/M), r = 1..15, k0 = 0..NS3-1.
// Butterfly Q sits at k0 = NS3 - t, and exp(-2*pi*i*r*(NS3-t)/M) = W16^r * conj(tw3[r][t]):
// the conjugate costs nothing inside the complex multiply and the W16^r factor is a
// one-bin rotation of the 16-point DFT's output (sum_r x_r W16^r W16^(rq) = X[q+1]).  So one
// table read serves both butterflies; thread 0 (P = 0, Q = NS3/2) reads column NS3/2 for Q
// and uses 1 for P.
// On return v[r] = Z[k0p + NS3*r] and v[16 + ((r+1)&15)]... is handled by q_index():
// Q's natural element r lives in v[16 + ((r + 1) & 15)].
MX_HD constexpr int q_index(int r) { return 16 + ((r + 1) & 15); }

template <int N>
MX_HD void pass3(int t, cpx (&v)[32], const cpx *tw3) {
  using C = Cfg<N>;
  const int col = t ? t : C::NS3 / 2;
  const bool t0 = (t == 0);
  cpx inp[16], inq[16], out[16];
  inp[0] = v[0];
  inq[0] = v[16];
#pragma unroll
  for (int r = 1; r < 16; ++r) {
#ifdef MX_ABL_NOTW
    const cpx w = mk(0.5f + r, 0.25f * col);
#else
    const cpx w = tw3[(r - 1) * C::NS3 + col];
#endif
    inp[r] = cmul(v[r], csel(t0, mk(1.0f, 0.0f), w));
    inq[r] = cmul(v[16 + r], cconj(w));
  }
  Dft<16>::run(inp, out);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = out[r];
  Dft<16>::run(inq, out);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[16 + r] = out[r];
}

// ---- real-FFT split + magnitude ---------------------------------------------
// After pass 3: v[r] = Z[k0p + NS3*r], v[16+r] = Z[k0q + NS3*r] (natural order).
// Slot s pairs A = Z[k_s] with B = conj(Z[M-k_s]):
//   X[k]   = ((A+B) - i*w_k*(A-B)) / 2,  X[M-k] = conj(((A+B) + i*w_k*(A-B)) / 2),
//   w_k = exp(-2*pi*i*k/N).
// Output: mg[2s] = |X[k_s]|/N at bin kb[2s] = k_s; mg[2s+1] = |X[M-k_s]|/N at
// bin kb[2s+1] = M-k_s  — exactly (float)(sqrt(re^2+im^2)/N) of spec.cpp:62-64,
// computed in fp32.  Thread 0's slot 8 second output is bin M/2 (bin M, the
// Nyquist bin, is not emitted by the reference).
// ub_lo / ub_hi: i*exp(-2*pi*i*t/N) for t > 0; thread 0: i*exp(-2*pi*i/64), -1.
// u[s] = i*w_k for the slot's bin (per-thread constants, see post_twiddles()).
template <int S>
struct PostSlot {
  template <int N>
  static MX_HD void run(bool t0, const cpx (&v)[32], const cpx (&u)[16], float (&mg)[32]) {
    cpx A, B;
    if constexpr (S < 8) {
      A = csel(t0, v[q_index(S)], v[S]);
      B = v[q_index(15 - S)];
    } else {
      A = csel(t0, v[S - 8], v[S]);
      B = csel(t0, v[(24 - S) & 15], v[q_index(15 - S)]);
    }
    B = cconj(B);
    const cpx Sm = cadd(A, B);
    const cpx Dm = csub(A, B);
    const cpx D = cmul(u[S], Dm);
    const cpx lo = csub(Sm, D), hi = cadd(Sm, D);
    mg[2 * S] = fast_sqrt(lo.x * lo.x + lo.y * lo.y);
    mg[2 * S + 1] = fast_sqrt(hi.x * hi.x + hi.y * hi.y);
    if constexpr (S == 8) {  // thread 0: bin M/2 instead of the Nyquist bin; |X[M/2]| = |Z[M/2]|
      const cpx z = v[8];
      const float m = fast_sqrt(z.x * z.x + z.y * z.y) * 2.0f;
      mg[2 * S + 1] = t0 ? m : mg[2 * S + 1];
    }
    if constexpr (S + 1 < 16) PostSlot<S + 1>::template run<N>(t0, v, u, mg);
  }
};

template <int N>
MX_HD void post(int t, const cpx (&v)[32], const cpx (&u)[16], float (&mg)[32]) {
  PostSlot<0>::template run<N>(t == 0, v, u, mg);
}

// Bin of output slot o (= 2s or 2s+1) of thread t:
//   even o: k_s = (s < 8 ? lo : hi) + NS3*s;  odd o: M - k_s  (thread 0, s = 8: M/2)
// with lo = hi = t for t > 0 and lo = NS3/2, hi = -8*NS3 for thread 0.
template <int N>
MX_HD void out_bases(int t, int &lo, int &hi) {
  using C = Cfg<N>;
  lo = t ? t : C::NS3 / 2;
  hi = t ? t : -8 * C::NS3;
}
template <int N>
MX_HD int out_bin(int t, int o) {
  using C = Cfg<N>;
  int lo, hi;
  out_bases<N>(t, lo, hi);
  const int s = o >> 1;
  int k = (s < 8 ? lo : hi) + C::NS3 * s;
  if (o & 1) {
    k = C::M - k;
    if (k == C::M) k = C::M / 2;
  }
  return k;
}
// bit o set iff out_bin(t, o) lies in [kmin, kmax]
template <int N>
MX_HD uint32_t band_mask(int t, int kmin, int kmax) {
  uint32_t m = 0;
#pragma unroll
  for (int o = 0; o < 32; ++o) {
    const int k = out_bin<N>(t, o);
    m |= (k >= kmin && k <= kmax) ? (1u << o) : 0u;
  }
  return m;
}

// Post-split twiddles of thread t: u[s] = i*exp(-2*pi*i*k_s/N) for the slot's bin k_s.
// ubase[t] = i*exp(-2*pi*i*t/N) comes from the table; k_s = t + NS3*s adds exp(-2*pi*i*s/32).
// Thread 0: s < 8 -> k = NS3/2 + NS3*s (base i*exp(-2*pi*i/64)); s >= 8 -> k = NS3*(s-8) (base -1).
template <int S>
struct PostTw {
  static MX_HD void run(cpx lo, cpx hi, cpx (&u)[16]) {
    u[S] = mulw64<2 * S>(S < 8 ? lo : hi);
    if constexpr (S + 1 < 16) PostTw<S + 1>::run(lo, hi, u);
  }
};
template <int N>
MX_HD void post_twiddles(int t, const cpx *ubase, cpx (&u)[16]) {
  cpx lo, hi;
  if (t) {
    lo = ubase[t];
    hi = lo;
  } else {
    lo = mk(kSin64[1], kCos64[1]);  // i*exp(-2*pi*i/64) = sin + i*cos
    hi = mk(-1.0f, 0.0f);
  }
  PostTw<0>::run(lo, hi, u);
}

}  // namespace mx
