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

template <int N, int WSTEP, bool ALIGNED8>
MX_HD void pass1(int t, cpx (&v)[32], const float *x, const float *w) {
  using C = Cfg<N>;
#pragma unroll
  for (int b = 0; b < C::NB1; ++b) {
    const int j = t + C::T * b;
    cpx in[C::R1], out[C::R1];
#pragma unroll
    for (int r = 0; r < C::R1; ++r) {
      const int p = 2 * (j + r * (C::M / C::R1));
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
      in[r] = mk(x0 * w0, x1 * w1);  // float product, as spec.cpp:58
    }
    Dft<C::R1>::run(in, out);
#pragma unroll
    for (int r = 0; r < C::R1; ++r) v[b * C::R1 + r] = out[r];
  }
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
    for (int r = 1; r < C::R2; ++r) in[r] = cmul(v[b * C::R2 + r], tw2[(r - 1) * C::R1 + k]);
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
// tw3[(r-1)*NS3 + k0] = exp(-2*pi*i*r*k0/M), r = 1..15, k0 = 0..NS3-1
template <int N>
MX_HD void pass3(int t, cpx (&v)[32], const cpx *tw3) {
  using C = Cfg<N>;
  const int kk[2] = {k0p<N>(t), k0q<N>(t)};
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    cpx in[16], out[16];
    in[0] = v[16 * b];
#pragma unroll
    for (int r = 1; r < 16; ++r) in[r] = cmul(v[16 * b + r], tw3[(r - 1) * C::NS3 + kk[b]]);
    Dft<16>::run(in, out);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[16 * b + r] = out[r];
  }
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
template <int S>
struct PostSlot {
  template <int N>
  static MX_HD void run(bool t0, const cpx (&v)[32], cpx ub_lo, cpx ub_hi, float (&mg)[32]) {
    constexpr float scale = 0.5f / static_cast<float>(N);
    cpx A, B;
    if constexpr (S < 8) {
      A = csel(t0, v[16 + S], v[S]);
      B = v[16 + 15 - S];
    } else {
      A = csel(t0, v[S - 8], v[S]);
      B = csel(t0, v[(24 - S) & 15], v[16 + 15 - S]);
    }
    B = cconj(B);
    const cpx Sm = cadd(A, B);
    const cpx Dm = csub(A, B);
    const cpx D = mulw64<2 * S>(cmul(S < 8 ? ub_lo : ub_hi, Dm));
    const cpx lo = csub(Sm, D), hi = cadd(Sm, D);
#if defined(__HIP_DEVICE_COMPILE__)
    mg[2 * S] = __builtin_amdgcn_sqrtf(lo.x * lo.x + lo.y * lo.y) * scale;
    mg[2 * S + 1] = __builtin_amdgcn_sqrtf(hi.x * hi.x + hi.y * hi.y) * scale;
#else
    mg[2 * S] = __builtin_sqrtf(lo.x * lo.x + lo.y * lo.y) * scale;
    mg[2 * S + 1] = __builtin_sqrtf(hi.x * hi.x + hi.y * hi.y) * scale;
#endif
    if constexpr (S == 8) {  // thread 0: bin M/2 instead of the Nyquist bin; |X[M/2]| = |Z[M/2]|
      const cpx z = v[8];
#if defined(__HIP_DEVICE_COMPILE__)
      const float m = __builtin_amdgcn_sqrtf(z.x * z.x + z.y * z.y) * (2.0f * scale);
#else
      const float m = __builtin_sqrtf(z.x * z.x + z.y * z.y) * (2.0f * scale);
#endif
      mg[2 * S + 1] = t0 ? m : mg[2 * S + 1];
    }
    if constexpr (S + 1 < 16) PostSlot<S + 1>::template run<N>(t0, v, ub_lo, ub_hi, mg);
  }
};

template <int N>
MX_HD void post(int t, const cpx (&v)[32], cpx ub_lo, cpx ub_hi, float (&mg)[32]) {
  PostSlot<0>::template run<N>(t == 0, v, ub_lo, ub_hi, mg);
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

// Post-split twiddle bases of thread t (ubase[t] = i*exp(-2*pi*i*t/N) from the table).
template <int N>
MX_HD void post_bases(int t, const cpx *ubase, cpx &ub_lo, cpx &ub_hi) {
  if (t) {
    ub_lo = ubase[t];
    ub_hi = ub_lo;
  } else {
    ub_lo = mk(kSin64[1], kCos64[1]);  // i*exp(-2*pi*i/64) = sin + i*cos
    ub_hi = mk(-1.0f, 0.0f);
  }
}

}  // namespace mx
