// stft_core.h — per-thread building blocks of the LDS-resident real-input FFT
// behind Spec::internalGetSpec (reference spec.cpp:44-66).
//
// The same text compiles for gfx950 (hipcc) and for the host (g++), so that
// tests/emu can run one workgroup thread-by-thread on the CPU and check the
// index maps, swizzles and twiddles against the oracle before a GPU is
// involved.  Nothing here is a CPU fallback: the product only instantiates
// these templates inside __global__ kernels (stft_kernels.hip).
//
// Scheme (N real samples, M = N/2 packed complex points z[m] = x[2m] + i*x[2m+1],
// E points per thread, T = M/E threads per frame = one workgroup per hop):
//   pass 1  radix R1, on the windowed samples (register image Y, see Slide)
//   -- transposition T1 through LDS (in place, XOR-swizzled) --
//   pass 2  radix R2 with twiddles exp(-2*pi*i*r*k/(R1*R2))
//   -- transposition T2 through LDS --
//   pass 3  radix R3 = E/2 with twiddles exp(-2*pi*i*r*k0/M); thread t owns the
//           butterflies k0 = t and NS3 - t, i.e. both members of every
//           (k, M-k) pair the real-FFT split needs, so the split, the
//           magnitude and the pitch pick never leave registers.
// Plans (Plan<N, E>):
//   N = 4096,  E = 16: R = 16,16,8   T = 128  two wavefronts per frame
//       (a one-wavefront plan, E = 32 / R = 8,16,16, was measured again under the package power limit in round 3 —
//        2.00 against 1.87 ms per hour, profiles/variants_r03_plan4096E.log — and removed)
//   N = 16384, E = 32: R = 32,16,16  T = 256
//   N = 32768, E = 32: R = 32,32,16  T = 512  (the reference's SpectrSize)
#pragma once
#include <stdint.h>
#include <type_traits>

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
// All fused multiply-adds are written out (fma_) and the kernels are built with -ffp-contract=off, so
// every instantiation of these templates — and the host emulation — performs the same roundings.
MX_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
MX_HD cpx cadd(cpx a, cpx b) { return mk(a.x + b.x, a.y + b.y); }
MX_HD cpx csub(cpx a, cpx b) { return mk(a.x - b.x, a.y - b.y); }
MX_HD cpx cmul(cpx a, cpx b) { return mk(fma_(a.x, b.x, -(a.y * b.y)), fma_(a.x, b.y, a.y * b.x)); }
MX_HD float cnorm2(cpx a) { return fma_(a.x, a.x, a.y * a.y); }
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
MX_HD constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x >> 1); }

}  // namespace mx
#include "pk_math.h"
namespace mx {

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
    return mk(fma_(a.x, c, a.y * s), fma_(a.y, c, -(a.x * s)));
  }
}

// ---- in-register DFT of size R (natural order in, natural order out) --------
// Radix-2 decimation in time with compile-time twiddles, in packed arithmetic (pk_math.h: a complex value
// per instruction).  A butterfly (E, O, W) -> (E + W*O, E - W*O) is three instructions:
//   t = E + O*c (both components), out0 = t -+ i*s*O (the cross terms), out1 = 2*E - out0;
// the trivial twiddles (1, -i) are an add and a subtract with the swap/negate folded into the operands.
template <int K>
MX_HD void bfly_w64(cpx E, cpx O, cpx &out0, cpx &out1) {  // W = exp(-2*pi*i*K/64)
  constexpr int k = ((K % 64) + 64) % 64;
  if constexpr (k == 0) {
    pk_bfly_1(E, O, out0, out1);
  } else if constexpr (k == 16) {  // W = -i
    pk_bfly_mi(E, O, out0, out1);
  } else if constexpr (k == 32) {
    pk_bfly_1(E, O, out1, out0);
  } else if constexpr (k == 48) {
    pk_bfly_mi(E, O, out1, out0);
  } else {
    pk_bfly_cs(E, O, mk(kCos64[k], kSin64[k]), out0, out1);  // W = cos - i*sin: a wave-uniform constant pair
  }
}

template <int R, int Q>
struct Combine {
  static MX_HD void run(const cpx *E, const cpx *O, cpx *out) {
    bfly_w64<Q * 64 / R>(E[Q], O[Q], out[Q], out[Q + R / 2]);
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
    pk_bfly_1(in[0], in[1], out[0], out[1]);
  }
};
template <>
struct Dft<1> {
  static MX_HD void run(const cpx *in, cpx *out) { out[0] = in[0]; }
};

// DFT of inputs that still carry external (run-time) twiddles: x[r] = v[r]*w[r], w[0] == 1 implied
// (w is indexed like v; w[0] is never read).  The multiplications are folded into the leaf
// butterflies of the recursion:  a = v0*w0 (two instructions, none for the leaf that holds x[0]),
// out0 = a + v1*w1 (two), out1 = 2*a - out0 (one).  CONJ: use conj(w[r]).
// S = stride of this sub-transform's elements in the original arrays, O0 = its first original index
template <int R, int S, int O0, bool CONJ>
struct DftTw {
  static MX_HD void run(const cpx *v, const cpx *w, cpx *out) {
    if constexpr (R == 2) {
      constexpr int i0 = O0, i1 = O0 + S;
      if constexpr (i0 == 0) pk_leaf0_tw<CONJ>(v[0], v[i1], w[i1], out[0], out[1]);
      else pk_leaf_tw<CONJ>(v[i0], w[i0], v[i1], w[i1], out[0], out[1]);
    } else {
      cpx E[R / 2], O[R / 2];
      DftTw<R / 2, 2 * S, O0, CONJ>::run(v, w, E);
      DftTw<R / 2, 2 * S, O0 + S, CONJ>::run(v, w, O);
      Combine<R, 0>::run(E, O, out);
    }
  }
};

// ---- geometry -------------------------------------------------------------
template <int N_, int E_>
struct Plan {
  static constexpr int N = N_;
  static constexpr int M = N / 2;   // packed complex points
  static constexpr int E = E_;      // points per thread
  static constexpr int T = M / E;   // threads per frame
  static constexpr int R3 = E / 2;  // last pass: one butterfly pair per thread
  static constexpr int R1 = (N == 4096) ? 16 : 32;
  static constexpr int R2 = M / (R1 * R3);
  static constexpr int NS3 = R1 * R2;  // finished sub-transform size entering pass 3 (= M/R3)
  static constexpr int NB1 = E / R1;   // butterflies per thread in pass 1
  static constexpr int NB2 = E / R2;
  static constexpr int TW2 = (R2 - 1) * R1;   // entries of the pass-2 twiddle table
  static constexpr int TW3 = (R3 - 1) * NS3;  // entries of the pass-3 twiddle table
  static constexpr int L1 = ilog2(R1);
  static_assert(N == 4096 || N == 16384 || N == 32768, "supported FFT sizes");
  static_assert((N == 4096) ? E == 16 : E == 32, "supported points per thread");
  static_assert(R1 * R2 * R3 == M && R2 <= E && R1 <= E && T % 64 == 0, "radix plan must cover M");
};

// Layout of the LDS image during the first transposition (T1).  Pass-1 outputs are written with lane stride R1
// (thread t owns the R1 consecutive points t*R1 .. t*R1 + R1 - 1) and read back contiguously; T2 is written in runs
// of R1 and read contiguously and needs nothing.  Two layouts keep the strided T1 writes off each other's banks:
//  * R1 = 32 (N = 16384 / 32768) — PADDED: one complex point of padding after every 32, logical index i lives at
//    i + (i >> 5).  A lane group of 16 writes dwords 66*t + {0,1}: every bank once; the contiguous reads skip one
//    point per 32 and stay conflict-free; and EVERY address is "per-thread base + compile-time offset".  The XOR
//    swizzle it replaced (round 3) cost one v_xor per stored point, 32 per thread and frame: N = 32768 12.58 ->
//    12.40 ms per hour at 375-sample columns, 4.63 -> 4.54 at hop 1024 (profiles/variants_r03_t1pad.log).  The image
//    grows by M/32 points (132 KiB at N = 32768: fits).
//  * R1 = 16 (N = 4096) — XOR: i ^ ((i >> 4) & 15), every aligned block of 32 points a permutation of itself, image
//    of exactly M points.  Padding measured 5 % SLOWER here (1.79 -> 1.89 ms): one point per 32 leaves two lanes of a
//    write group on one bank pair, one per 16 costs the sixth workgroup per CU its LDS.
template <class P>
MX_HD constexpr bool t1_padded() { return P::R1 == 32; }
template <class P>
MX_HD constexpr int t1_size() { return t1_padded<P>() ? P::M + P::M / 32 : P::M; }
template <class P>
MX_HD int t1_index(int i) {
  if constexpr (t1_padded<P>()) return i + (i >> 5);
  else return i ^ ((i >> P::L1) & 15);
}

// ---- the windowed frame -----------------------------------------------------
// Y[e], e = b + NB1*r, is the packed complex point c = t + T*e (samples 2c, 2c+1 of the
// frame), i.e. input r of pass-1 butterfly j = t + T*b.  The window tables carry the
// output scale 1/(2N) (a power of two, so x*(w*2^-k) == (x*w)*2^-k bit for bit):
// magnitudes come out of the split already scaled and the per-bin multiply disappears.
// x points at the frame's first sample (file index end-N); w at the weight of that sample.
// WSTEP = +1: w[p] (bulk table, forward); WSTEP = -1: w[-p] (the d-indexed table walked
// downwards, ranges mode).  ALIGNED8: x and w are 8-byte aligned (one 64-bit load per pair).
struct alignas(4) f2u {  // 4-byte aligned pair for frames starting at odd samples
  float x, y;
};
// The pair of floats at element index i (a 32-bit, non-negative offset) from base: with a wave-uniform base the load
// takes the SGPR-base + 32-bit-VGPR-offset form — one address add per load instead of a 64-bit sign-extend/shift/add.
template <bool ALIGNED8>
MX_HD cpx ld_pair(const float *base, int i) {
  const char *a = reinterpret_cast<const char *>(base) + (size_t)(4u * (unsigned)i);
  if constexpr (ALIGNED8) {
    return *reinterpret_cast<const cpx *>(a);
  } else {
    const f2u u = *reinterpret_cast<const f2u *>(a);
    return mk(u.x, u.y);
  }
}

// one float at element index i (a 32-bit, non-negative offset) from a wave-uniform base
MX_HD float ld_one(const float *base, int i) {
  return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (size_t)(4u * (unsigned)i));
}

template <class P, int WSTEP, bool ALIGNED8>
MX_HD void load_frame(int t, cpx (&Y)[P::E], const float *x, const float *w) {
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    const int p = 2 * (t + P::T * e);
    cpx xs, ws;
    if constexpr (ALIGNED8 && WSTEP == 1) {
      xs = ld_pair<true>(x, p);
      ws = ld_pair<true>(w, p);
    } else {
      xs = ld_pair<false>(x, p);
      if constexpr (WSTEP == 1) {
        ws = ld_pair<false>(w, p);
      } else {
        const f2u wu = *reinterpret_cast<const f2u *>(w - p - 1);
        ws = mk(wu.y, wu.x);
      }
    }
    Y[e] = pk_mul(xs, ws);  // rounded binary32 products, as spec.cpp:58 (times the folded 2^-k)
  }
}

// Bulk direct modes: the weights of a thread's points form a geometric sequence
// (w(p + 2T) = w(p) * exp(2.5e-4 * 2T) until the flat top d <= 0, where w = sc), so only the
// samples are loaded: wb[p], p < 2T, is the UNCLAMPED weight sc*exp(-2.5e-4f*(N-hop-p)) of the
// thread's first pair (stft_tables.h make_wtab, second section) and the rest are one multiply and
// one min each.  This halves the L1 traffic of a frame (the weight table was as large as the
// frame).  The weights then differ from expf(-2.5e-4f*d) by <= 2 ulp — like the sliding kernel's
// decay chain, far inside the magnitude tolerance; the ranges mode (the reference's per-column
// calls) keeps the exact table.
template <int T>
MX_HD constexpr float win_grow(int e) {
  return T == 64 ? kWinGrow64[e] : T == 128 ? kWinGrow128[e] : T == 256 ? kWinGrow256[e] : kWinGrow512[e];
}
// the geometric weights of slot e from the thread's seed pair a0, clamped at the flat top.  The clamp only ever binds
// inside the flat top (the newest `hop` samples: one sample before it the weight is already sc*(1 - 2.5e-4), a
// thousand ulps below sc), i.e. in the last ceil(hop / 2T) slots: CLAMP = false leaves it out for the slots in front
// of those — the same values with two instructions less per slot.  Positive floats order like their bit patterns, so
// the clamp itself is an integer minimum (no NaN canonicalisation in front of it).
template <class P, bool CLAMP = true>
MX_HD cpx geo_weight(cpx a0, int e) {
  constexpr float sc = 0.5f / (float)P::N;
  const float g = e == 0 ? 1.0f : win_grow<P::T>(e);
  // two literal-operand multiplies rather than one packed multiply by an SGPR pair: the 31 growth factors would
  // otherwise occupy 62 scalar registers for the whole frame loop and push the twiddle constants into VGPR lanes
  // (53 SGPR spills, ~100 v_readlane / v_writelane per frame in the 32-points-per-thread kernels)
  const cpx w = e == 0 ? a0 : mk(a0.x * g, a0.y * g);
  if constexpr (!CLAMP) {
    return w;
  } else {
    const unsigned sb = __builtin_bit_cast(unsigned, sc), xb = __builtin_bit_cast(unsigned, w.x), yb = __builtin_bit_cast(unsigned, w.y);
    return mk(__builtin_bit_cast(float, xb < sb ? xb : sb), __builtin_bit_cast(float, yb < sb ? yb : sb));
  }
}
// number of trailing slots whose weights can reach the flat top
template <class P>
MX_HD int geo_clamped_slots(int hop) { return (hop + 2 * P::T - 1) / (2 * P::T); }
constexpr int kGeoTail = 2;  // slots the fast path still clamps (hop <= 2 * 2T)
template <class P, bool ALIGNED8>
MX_HD void load_frame_geo(int t, cpx (&Y)[P::E], const float *x, const float *wb, int hop) {
  static_assert(P::T == 64 || P::T == 128 || P::T == 256 || P::T == 512, "growth table per T");
  static_assert(P::E <= 32, "growth table length");
  const cpx a0 = ld_pair<true>(wb, 2 * t);
  if (geo_clamped_slots<P>(hop) <= kGeoTail) {  // wave-uniform
#pragma unroll
    for (int e = 0; e < P::E; ++e) {
      const cpx xs = ld_pair<ALIGNED8>(x, 2 * (t + P::T * e));
      Y[e] = pk_mul(xs, e < P::E - kGeoTail ? geo_weight<P, false>(a0, e) : geo_weight<P, true>(a0, e));
    }
  } else {
#pragma unroll
    for (int e = 0; e < P::E; ++e) Y[e] = pk_mul(ld_pair<ALIGNED8>(x, 2 * (t + P::T * e)), geo_weight<P, true>(a0, e));
  }
}

// The same weights applied to samples fetched earlier (load_raw, the prefetching schedule).
template <class P>
MX_HD void apply_window_geo(int t, cpx (&Y)[P::E], const cpx (&xr)[P::E], const float *wb, int hop) {
  const cpx a0 = ld_pair<true>(wb, 2 * t);
  if (geo_clamped_slots<P>(hop) <= kGeoTail) {  // wave-uniform
#pragma unroll
    for (int e = 0; e < P::E; ++e)
      Y[e] = pk_mul(xr[e], e < P::E - kGeoTail ? geo_weight<P, false>(a0, e) : geo_weight<P, true>(a0, e));
  } else {
#pragma unroll
    for (int e = 0; e < P::E; ++e) Y[e] = pk_mul(xr[e], geo_weight<P, true>(a0, e));
  }
}

// Direct modes, split in two so that the raw samples of the NEXT frame can be in flight while the
// current one is transformed: load_raw issues the 64-bit sample loads, apply_window multiplies by
// the weights (same rounded binary32 product as load_frame) once the frame is needed.
template <class P, bool ALIGNED8>
MX_HD void load_raw(int t, cpx (&xr)[P::E], const float *x) {
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    xr[e] = ld_pair<ALIGNED8>(x, 2 * (t + P::T * e));
  }
}
template <class P, int WSTEP, bool ALIGNED8>
MX_HD void apply_window(int t, cpx (&Y)[P::E], const cpx (&xr)[P::E], const float *w) {
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    const int p = 2 * (t + P::T * e);
    cpx ws;
    if constexpr (WSTEP == 1) {
      ws = ld_pair<ALIGNED8>(w, p);
    } else {
      const f2u wu = *reinterpret_cast<const f2u *>(w - p - 1);
      ws = mk(wu.y, wu.x);
    }
    Y[e] = pk_mul(xr[e], ws);
  }
}

// ---- pass 1 ------------------------------------------------------------------
template <class P>
MX_HD void pass1(const cpx (&Y)[P::E], cpx (&v)[P::E]) {
#pragma unroll
  for (int b = 0; b < P::NB1; ++b) {
    cpx in[P::R1], out[P::R1];
#pragma unroll
    for (int r = 0; r < P::R1; ++r) in[r] = Y[b + P::NB1 * r];
    Dft<P::R1>::run(in, out);
#pragma unroll
    for (int r = 0; r < P::R1; ++r) v[b * P::R1 + r] = out[r];
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
template <class P, int HOP>
struct Slide {
  static constexpr int H = HOP / 2;  // packed points per hop
  static constexpr bool ok = (HOP > 0) && (HOP % 2 == 0) && (H % P::T == 0) && (2 * (H / P::T) <= P::E);
  static constexpr int D = ok ? H / P::T : 1;  // slots per hop
};

// edge[i] (i < D): table weights of the slots [E-2D, E-D), i.e. of the hop that has just left
// the weight-1 tail.  The slot already carries the folded scale and so does the table: take it
// out once (exact, power of two).
template <class P, int HOP>
MX_HD void slide_edge(int t, const float *wtab, float inv_sc, cpx (&edge)[Slide<P, HOP>::D]) {
  using S = Slide<P, HOP>;
#pragma unroll
  for (int i = 0; i < S::D; ++i) {
    const int p = 2 * (t + P::T * (P::E - 2 * S::D + i));
    edge[i] = mk(wtab[p] * inv_sc, wtab[p + 1] * inv_sc);
  }
}

// newest hop of the frame that ends at sample pointer xe (one past the frame's last sample)
template <class P, int HOP>
MX_HD void slide_fetch(int t, const float *xe, cpx (&nx)[Slide<P, HOP>::D]) {
  using S = Slide<P, HOP>;
#pragma unroll
  for (int i = 0; i < S::D; ++i) nx[i] = *reinterpret_cast<const cpx *>(xe - HOP + 2 * (t + P::T * i));
}

template <class P, int HOP>
MX_HD void slide_step(cpx (&Y)[P::E], const cpx (&nx)[Slide<P, HOP>::D], const cpx (&edge)[Slide<P, HOP>::D],
                      float g, float sc) {
  // every windowed point stays a rounded binary32 product, whatever consumes it (one packed multiply per point)
  using S = Slide<P, HOP>;
  const cpx gg = mk(g, g), ss = mk(sc, sc);
#pragma unroll
  for (int e = 0; e < P::E - 2 * S::D; ++e) Y[e] = pk_mul_xs(Y[e + S::D], gg);
#pragma unroll
  for (int i = 0; i < S::D; ++i) Y[P::E - 2 * S::D + i] = pk_mul(Y[P::E - S::D + i], edge[i]);  // weight-1 slot: raw * edge
#pragma unroll
  for (int i = 0; i < S::D; ++i) Y[P::E - S::D + i] = pk_mul_xs(nx[i], ss);
}


// ---- circular sliding window (uniform hops that are not a multiple of 2T samples) ----------------------
// |DFT| does not change under a circular shift of the (real) input, so the register image need not start at the
// frame's first sample.  Here it starts at the last multiple of 2T samples (one slot) at or before it: frame position
// q lives at circular position (o + q) mod N, o = (frame start) mod 2T — slot 0 holds the frame's newest o samples in
// front of its oldest 2T - o ones.  Stepping to the next frame then is
//   * the whole image ages by one hop (one multiply by exp(-2.5e-4*hop) per point, as in slide_step), rotated down by
//     s = (o + hop) div 2T slots on the way (s in {0, 1, 2}: the multiply reads the register s slots up, no moves);
//   * the newest 2*hop samples are overwritten with audio * exact table weight: the newest hop enter with the flat
//     top, the hop before them leaves it, each sample by its own factor.  They always sit in the LAST slots and in
//     slot 0 — compile-time registers — and are 2*hop samples from L2 instead of the N a direct load re-reads.
// Only products older than two hops carry a multiply chain, restarted from a direct load at every workgroup's first
// frame.  The slot-aligned sliding kernels are the special case o = 0, hop = 2T*D.
template <class P>
struct Circ {
  static constexpr int W = 2 * P::T;                 // samples per slot
  static constexpr int CS = (P::T == 128) ? 4 : (P::T == 256) ? 3 : 2;  // slots the newest 2*hop samples can touch (incl. slot 0)
  static MX_HD bool ok(int hop) { return hop >= 1 && hop <= (CS - 1) * P::T && hop <= 2 * W; }
  static constexpr int slot(int k) { return (P::E - (CS - 1) + k) & (P::E - 1); }  // k = 0..CS-1; the last one is slot 0
};
template <class P>
struct CircGeo {
  int o;    // (frame start) mod 2T: circular position of frame position 0
  int s;    // slots the image rotates down coming from the previous frame
  int cr;   // circular position of the first of the newest 2*hop samples
  int len;  // 2 * hop
};
template <class P>
MX_HD CircGeo<P> circ_geo(long long frame_end, int hop) {
  constexpr int W = Circ<P>::W;
  const long long start = frame_end - P::N;
  CircGeo<P> g;
  g.o = (int)(start & (long long)(W - 1));
  g.s = ((int)((start - hop) & (long long)(W - 1)) + hop) / W;
  g.len = 2 * hop;
  g.cr = (g.o - g.len) & (P::N - 1);
  return g;
}
// a workgroup's first frame: direct load.  xa = the frame's first sample, w = the exact weight table
template <class P>
MX_HD void circ_load_first(int t, cpx (&Y)[P::E], const float *xa, const float *w, int o) {
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    const int c2 = 2 * (t + P::T * e);
    const int q0 = (c2 - o) & (P::N - 1), q1 = (c2 + 1 - o) & (P::N - 1);  // positions inside the frame
    Y[e] = mk(xa[q0] * w[q0], xa[q1] * w[q1]);  // rounded binary32 products, as everywhere
  }
}
// the newest 2*hop samples of the frame `g` describes and their weights, for this thread's points in the candidate
// slots: xs = audio at (frame end - 2*hop), wt = wtab + (N - 2*hop).  Lanes outside fetch a clamped (valid, unused) position.
template <class P>
MX_HD void circ_fetch(int t, const float *xs, const float *wt, const CircGeo<P> &g, float (&px)[2 * Circ<P>::CS],
                      float (&pw)[2 * Circ<P>::CS]) {
#pragma unroll
  for (int k = 0; k < Circ<P>::CS; ++k) {
    const int c2 = 2 * (t + P::T * Circ<P>::slot(k));
    int d0 = (c2 - g.cr) & (P::N - 1), d1 = (c2 + 1 - g.cr) & (P::N - 1);
    d0 = d0 < g.len ? d0 : g.len - 1;
    d1 = d1 < g.len ? d1 : g.len - 1;
    // (wave-uniform bases + non-negative 32-bit lane offsets: SGPR-base loads, no 64-bit address arithmetic per load)
    px[2 * k] = ld_one(xs, d0);
    px[2 * k + 1] = ld_one(xs, d1);
    pw[2 * k] = ld_one(wt, d0);
    pw[2 * k + 1] = ld_one(wt, d1);
  }
}
template <class P, int S>
MX_HD void circ_age(cpx (&Y)[P::E], cpx gg) {
  cpx head[S > 0 ? S : 1];
#pragma unroll
  for (int i = 0; i < S; ++i) head[i] = Y[i];
#pragma unroll
  for (int e = 0; e < P::E - S; ++e) Y[e] = pk_mul_xs(Y[e + S], gg);
#pragma unroll
  for (int i = 0; i < S; ++i) Y[P::E - S + i] = pk_mul_xs(head[i], gg);
}
// the step into the frame `g` describes
template <class P>
MX_HD void circ_step(int t, cpx (&Y)[P::E], float decay, const CircGeo<P> &g, const float (&px)[2 * Circ<P>::CS],
                     const float (&pw)[2 * Circ<P>::CS]) {
  const cpx gg = mk(decay, decay);
  constexpr int SMAX = (Circ<P>::W - 1 + (Circ<P>::CS - 1) * P::T) / Circ<P>::W;  // largest rotation ok() admits
  static_assert(SMAX <= 2, "rotations by 0, 1 or 2 slots");
  if (g.s == 0) circ_age<P, 0>(Y, gg);  // wave-uniform
  else if (SMAX == 1 || g.s == 1) circ_age<P, 1>(Y, gg);
  else circ_age<P, 2>(Y, gg);
#pragma unroll
  for (int k = 0; k < Circ<P>::CS; ++k) {
    const int e = Circ<P>::slot(k);
    const int c2 = 2 * (t + P::T * e);
    const int d0 = (c2 - g.cr) & (P::N - 1), d1 = (c2 + 1 - g.cr) & (P::N - 1);
    const float n0 = px[2 * k] * pw[2 * k], n1 = px[2 * k + 1] * pw[2 * k + 1];
    Y[e] = mk(d0 < g.len ? n0 : Y[e].x, d1 < g.len ? n1 : Y[e].y);
  }
}


// ---- unmerged LDS reads ---------------------------------------------------------
// The compiler fuses neighbouring 8-byte LDS reads into ds_read2(st64)_b64, which the LDS
// serves at half the rate of two plain ds_read_b64 (MI355X_MICROARCH.md, LDS table: 8 cycles
// per 1 KiB wave-instruction against 2 cycles per 512 B).  The transposition reads are a
// third of the kernel's LDS time, so on the device they are issued as plain ds_read_b64 by
// hand; a single s_waitcnt closes the batch, and the values are threaded through that
// statement so nothing that consumes them can be scheduled above it.
#if defined(__HIP_DEVICE_COMPILE__)
#define MX_LDS_ASM 1  // the device pass; the host pass (tests/emu) takes the plain C++ forms of the same reads
typedef float mx_f2v __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ mx_f2v lds_rd64(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 163840, "LDS is 160 KiB");
  mx_f2v r;
  if constexpr (OFF < 65536) {  // the ds offset field is 16 bits
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  } else {
    const uint32_t hi = addr + (uint32_t)(OFF & ~65535);  // one add per base and 64 KiB window, shared by that window's reads
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(hi), "n"(OFF & 65535) : "memory");
  }
  return r;
}
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
template <int N>
__device__ __forceinline__ void lds_wait(mx_f2v (&q)[N]) {
  static_assert(N % 8 == 0, "tied in groups of 8");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; i += 8)
    asm volatile("" : "+v"(q[i]), "+v"(q[i + 1]), "+v"(q[i + 2]), "+v"(q[i + 3]), "+v"(q[i + 4]),
                 "+v"(q[i + 5]), "+v"(q[i + 6]), "+v"(q[i + 7]));
}
#endif
// f(integral_constant<int, I>) for I = I0 .. N-1: a loop whose index is a constant expression in the body
template <int I, int N, class F>
MX_HD void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ---- LDS transpositions -------------------------------------------------------
// Every access below is "per-thread base + compile-time offset" (or base ^ constant for
// the T1 store), so a frame needs a dozen address registers instead of one per access;
// the closed forms are the swizzles above evaluated symbolically (tests/emu checks them
// against t1_index/swz2 for every thread).
template <class P>
MX_HD void store_t1(int t, const cpx (&v)[P::E], cpx *lds) {
  if constexpr (t1_padded<P>()) {
    // t1_index((t + T*b)*R1 + r) = (t*R1 + (t*R1 >> 5)) + b*(T*R1 + T*R1/32) + r:  r < R1 <= 32 never carries into the
    // pad term (t*R1 has its low log2(R1) bits clear) and T*R1 is a multiple of 32
    static_assert(P::R1 <= 32 && (P::T * P::R1) % 32 == 0, "T1 padding is one point per 32");
    cpx *const p = lds + (t * P::R1 + ((t * P::R1) >> 5));
#pragma unroll
    for (int b = 0; b < P::NB1; ++b) {
#pragma unroll
      for (int r = 0; r < P::R1; ++r) p[b * (P::T * P::R1 + P::T * P::R1 / 32) + r] = v[b * P::R1 + r];
    }
  } else {
    // t1_index((t + T*b)*R1 + r) = (((t*R1) ^ (t & 15)) ^ r) + b*T*R1   (T is a multiple of 16)
    // (in bytes, so that each of the R1 scatter addresses is ONE xor with a constant — the shift by 3 is done once)
    const unsigned B8 = (unsigned)((t * P::R1) ^ (t & 15)) << 3;
    char *const base = reinterpret_cast<char *>(lds);
#pragma unroll
    for (int r = 0; r < P::R1; ++r) {
      cpx *p = reinterpret_cast<cpx *>(base + (B8 ^ ((unsigned)r << 3)));
#pragma unroll
      for (int b = 0; b < P::NB1; ++b) p[b * P::T * P::R1] = v[b * P::R1 + r];
    }
  }
}

// The T1 read of thread t: point (t + T*b) + r*S, S = M/R2, is at base[(r & 1) ? 1 : 0] + b*TP + r*SP.
//   padded: t1_index(j + r*S) = (j + (j >> 5)) + r*(S + S/32), one base (T and S are multiples of 32);
//   XOR:    t1_index(j + r*S) = (t1_index(j) ^ c_r) + r*S with c_r = (r * (S >> L1)) & 15: the stride only reaches the
//           swizzle's source field through its top bit (c_r in {0, 8}) or not at all — an alternate base for odd r.
template <class P>
struct T1Read {
  static constexpr int S = P::M / P::R2;
  static constexpr int SP = t1_padded<P>() ? S + S / 32 : S;
  static constexpr int TP = t1_padded<P>() ? P::T + P::T / 32 : P::T;
  static constexpr int step = t1_padded<P>() ? 0 : ((S >> P::L1) & 15);
  static_assert(S % 32 == 0 && P::T % 32 == 0 && (step == 0 || step == 8), "T1 read is base(+alt base) + offset");
  static_assert(t1_padded<P>() || P::NB2 == 1, "the XOR layout's bases are written for one pass-2 butterfly per thread");
};

template <class P>
MX_HD void load_t1(int t, cpx (&v)[P::E], const cpx *lds) {
  using R = T1Read<P>;
  const int s1 = t1_index<P>(t);
#ifdef MX_LDS_ASM
  mx_f2v q[P::E];
  const uint32_t ae = lds_addr(lds + s1), ao = lds_addr(lds + (s1 ^ R::step));
  static_for<0, P::NB2>([&](auto bb) {
    constexpr int b = decltype(bb)::value;
    static_for<0, P::R2>([&](auto rr) {
      constexpr int r = decltype(rr)::value;
      q[b * P::R2 + r] = lds_rd64<(b * R::TP + r * R::SP) * 8>((r & 1) ? ao : ae);
    });
  });
  lds_wait(q);
#pragma unroll
  for (int i = 0; i < P::E; ++i) v[i] = mk(q[i].x, q[i].y);
#else
  const cpx *pe = lds + s1, *po = lds + (s1 ^ R::step);
#pragma unroll
  for (int b = 0; b < P::NB2; ++b) {
#pragma unroll
    for (int r = 0; r < P::R2; ++r) v[b * P::R2 + r] = ((r & 1) ? po : pe)[b * R::TP + r * R::SP];
  }
#endif
}

// T1 read plus this thread's pass-2 twiddles (table in LDS) as one batch of plain
// ds_read_b64 behind a single wait.  One butterfly per thread (NB2 == 1) only: the
// twiddles then cost 2*(R2-1) registers for the length of pass 2.
template <class P>
MX_HD void load_t1_tw2(int t, cpx (&v)[P::E], const cpx *lds, const cpx *ltw2, cpx (&w)[P::R2 - 1]) {
  static_assert(P::NB2 == 1 && P::E == P::R2 && P::E % 8 == 0, "one pass-2 butterfly per thread");
  using R = T1Read<P>;
  const int s1 = t1_index<P>(t);
#ifdef MX_LDS_ASM
  const uint32_t ae = lds_addr(lds + s1), ao = lds_addr(lds + (s1 ^ R::step));
  const uint32_t aw = lds_addr(ltw2 + (t & (P::R1 - 1)));
  mx_f2v q[2 * P::E];
  static_for<0, P::R2>([&](auto rr) {
    constexpr int r = decltype(rr)::value;
    q[r] = lds_rd64<r * R::SP * 8>((r & 1) ? ao : ae);
  });
  static_for<1, P::R2>([&](auto rr) {
    constexpr int r = decltype(rr)::value;
    q[P::E + r] = lds_rd64<(r - 1) * P::R1 * 8>(aw);
  });
  q[P::E] = mx_f2v{0.0f, 0.0f};  // pads the tie list to a multiple of 8 (a constant: never a copy of a read still in flight)
  lds_wait(q);
#pragma unroll
  for (int i = 0; i < P::E; ++i) v[i] = mk(q[i].x, q[i].y);
#pragma unroll
  for (int r = 1; r < P::R2; ++r) w[r - 1] = mk(q[P::E + r].x, q[P::E + r].y);
#else
  const cpx *pe = lds + s1, *po = lds + (s1 ^ R::step), *pw = ltw2 + (t & (P::R1 - 1));
#pragma unroll
  for (int r = 0; r < P::R2; ++r) v[r] = ((r & 1) ? po : pe)[r * R::SP];
#pragma unroll
  for (int r = 1; r < P::R2; ++r) w[r - 1] = pw[(r - 1) * P::R1];
#endif
}

// ---- pass 2 ----------------------------------------------------------------
// tw2[(r-1)*R1 + k] = exp(-2*pi*i*r*k/(R1*R2)), r = 1..R2-1, k = 0..R1-1
template <class P>
MX_HD void pass2(int t, cpx (&v)[P::E], const cpx *tw2) {
#pragma unroll
  for (int b = 0; b < P::NB2; ++b) {
    const int j = t + P::T * b;
    const int k = j & (P::R1 - 1);
    cpx in[P::R2], w[P::R2], out[P::R2];
    w[0] = mk(1.0f, 0.0f);
#pragma unroll
    for (int r = 0; r < P::R2; ++r) in[r] = v[b * P::R2 + r];
#pragma unroll
    for (int r = 1; r < P::R2; ++r) {
      w[r] = tw2[(r - 1) * P::R1 + k];
    }
    DftTw<P::R2, 1, 0, false>::run(in, w, out);
#pragma unroll
    for (int r = 0; r < P::R2; ++r) v[b * P::R2 + r] = out[r];
  }
}

template <class P>
MX_HD void pass2_reg(cpx (&v)[P::E], const cpx (&w)[P::R2 - 1]) {  // NB2 == 1: the twiddles load_t1_tw2 brought
  static_assert(P::NB2 == 1, "one pass-2 butterfly per thread");
  cpx in[P::R2], ww[P::R2], out[P::R2];
  ww[0] = mk(1.0f, 0.0f);
#pragma unroll
  for (int r = 0; r < P::R2; ++r) in[r] = v[r];
#pragma unroll
  for (int r = 1; r < P::R2; ++r) ww[r] = w[r - 1];
  DftTw<P::R2, 1, 0, false>::run(in, ww, out);
#pragma unroll
  for (int r = 0; r < P::R2; ++r) v[r] = out[r];
}

template <class P>
MX_HD void store_t2(int t, const cpx (&v)[P::E], cpx *lds) {
#pragma unroll
  for (int b = 0; b < P::NB2; ++b) {
    const int j = t + P::T * b;
    const int k = j & (P::R1 - 1);
    const int base = (j - k) * P::R2 + k;  // (j / R1) * R1 * R2 + k
    cpx *p = lds + base;
#pragma unroll
    for (int r = 0; r < P::R2; ++r) p[r * P::R1] = v[b * P::R2 + r];
  }
}

// Butterfly indices of pass 3: P-butterfly = k0p(t), Q-butterfly = k0q(t); {P,Q} = {t, NS3-t},
// thread 0 takes the two self-paired ones {0, NS3/2}.
template <class P>
MX_HD int k0p(int t) { return t; }
template <class P>
MX_HD int k0q(int t) { return t ? P::NS3 - t : P::NS3 / 2; }

template <class P>
MX_HD void load_t2(int t, cpx (&v)[P::E], const cpx *lds) {
  const int p = k0p<P>(t), q = k0q<P>(t);
#ifdef MX_LDS_ASM
  {
    const uint32_t pa = lds_addr(lds + p), qa = lds_addr(lds + q);
    mx_f2v w[P::E];
    static_for<0, P::R3>([&](auto rr) {
      constexpr int r = decltype(rr)::value;
      w[r] = lds_rd64<P::NS3 * r * 8>(pa);
      w[P::R3 + r] = lds_rd64<P::NS3 * r * 8>(qa);
    });
    lds_wait(w);
#pragma unroll
    for (int i = 0; i < P::E; ++i) v[i] = mk(w[i].x, w[i].y);
  }
#else
  const cpx *pp = lds + p, *qq = lds + q;
#pragma unroll
  for (int r = 0; r < P::R3; ++r) {
    v[r] = pp[P::NS3 * r];
    v[P::R3 + r] = qq[P::NS3 * r];
  }
#endif
}

// ---- pass 3 ----------------------------------------------------------------
// tw3[(r-1)*NS3 + k0] = exp(-2*pi*i*r*k0/M), r = 1..R3-1, k0 = 0..NS3-1.
// Butterfly Q sits at k0 = NS3 - t, and exp(-2*pi*i*r*(NS3-t)/M) = W_R3^r * conj(tw3[r][t]):
// the conjugate costs nothing inside the complex multiply and the W_R3^r factor is a one-bin
// rotation of the R3-point DFT's output (sum_r x_r W^r W^(rq) = X[q+1]).  So one table read
// serves both butterflies; thread 0 (P = 0, Q = NS3/2) reads column NS3/2 for Q and uses 1 for P.
// Q's natural element r therefore lives in v[q_index(r)].
template <class P>
MX_HD constexpr int q_index(int r) { return P::R3 + ((r + 1) & (P::R3 - 1)); }

// (MAY0 = false below: the caller knows thread 0 is not in this wavefront, all thread-0 selects fold away)
template <class P>
MX_HD void fetch_tw3(int t, const cpx *tw3, cpx (&w)[P::R3 - 1]) {
  const int col = t ? t : P::NS3 / 2;
#pragma unroll
  for (int r = 1; r < P::R3; ++r) w[r - 1] = tw3[(r - 1) * P::NS3 + col];
}
// Two-level pass-3 twiddles (R3 = 16): w_r = gamma^r, gamma = exp(-2*pi*i*col/M) this thread's base.  Six powers stay
// in registers (gamma^1..3 and gamma^4, 8, 12), the other nine are one complex product each per frame: 12 registers
// instead of 30 — what lets the plans with 32 points per thread keep their pass-3 twiddles out of L2 next to a
// sliding frame image (15 loads per thread and frame otherwise).
template <class P>
MX_HD void fetch_tw3_bases(int t, const cpx *tw3, cpx (&wb)[6]) {
  static_assert(P::R3 == 16, "two-level pass-3 twiddles are written for radix 16");
  const int col = t ? t : P::NS3 / 2;
  constexpr int rr[6] = {1, 2, 3, 4, 8, 12};
#pragma unroll
  for (int i = 0; i < 6; ++i) wb[i] = tw3[(rr[i] - 1) * P::NS3 + col];
}
template <class P, bool MAY0>
MX_HD void pass3_reg(int t, cpx (&v)[P::E], const cpx (&w)[P::R3 - 1]);
template <class P, bool MAY0 = true>
MX_HD void pass3_bases(int t, cpx (&v)[P::E], const cpx (&wb)[6]) {
  cpx w[P::R3 - 1];
#pragma unroll
  for (int r = 1; r < P::R3; ++r) {
    const int hi = r >> 2, lo = r & 3;
    if (hi == 0) w[r - 1] = wb[lo - 1];
    else if (lo == 0) w[r - 1] = wb[2 + hi];
    else w[r - 1] = pk_cmul2(wb[2 + hi], wb[lo - 1]);
  }
  pass3_reg<P, MAY0>(t, v, w);
}
template <class P, bool MAY0 = true>
MX_HD void pass3_reg(int t, cpx (&v)[P::E], const cpx (&w)[P::R3 - 1]) {
  constexpr int R = P::R3;
  const bool t0 = MAY0 && (t == 0);
  cpx inp[R], inq[R], wp[R], wq[R], out[R];
  wp[0] = wq[0] = mk(1.0f, 0.0f);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    inp[r] = v[r];
    inq[r] = v[R + r];
  }
#pragma unroll
  for (int r = 1; r < R; ++r) {
    wp[r] = csel(t0, mk(1.0f, 0.0f), w[r - 1]);
    wq[r] = w[r - 1];
  }
  DftTw<R, 1, 0, false>::run(inp, wp, out);
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = out[r];
  DftTw<R, 1, 0, true>::run(inq, wq, out);
#pragma unroll
  for (int r = 0; r < R; ++r) v[R + r] = out[r];
}

// ---- real-FFT split + magnitude ---------------------------------------------
// After pass 3: v[r] = Z[k0p + NS3*r], v[q_index(r)] = Z[k0q + NS3*r] (natural order).
// Slot s pairs A = Z[k_s] with B = conj(Z[M-k_s]):
//   X[k]   = ((A+B) - i*w_k*(A-B)) / 2,  X[M-k] = conj(((A+B) + i*w_k*(A-B)) / 2),
//   w_k = exp(-2*pi*i*k/N).
// Output: mg[2s] = |X[k_s]|/N at bin k_s; mg[2s+1] = |X[M-k_s]|/N at bin M-k_s — the
// (float)(sqrt(re^2+im^2)/N) of spec.cpp:62-64 in fp32; the 1/(2N) is already in the data.
// Thread 0 owns the two self-paired butterflies: its slots s < R3/2 pair Q[s] with Q[R3-1-s],
// its slots s >= R3/2 pair P[s-R3/2] with P[(3*R3/2 - s) mod R3]; its slot R3/2 yields bin 0
// and — instead of the Nyquist bin, which the reference does not emit — bin M/2 = |Z[M/2]|.
// u[s] = i*w_k for the slot's bin (per-thread constants, see post_twiddles()).
template <class P, int S>
struct PostSlot {
  static MX_HD void run(bool t0, const cpx (&v)[P::E], const cpx (&u)[P::R3], float (&mg)[P::E]) {
    constexpr int R = P::R3, H = R / 2;
    cpx A, B;
    if constexpr (S < H) {
      A = csel(t0, v[q_index<P>(S)], v[S]);
      B = v[q_index<P>(R - 1 - S)];
    } else {
      A = csel(t0, v[S - H], v[S]);
      B = csel(t0, v[(3 * H - S) & (R - 1)], v[q_index<P>(R - 1 - S)]);
    }
    // lo = Sm - D, hi = Sm + D with Sm = A + conj(B), D = u*(A - conj(B)); the two squared magnitudes are
    // formed side by side: px = (lo.x, hi.x), py = (lo.y, hi.y), n2 = px*px + py*py
    const cpx n2 = pk_split_norm2(A, B, u[S]);
    mg[2 * S] = fast_sqrt(n2.x);
    mg[2 * S + 1] = fast_sqrt(n2.y);
    if constexpr (S == H) {  // thread 0: bin M/2 instead of the Nyquist bin; |X[M/2]| = |Z[M/2]|
      const cpx z = v[H];
      const float m = fast_sqrt(cnorm2(z)) * 2.0f;
      mg[2 * S + 1] = t0 ? m : mg[2 * S + 1];
    }
    if constexpr (S + 1 < R) PostSlot<P, S + 1>::run(t0, v, u, mg);
  }
};

template <class P, bool MAY0 = true>
MX_HD void post(int t, const cpx (&v)[P::E], const cpx (&u)[P::R3], float (&mg)[P::E]) {
  PostSlot<P, 0>::run(MAY0 && t == 0, v, u, mg);
}

// The same with the R3 post-split twiddles rebuilt from their base every frame (u[s] = base * exp(-2*pi*i*s/(2*R3)):
// a rotation by a compile-time angle, two packed instructions) instead of being held in 2*R3 registers — the
// registers go to the sliding / circular frame image of the 32-points-per-thread plans (stft_kernel_impl.h, R3 == 16).
template <class P, int S>
struct PostFly {
  static MX_HD void run(cpx lo, cpx hi, cpx (&u)[P::R3]) {
    constexpr int k = ((32 / P::R3) * S) % 64;
    const cpx b = S < P::R3 / 2 ? lo : hi;
    if constexpr (k == 0) u[S] = b;
    else if constexpr (k == 16) u[S] = mk(b.y, -b.x);
    else u[S] = pk_rot_cs(b, mk(kCos64[k], kSin64[k]));
    if constexpr (S + 1 < P::R3) PostFly<P, S + 1>::run(lo, hi, u);
  }
};
template <class P>
MX_HD void post_bases(int t, const cpx *ubase, cpx &lo, cpx &hi) {
  if (t) {
    lo = ubase[t];
    hi = lo;
  } else {
    constexpr int k = 16 / P::R3;
    lo = mk(kSin64[k], kCos64[k]);
    hi = mk(-1.0f, 0.0f);
  }
}
template <class P, bool MAY0 = true>
MX_HD void post_fly(int t, const cpx (&v)[P::E], cpx lo, cpx hi, float (&mg)[P::E]) {
  cpx u[P::R3];
  PostFly<P, 0>::run(lo, MAY0 ? hi : lo, u);
  PostSlot<P, 0>::run(MAY0 && t == 0, v, u, mg);
}

// The same split with the complex bins kept (phase-vocoder analysis, pv_kernels.hip):
// X[2s] = X[k_s]/N, X[2s+1] = X[M-k_s]/N, same slot -> bin map as post() (out_bin), thread 0's
// slot R3/2 again yields bin 0 and bin M/2 (X[M/2] = conj(Z[M/2])) instead of the Nyquist bin.
template <class P, int S>
struct PostSlotC {
  static MX_HD void run(bool t0, const cpx (&v)[P::E], const cpx (&u)[P::R3], cpx (&X)[P::E]) {
    constexpr int R = P::R3, H = R / 2;
    cpx A, B;
    if constexpr (S < H) {
      A = csel(t0, v[q_index<P>(S)], v[S]);
      B = v[q_index<P>(R - 1 - S)];
    } else {
      A = csel(t0, v[S - H], v[S]);
      B = csel(t0, v[(3 * H - S) & (R - 1)], v[q_index<P>(R - 1 - S)]);
    }
    const cpx Sm = pk_add_cj(A, B);
    const cpx Dm = pk_sub_cj(A, B);
    const cpx D = pk_cmul<false>(Dm, u[S]);
    X[2 * S] = pk_sub(Sm, D);
    X[2 * S + 1] = cconj(pk_add(Sm, D));
    if constexpr (S == H) {
      const cpx z = v[H];
      X[2 * S + 1] = csel(t0, mk(2.0f * z.x, -2.0f * z.y), X[2 * S + 1]);
    }
    if constexpr (S + 1 < R) PostSlotC<P, S + 1>::run(t0, v, u, X);
  }
};
template <class P, bool MAY0 = true>
MX_HD void post_cplx(int t, const cpx (&v)[P::E], const cpx (&u)[P::R3], cpx (&X)[P::E]) {
  PostSlotC<P, 0>::run(MAY0 && t == 0, v, u, X);
}

// Bin of output slot o (= 2s or 2s+1) of thread t:
//   even o: k_s = (s < R3/2 ? lo : hi) + NS3*s;  odd o: M - k_s  (thread 0, s = R3/2: M/2)
// with lo = hi = t for t > 0 and lo = NS3/2, hi = -(R3/2)*NS3 for thread 0.
template <class P>
MX_HD void out_bases(int t, int &lo, int &hi) {
  lo = t ? t : P::NS3 / 2;
  hi = t ? t : -(P::R3 / 2) * P::NS3;
}
template <class P>
MX_HD int out_bin(int t, int o) {
  int lo, hi;
  out_bases<P>(t, lo, hi);
  const int s = o >> 1;
  int k = (s < P::R3 / 2 ? lo : hi) + P::NS3 * s;
  if (o & 1) {
    k = P::M - k;
    if (k == P::M) k = P::M / 2;
  }
  return k;
}
// bit o set iff out_bin(t, o) lies in [kmin, kmax]
template <class P>
MX_HD uint32_t band_mask(int t, int kmin, int kmax) {
  uint32_t m = 0;
#pragma unroll
  for (int o = 0; o < P::E; ++o) {
    const int k = out_bin<P>(t, o);
    m |= (k >= kmin && k <= kmax) ? (1u << o) : 0u;
  }
  return m;
}

// Post-split twiddles of thread t: u[s] = i*exp(-2*pi*i*k_s/N) for the slot's bin k_s.
// ubase[t] = i*exp(-2*pi*i*t/N) comes from the table; k_s = t + NS3*s adds exp(-2*pi*i*s/(2*R3)).
// Thread 0: s < R3/2 -> k = NS3/2 + NS3*s (base i*exp(-2*pi*i/(4*R3))); s >= R3/2 -> k = NS3*(s-R3/2) (base -1).
template <class P, int S>
struct PostTw {
  static MX_HD void run(cpx lo, cpx hi, cpx (&u)[P::R3]) {
    u[S] = mulw64<(32 / P::R3) * S>(S < P::R3 / 2 ? lo : hi);
    if constexpr (S + 1 < P::R3) PostTw<P, S + 1>::run(lo, hi, u);
  }
};
template <class P>
MX_HD void post_twiddles(int t, const cpx *ubase, cpx (&u)[P::R3]) {
  cpx lo, hi;
  if (t) {
    lo = ubase[t];
    hi = lo;
  } else {
    constexpr int k = 16 / P::R3;   // exp(-2*pi*i/(4*R3)) in 64ths of a turn
    lo = mk(kSin64[k], kCos64[k]);  // i*exp(-i*a) = sin a + i*cos a
    hi = mk(-1.0f, 0.0f);
  }
  PostTw<P, 0>::run(lo, hi, u);
}

}  // namespace mx
