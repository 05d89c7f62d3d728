// stft_kernel_impl.h — the STFT workgroup kernel template (see stft_kernels.hip for
// the design notes).  Kept in a header so tools/stft_variants.hip can instantiate
// tuning variants of exactly the shipped code.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "stft_core.h"
#include "stft_tables.h"

namespace mx {

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned long long o = __shfl_xor(k, m, 64);
    k = o > k ? o : k;
  }
  return k;
}

// WPE: waves per SIMD the register allocator must leave room for (amdgpu_waves_per_eu);
// NOHOIST: re-materialise the table pointers every frame so the (frame-invariant) twiddle
// and window loads are not hoisted out of the frame loop into hundreds of registers.
template <int N, int MODE, int HOP, int WPE, bool NOHOIST, bool XCDMAP = true>
__global__ __launch_bounds__(Cfg<N>::T) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void stft_kernel(const StftArgs a0) {
  const StftArgs &a = a0;
  using C = Cfg<N>;
  constexpr int NW = C::T / 64;  // wavefronts per frame
  // one LDS object: the M-point image, then NW 8-byte reduction slots
  __shared__ __attribute__((aligned(16))) float2 lds[C::M + (NW > 1 ? NW : 0)];

  const int t_ = threadIdx.x;
  cpx u[16];  // post-split twiddles: 32 registers that replace 16 complex multiplies per frame
  post_twiddles<N>(t_, a.ubase, u);
  const uint32_t bmask_ = band_mask<N>(t_, a.kmin, a.kmax);
  // output slots some lane of this wavefront needs for the pitch pick (wave-uniform)
  uint32_t umask = 0;
#pragma unroll
  for (int o = 0; o < 32; ++o) umask |= (__ballot((bmask_ >> o) & 1u) != 0ull) ? (1u << o) : 0u;
  umask = __builtin_amdgcn_readfirstlane(umask);
  constexpr bool kSlide = (MODE == kBulkAligned) && (HOP > 0) && Slide<N, (HOP > 0 ? HOP : 2)>::ok;
  constexpr int SD = Slide<N, (HOP > 0 ? HOP : 2)>::D;
  constexpr float kSc = 0.5f / (float)N;

  // XCD-aware block -> frame-range map: the dispatcher places block b on XCD b % 8 and each
  // XCD has a private L2, so hand every XCD one contiguous eighth of the frame range: the
  // 15/16 overlap between neighbouring frame blocks is then an L2 hit instead of a second
  // fetch over the fabric.  (Bijective for any grid size; a different placement only costs speed.)
  unsigned lb = blockIdx.x;
  if constexpr (XCDMAP) {
    const unsigned nb = gridDim.x, xcd = lb & 7u, q = nb >> 3, r = nb & 7u;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lb >> 3);
  }
  const int64_t f0 = (int64_t)lb * a.frames_per_block;
  const int64_t f1 = (f0 + a.frames_per_block < a.count) ? f0 + a.frames_per_block : a.count;

  cpx Y[32];        // the windowed frame (sliding mode: carried from frame to frame)
  cpx edge[SD], nx[SD];
  if constexpr (kSlide) {
    const int64_t e0 = (a.first_frame + f0 + 1) * (int64_t)HOP;
    if (f0 < f1) load_frame<N, 1, true>(t_, Y, a.audio + MX_AUDIO_PAD + (e0 - N), a.wtab);
    slide_edge<N, HOP>(t_, a.wtab, 2.0f * (float)N, edge);
  }

  for (int64_t f = f0; f < f1; ++f) {
    // Everything below that depends only on the thread index is frame-invariant; left alone,
    // LICM hoists ~150 addresses, masks and table values out of the frame loop and the kernel
    // spills.  Re-materialising the thread index (and a zero table offset) per frame keeps the
    // invariants as a handful of cheap VALU ops inside the loop instead.
    int t = t_;
    unsigned bmask = bmask_;
    int zoff = 0;
    if constexpr (NOHOIST) {
      asm volatile("" : "+v"(t), "+v"(bmask), "+s"(zoff));
    }
    const float2 *tw2 = a.tw2 + zoff, *tw3 = a.tw3 + zoff;
    int out_lo, out_hi;
    out_bases<N>(t, out_lo, out_hi);

    if constexpr (kSlide) {
      // prefetch the next frame's newest hop (1 KiB per wavefront) under this frame's math
      if (f + 1 < f1) slide_fetch<N, HOP>(t, a.audio + MX_AUDIO_PAD + (a.first_frame + f + 2) * (int64_t)HOP, nx);
    } else {
      const float *x;
      const float *w;
      if constexpr (MODE == kRanges) {
        const int s = a.ranges[2 * f], e = a.ranges[2 * f + 1];
        const bool outside = (e <= 0) || ((int64_t)e - N >= a.n);  // spec.cpp:50-54: all zeros
        // the leading pad holds MX_AUDIO_PAD >= N zeros: an all-zero frame
        x = outside ? a.audio : a.audio + MX_AUDIO_PAD + ((int64_t)e - N);
        int64_t d0 = (int64_t)N - ((int64_t)e - (int64_t)s);
        d0 = d0 < (int64_t)(N - 1 - kWOff) ? (int64_t)(N - 1 - kWOff) : d0;
        d0 = d0 > (int64_t)(kWDmax + kWTail) ? (int64_t)(kWDmax + kWTail) : d0;
        w = a.wext + kWOff + d0;
      } else {
        const int64_t e = (a.first_frame + f + 1) * (int64_t)a.hop;  // end of frame h: (h+1)*hop
        x = a.audio + MX_AUDIO_PAD + (e - N);
        w = a.wtab + zoff;
      }
      load_frame<N, (MODE == kRanges ? -1 : 1), (MODE == kBulkAligned)>(t, Y, x, w);
    }

    cpx v[32];
    pass1<N>(Y, v);
    if constexpr (kSlide) {
      if (f + 1 < f1) slide_step<N, HOP>(Y, nx, edge, a.decay, kSc);
    }
    store_t1<N>(t, v, lds);
    __syncthreads();
    load_t1<N>(t, v, lds);
    __syncthreads();
    pass2<N>(t, v, tw2);
    store_t2<N>(t, v, lds);
    __syncthreads();
    load_t2<N>(t, v, lds);
    __syncthreads();  // image free for the next frame's T1
    pass3<N>(t, v, tw3);
    float mg[32];
    post<N>(t, v, u, mg);

    // ---- outputs ----
    // bins: even o -> (s<8 ? lo : hi) + NS3*s, odd o -> M - that (thread 0, s = 8: M/2)
    unsigned long long best = 0ull;
    if (a.mags) {
      // Transpose the magnitudes through the LDS image (idle between load_t2 and the next frame's
      // T1): each lane scatters its 32 bins as dwords (consecutive lanes -> consecutive bins, so
      // conflict-free), then every lane owns 4 consecutive bins and the row leaves as 8
      // global_store_dwordx4 per lane, 1 KiB contiguous per wavefront instruction, instead of 32
      // dword stores with one stray element each (thread 0's self-paired bins).
      float *lf = reinterpret_cast<float *>(lds);
      float *plo = lf + out_lo, *phi = lf + out_hi;
      float *mlo = lf + (C::M - out_lo), *mhi = lf + (C::M - out_hi);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        (s < 8 ? plo : phi)[C::NS3 * s] = mg[2 * s];
        if (s == 8) (t == 0 ? lf + C::M / 2 : mhi - C::NS3 * 8)[0] = mg[2 * s + 1];
        else (s < 8 ? mlo : mhi)[-C::NS3 * s] = mg[2 * s + 1];
      }
      __syncthreads();
      using f32x4 = float __attribute__((ext_vector_type(4)));
      const f32x4 *l4 = reinterpret_cast<const f32x4 *>(lds) + t;
      f32x4 *row4 = reinterpret_cast<f32x4 *>(a.mags + (size_t)f * (size_t)(N / 2)) + t;
      f32x4 q[C::M / 4 / C::T];
#pragma unroll
      for (int i = 0; i < C::M / 4 / C::T; ++i) q[i] = l4[C::T * i];
#pragma unroll
      for (int i = 0; i < C::M / 4 / C::T; ++i) __builtin_nontemporal_store(q[i], &row4[C::T * i]);
      __syncthreads();  // image free again
    }
    if (a.pitch) {
      // key = (magnitude bits << 32) | (0x7fffffff - bin): non-negative floats order like their
      // bit patterns, so max(key) = largest magnitude, lowest bin on ties; out-of-band -> 0
      const unsigned klo = 0x7fffffffu - (unsigned)out_lo, khi = 0x7fffffffu - (unsigned)out_hi;
      const unsigned nlo = 0x7fffffffu - (unsigned)(C::M - out_lo), nhi = 0x7fffffffu - (unsigned)(C::M - out_hi);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (!((umask >> (2 * s)) & 3u)) continue;  // wave-uniform: no lane has these two bins in band
        const unsigned b0 = (s < 8 ? klo : khi) - (unsigned)(C::NS3 * s);
        unsigned b1 = (s < 8 ? nlo : nhi) + (unsigned)(C::NS3 * s);
        if (s == 8) b1 = (t == 0) ? 0x7fffffffu - (unsigned)(C::M / 2) : b1;
        const unsigned v0 = (unsigned)((int)(bmask << (31 - 2 * s)) >> 31);      // all-ones iff bit 2s
        const unsigned v1 = (unsigned)((int)(bmask << (31 - (2 * s + 1))) >> 31);  // all-ones iff bit 2s+1
        const unsigned long long k0 = ((unsigned long long)(__float_as_uint(mg[2 * s]) & v0) << 32) | (b0 & v0);
        const unsigned long long k1 = ((unsigned long long)(__float_as_uint(mg[2 * s + 1]) & v1) << 32) | (b1 & v1);
        best = k0 > best ? k0 : best;
        best = k1 > best ? k1 : best;
      }
    }
    if (a.pitch) {
      best = wave_max_u64(best);
      if constexpr (NW > 1) {
        unsigned long long *red = reinterpret_cast<unsigned long long *>(lds + C::M);
        if ((t & 63) == 0) red[t >> 6] = best;
        __syncthreads();
        if (t == 0) {
#pragma unroll
          for (int i = 1; i < NW; ++i) best = red[i] > best ? red[i] : best;
        }
        // red[] is rewritten only after the next frame's barriers
      }
      if (t == 0) {
        mx_pitch p;
        p.bin = 0x7fffffff - (int)(unsigned)(best & 0xffffffffull);
        p.mag = __uint_as_float((unsigned)(best >> 32));
        a.pitch[f] = p;
      }
    }
  }
}


}  // namespace mx
