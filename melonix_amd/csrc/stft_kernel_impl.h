// stft_kernel_impl.h — the STFT workgroup kernel template (see stft_kernels.hip for
// the design notes).  Kept in a header so tools/stft_variants.hip can instantiate
// tuning variants of exactly the shipped code.
#pragma once
#include <hip/hip_runtime.h>

#include "colormap_core.h"
#include "kernels.h"
#include "stft_core.h"
#include "stft_tables.h"

#define MX_BARRIER() __syncthreads()

namespace mx {

// Wavefront reductions through the DPP crossbar (no LDS round trips, unlike __shfl_xor which lowers
// to ds_bpermute): xor-1, xor-2 (quad_perm), row_half_mirror, row_mirror leave every 16-lane row
// with its result; row_bcast:15 / row_bcast:31 fold the rows; lane 63 holds the wavefront's.
// `old` is the operation's identity, so lanes a control does not reach are unaffected and the
// compiler's DPP-combine pass can fold each move into the v_max/v_min that consumes it.
template <bool MAXOP>
__device__ __forceinline__ unsigned wave_reduce_u32(unsigned k) {
  constexpr unsigned ident = MAXOP ? 0u : 0xffffffffu;
#define MX_DPP_STEP(CTRL, ROWS)                                                                      \
  {                                                                                                  \
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)ident, (int)k, CTRL, ROWS, 0xf, false); \
    k = MAXOP ? (o > k ? o : k) : (o < k ? o : k);                                                   \
  }
  MX_DPP_STEP(0xB1, 0xf)   // quad_perm [1,0,3,2]
  MX_DPP_STEP(0x4E, 0xf)   // quad_perm [2,3,0,1]
  MX_DPP_STEP(0x141, 0xf)  // row_half_mirror
  MX_DPP_STEP(0x140, 0xf)  // row_mirror
  MX_DPP_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
  MX_DPP_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef MX_DPP_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)k, 63);
}
// max over the wavefront of (magnitude bits, then lowest bin): two 32-bit reductions.
// key = (mag bits << 32) | (0x7fffffff - bin); returns the wavefront's best key.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
  const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
  const unsigned mhi = wave_reduce_u32<true>(hi);
  // the low word of the lane(s) that hold the maximum: almost always ONE lane — read it directly; ties (silence,
  // equal magnitudes in two lanes) take the second reduction
  const unsigned long long tie = __ballot(hi == mhi);
  unsigned mlo;
  if (__builtin_popcountll(tie) == 1) mlo = (unsigned)__builtin_amdgcn_readlane((int)lo, __builtin_ctzll(tie));
  else mlo = wave_reduce_u32<true>(hi == mhi ? lo : 0u);
  return ((unsigned long long)mhi << 32) | mlo;
}

// One dword of a magnitude row: *(float *)((char *)base + off + IMM) = x, streaming (NT: nt) or plain write-back, with
// `base` wave-uniform (an SGPR pair), `off` a 32-bit lane offset and IMM the instruction's 13-bit signed immediate.
template <int IMM, bool NT = true>
__device__ __forceinline__ void st_row_nt(const float *base, unsigned off, float x) {
  static_assert(IMM >= -4096 && IMM <= 4095, "global_store immediate offset");
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (NT) asm volatile("global_store_dword %0, %1, %2 offset:%3 nt" : : "v"(off), "v"(x), "s"(base), "n"(IMM) : "memory");
  else asm volatile("global_store_dword %0, %1, %2 offset:%3" : : "v"(off), "v"(x), "s"(base), "n"(IMM) : "memory");
#endif
}

// Template parameters (every combination that is instantiated ships: stft_kernels.hip):
//   MODE / HOP   kBulkAligned with HOP > 0: the sliding window for that hop; HOP = 0: direct loads (or CIRC)
//   WPE          waves per SIMD the register allocator must leave room for (amdgpu_waves_per_eu)
//   DEFER        (the two-wave N = 4096 plan) the magnitude row and the pitch record of frame f leave the workgroup
//                during frame f+1 through an LDS region of their own; the barrier that frees the image sits right
//                after the T2 read, so the next T1 scatter can be issued while pass 1 is still computing
//   PREFETCH     (direct modes) the next frame's raw samples are requested once pass 3 has freed the transform's registers
//   CMAP         RGB8 texel output (the fused colormap of SpecCache::populateTex)
//   DIRECT       (N = 32768) the row leaves straight from the registers, one dword per lane and slot
//   CIRC         the circular sliding window (stft_core.h)
// Twiddle placement follows from the plan: the pass-2 table always lives in LDS (shared by the workgroup's waves); with
// R3 = 8 (N = 4096) a thread's seven pass-3 twiddles and its post-split twiddles stay in registers for the whole
// workgroup; with R3 = 16 (N = 16384 / 32768) six pass-3 base powers stay in registers, the other nine twiddles are one
// packed complex product each per frame and the post-split twiddles are rebuilt from their base per frame (12 + 2
// registers instead of 30 + 32) — nothing comes from L2 per frame.
// (Everything that depends only on the thread index is frame-invariant; left alone, LICM hoists ~150 addresses, masks
// and table values out of the frame loop and the kernel spills.  The thread index is therefore re-materialised per
// frame — an empty asm — which keeps the invariants as a handful of cheap VALU ops inside the loop.)
template <class P, int MODE, int HOP, int WPE, bool DEFER = false, bool PREFETCH = false, bool CMAP = false, bool DIRECT = false,
          bool CIRC = false>
__global__ __launch_bounds__(P::T) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void stft_kernel(const StftArgs a0) {
  const StftArgs &a = a0;
  using C = P;
  constexpr int N = P::N;
  constexpr int NW = C::T / 64;  // wavefronts per frame
  // one LDS object: the image (with the first transposition's padding), NW 8-byte reduction slots (padded to 16 B),
  // — DEFER — a separate M-float region for the magnitude transposition, so that the image can be refilled by the next
  // frame without an extra barrier, and the pass-2 twiddle table
  constexpr int kRed = (NW > 1) ? ((NW + 1) / 2) * 2 : 0;
  constexpr bool kTw3Bases = (P::R3 == 16);
  constexpr int kTw2 = ((C::TW2 + 1) / 2) * 2;
  static_assert(P::R3 == 16 || P::R3 == 8, "twiddle placement per plan");
  static_assert(!DIRECT || (NW > 1 && !DEFER && !CMAP), "direct row stores: multi-wave, non-deferred plans");
  static_assert(!CIRC || (MODE != kRanges && HOP == 0 && !PREFETCH && !CMAP), "circular window: bulk modes");
  constexpr int IMG = t1_size<C>();
  __shared__ __attribute__((aligned(16))) float2 lds[IMG + kRed + (DEFER ? C::M / 2 : 0) + kTw2];
  float *const lout = reinterpret_cast<float *>(DEFER ? lds + IMG + kRed : lds);
  float2 *const ltw2 = lds + IMG + kRed + (DEFER ? C::M / 2 : 0);

  const int t_ = threadIdx.x;
  const bool wave0 = __builtin_amdgcn_readfirstlane(t_) < 64;  // wave-uniform
  cpx u[kTw3Bases ? 1 : P::R3];  // post-split twiddles (R3 = 8: registers that replace R3 complex multiplies per frame)
  cpx ulo;  // (thread 0's second base, (-1, 0), is re-selected per frame in the one wavefront that holds thread 0)
  cpx w3r[kTw3Bases ? 1 : P::R3 - 1], w3base[6];
  if constexpr (kTw3Bases) {
    cpx uhi_unused;
    post_bases<P>(t_, a.ubase, ulo, uhi_unused);
    fetch_tw3_bases<P>(t_, a.tw3, w3base);
  } else {
    post_twiddles<P>(t_, a.ubase, *reinterpret_cast<cpx(*)[P::R3]>(&u));
    fetch_tw3<P>(t_, a.tw3, *reinterpret_cast<cpx(*)[P::R3 - 1]>(&w3r));
  }
  const uint32_t bmask_ = band_mask<P>(t_, a.kmin, a.kmax);
  // output slots some lane of this wavefront needs for the pitch pick (wave-uniform)
  uint32_t umask = 0;
#pragma unroll
  for (int o = 0; o < P::E; ++o) umask |= (__ballot((bmask_ >> o) & 1u) != 0ull) ? (1u << o) : 0u;
  umask = __builtin_amdgcn_readfirstlane(umask);
  constexpr bool kSlide = (MODE == kBulkAligned) && (HOP > 0) && Slide<P, (HOP > 0 ? HOP : 2)>::ok;
  constexpr int SD = Slide<P, (HOP > 0 ? HOP : 2)>::D;
  constexpr float kSc = 0.5f / (float)N;
  for (int i = t_; i < C::TW2; i += C::T) ltw2[i] = a.tw2[i];
  MX_BARRIER();

  // XCD-aware block -> frame-range map: the dispatcher places block b on XCD b % 8 and each
  // XCD has a private L2, so hand every XCD one contiguous eighth of the frame range: the
  // 15/16 overlap between neighbouring frame blocks is then an L2 hit instead of a second
  // fetch over the fabric.  (Bijective for any grid size; a different placement only costs speed.)
  unsigned lb = blockIdx.x;
  {
    const unsigned nb = gridDim.x, xcd = lb & 7u, q = nb >> 3, r = nb & 7u;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lb >> 3);
  }
  const int64_t f0 = (int64_t)lb * a.frames_per_block;
  const int64_t f1 = (f0 + a.frames_per_block < a.count) ? f0 + a.frames_per_block : a.count;

  cpx Y[P::E];      // the windowed frame (sliding mode: carried from frame to frame)
  cpx edge[SD], nx[SD];
  if constexpr (kSlide) {
    const int64_t e0 = (a.first_frame + f0 + 1) * (int64_t)HOP;
    if (f0 < f1) load_frame<P, 1, true>(t_, Y, a.audio + MX_AUDIO_PAD + (e0 - N), a.wtab);
    slide_edge<P, HOP>(t_, a.wtab, 2.0f * (float)N, edge);
  }

  // DEFER: the magnitude row and the pitch record of frame f leave the workgroup during frame f+1 — scatter at the
  // end of f, LDS read + global stores after f+1's T1 barrier — so no barrier and no LDS round trip sits in the
  // output path.
  unsigned long long *const red = reinterpret_cast<unsigned long long *>(lds + IMG);
  auto flush_pitch = [&](int64_t fr, int tt) {  // after a barrier that follows red[] of frame fr
    if (a.pitch && tt == 0) {
      unsigned long long b = red[0];
#pragma unroll
      for (int i = 1; i < NW; ++i) b = red[i] > b ? red[i] : b;
      mx_pitch p;
      p.bin = 0x7fffffff - (int)(unsigned)(b & 0xffffffffull);
      p.mag = __uint_as_float((unsigned)(b >> 32));
      a.pitch[fr] = p;
    }
  };
  // CMAP (ranges mode): the row can also leave as RGB8 texels — SpecCache::populateTex's colormap
  // (spec-cache.cpp:77-96) applied to the four consecutive bins a lane holds, 12 bytes per lane.
  const bool want_rows = a.mags != nullptr || (CMAP && a.rgb != nullptr);
  auto flush_row = [&](int64_t fr, int tt) {  // after a barrier that follows the scatter of frame fr
    if (want_rows) {
      using f32x4 = float __attribute__((ext_vector_type(4)));
      const f32x4 *l4 = reinterpret_cast<const f32x4 *>(lout) + tt;
      f32x4 *row4 = reinterpret_cast<f32x4 *>(a.mags + (size_t)fr * (size_t)(N / 2)) + tt;
      f32x4 q[C::M / 4 / C::T];
#pragma unroll
      for (int i = 0; i < C::M / 4 / C::T; ++i) q[i] = l4[C::T * i];
      if (!CMAP || a.mags) {
#pragma unroll
        for (int i = 0; i < C::M / 4 / C::T; ++i) {
          __builtin_nontemporal_store(q[i], &row4[C::T * i]);
        }
      }
      if constexpr (CMAP) {
        if (a.rgb) {
          uint32_t *trow = reinterpret_cast<uint32_t *>(a.rgb + (size_t)fr * (size_t)(N / 2) * 3) + 3 * tt;
#pragma unroll 1
          for (int i = 0; i < C::M / 4 / C::T; ++i) {
            uint32_t w3[3];
            texel4(q[i].x, q[i].y, q[i].z, q[i].w, a.cmap_k, w3);
            uint32_t *o = trow + 3 * C::T * i;
            o[0] = w3[0];
            o[1] = w3[1];
            o[2] = w3[2];
          }
        }
      }
    }
  };

  // sample / weight pointers of frame f in the direct modes
  auto frame_ptrs = [&](int64_t fr, int zo, const float *&x, const float *&w) {
    if constexpr (MODE == kRanges) {
      const int s = a.ranges[2 * fr], e = a.ranges[2 * fr + 1];
      const bool outside = (e <= 0) || ((int64_t)e - N >= a.n);  // spec.cpp:50-54: all zeros
      // the leading pad holds MX_AUDIO_PAD >= N zeros: an all-zero frame
      x = outside ? a.audio : a.audio + MX_AUDIO_PAD + ((int64_t)e - N);
      int64_t d0 = (int64_t)N - ((int64_t)e - (int64_t)s);
      d0 = d0 < (int64_t)(N - 1 - kWOff) ? (int64_t)(N - 1 - kWOff) : d0;
      d0 = d0 > (int64_t)(kWDmax + kWTail) ? (int64_t)(kWDmax + kWTail) : d0;
      w = a.wext + kWOff + d0;
    } else {
      const int64_t e = (a.first_frame + fr + 1) * (int64_t)a.hop;  // end of frame h: (h+1)*hop
      x = a.audio + MX_AUDIO_PAD + (e - N);
      w = a.wtab + zo;
    }
  };
  float cpx_[2 * Circ<P>::CS], cpw_[2 * Circ<P>::CS];  // CIRC: a frame's newest samples and their weights
  if constexpr (CIRC) {
    if (f0 < f1) {
      const int64_t pe = (a.first_frame + f0 + 1) * (int64_t)a.hop;
      circ_load_first<P>(t_, Y, a.audio + MX_AUDIO_PAD + (pe - N), a.wtab, circ_geo<P>(pe, a.hop).o);
    }
  }
  cpx xr[(PREFETCH && !kSlide) ? P::E : 1];  // raw samples of the next frame, in flight
  if constexpr (PREFETCH && !kSlide) {
    if (f0 < f1) {
      const float *x0;
      const float *w0;
      frame_ptrs(f0, 0, x0, w0);
      load_raw<P, (MODE == kBulkAligned)>(t_, xr, x0);
    }
  }

  for (int64_t f = f0; f < f1; ++f) {
    // Everything below that depends only on the thread index is frame-invariant; left alone,
    // LICM hoists ~150 addresses, masks and table values out of the frame loop and the kernel
    // spills.  Re-materialising the thread index (and a zero table offset) per frame keeps the
    // invariants as a handful of cheap VALU ops inside the loop instead.
    int t = t_;
    unsigned bmask = bmask_;
    int zoff = 0;
    asm volatile("" : "+v"(t), "+v"(bmask), "+s"(zoff));
    int out_lo, out_hi;
    out_bases<P>(t, out_lo, out_hi);

    if constexpr (CIRC) {
      if (f > f0) {
        // the newest 2*hop samples and their weights are requested here, at the top of their own frame, and land under
        // the ageing multiplies: requested a frame ahead (after pass 3) their registers stay live across the output
        // phase and spill — measured 2 % slower at N = 32768, 6 % at N = 16384
        const int64_t pe = (a.first_frame + f + 1) * (int64_t)a.hop;
        const CircGeo<P> geo = circ_geo<P>(pe, a.hop);
        circ_fetch<P>(t, a.audio + MX_AUDIO_PAD + (pe - 2 * (int64_t)a.hop), a.wtab + zoff + (N - 2 * a.hop), geo, cpx_, cpw_);
        circ_step<P>(t, Y, a.decay, geo, cpx_, cpw_);
      }
    } else if constexpr (kSlide) {
      // in-place shift+decay into this frame (Y[e] <- Y[e+D]*g reads ahead of what it writes), then
      // prefetch the next frame's newest hop (1 KiB per wavefront) under this frame's math
      if (f > f0) slide_step<P, HOP>(Y, nx, edge, a.decay, kSc);
      if (f + 1 < f1) slide_fetch<P, HOP>(t, a.audio + MX_AUDIO_PAD + (a.first_frame + f + 2) * (int64_t)HOP, nx);
    } else {
      const float *x;
      const float *w;
      frame_ptrs(f, zoff, x, w);
      if constexpr (PREFETCH) {
        // this frame's samples were requested while the previous frame was being finished (below)
        if constexpr (MODE == kRanges) apply_window<P, -1, false>(t, Y, xr, w);
        else apply_window_geo<P>(t, Y, xr, a.wtab + N + zoff, a.hop);
      } else if constexpr (MODE == kRanges) {
        load_frame<P, -1, false>(t, Y, x, w);  // exact d-indexed weights (per-column calls)
      } else {
        load_frame_geo<P, (MODE == kBulkAligned)>(t, Y, x, a.wtab + N + zoff, a.hop);  // samples only
      }
    }

    cpx v[P::E];
    pass1<P>(Y, v);
    // one pass-2 butterfly per thread (N = 4096, N = 32768): its twiddles ride with the T1 read, one batch, one wait
    constexpr bool kTw2Batch = (P::NB2 == 1);
    store_t1<P>(t, v, lds);
    MX_BARRIER();
    if constexpr (DEFER || DIRECT) {
      if (f > f0) flush_pitch(f - 1, t);
    }
    if constexpr (kTw2Batch) {
      cpx w2[P::R2 - 1];
      load_t1_tw2<P>(t, v, lds, ltw2, w2);
      if constexpr (DEFER) {
        if (f > f0) flush_row(f - 1, t);  // previous frame's row: LDS -> HBM in the shadow of T1
      }
      MX_BARRIER();
      pass2_reg<P>(v, w2);
    } else {
      load_t1<P>(t, v, lds);
      if constexpr (DEFER) {
        if (f > f0) flush_row(f - 1, t);
      }
      MX_BARRIER();
      pass2<P>(t, v, ltw2);
    }
    store_t2<P>(t, v, lds);
    MX_BARRIER();
    load_t2<P>(t, v, lds);
    MX_BARRIER();  // image free (for the magnitude scatter / — DEFER — the next T1 scatter)
    float mg[P::E];
    if (NW == 1 || wave0) {  // wave-uniform: only the first wavefront contains thread 0
      if constexpr (kTw3Bases) {
        pass3_bases<P, true>(t, v, w3base);
        post_fly<P, true>(t, v, ulo, csel(t == 0, mk(-1.0f, 0.0f), ulo), mg);
      } else {
        pass3_reg<P, true>(t, v, *reinterpret_cast<cpx(*)[P::R3 - 1]>(&w3r));
        post<P, true>(t, v, *reinterpret_cast<cpx(*)[P::R3]>(&u), mg);
      }
    } else {
      if constexpr (kTw3Bases) {
        pass3_bases<P, false>(t, v, w3base);
        post_fly<P, false>(t, v, ulo, ulo, mg);
      } else {
        pass3_reg<P, false>(t, v, *reinterpret_cast<cpx(*)[P::R3 - 1]>(&w3r));
        post<P, false>(t, v, *reinterpret_cast<cpx(*)[P::R3]>(&u), mg);
      }
    }

    if constexpr (PREFETCH && !kSlide) {
      // the transform's registers are free again: request the next frame's samples now, so that they
      // arrive under the pitch pick, the magnitude transposition and the row's stores
      if (f + 1 < f1) {
        const float *xn;
        const float *wn;
        frame_ptrs(f + 1, zoff, xn, wn);
        load_raw<P, (MODE == kBulkAligned)>(t, xr, xn);
      }
    }

    // ---- pitch pick: per-thread best, then wavefront max (registers only) ----
    // key = (magnitude bits << 32) | (0x7fffffff - bin): non-negative floats order like their
    // bit patterns, so max(key) = largest magnitude, lowest bin on ties; out-of-band -> 0.
    // bins: even o -> (s < R3/2 ? lo : hi) + NS3*s, odd o -> M - that (thread 0, s = R3/2: M/2)
    unsigned long long best = 0ull;
    if (a.pitch) {
      const unsigned klo = 0x7fffffffu - (unsigned)out_lo, khi = 0x7fffffffu - (unsigned)out_hi;
      const unsigned nlo = 0x7fffffffu - (unsigned)(C::M - out_lo), nhi = 0x7fffffffu - (unsigned)(C::M - out_hi);
#pragma unroll
      for (int s = 0; s < C::R3; ++s) {
        constexpr int H = C::R3 / 2;
        if (!((umask >> (2 * s)) & 3u)) continue;  // wave-uniform: no lane has these two bins in band
        const unsigned b0 = (s < H ? klo : khi) - (unsigned)(C::NS3 * s);
        unsigned b1 = (s < H ? nlo : nhi) + (unsigned)(C::NS3 * s);
        if (s == H) b1 = (t == 0) ? 0x7fffffffu - (unsigned)(C::M / 2) : b1;
        const unsigned v0 = (unsigned)((int)(bmask << (31 - 2 * s)) >> 31);      // all-ones iff bit 2s
        const unsigned v1 = (unsigned)((int)(bmask << (31 - (2 * s + 1))) >> 31);  // all-ones iff bit 2s+1
        const unsigned long long k0 = ((unsigned long long)(__float_as_uint(mg[2 * s]) & v0) << 32) | (b0 & v0);
        const unsigned long long k1 = ((unsigned long long)(__float_as_uint(mg[2 * s + 1]) & v1) << 32) | (b1 & v1);
        best = k0 > best ? k0 : best;
        best = k1 > best ? k1 : best;
      }
      best = wave_max_u64(best);
      if constexpr (NW > 1) {
        if ((t & 63) == 0) red[t >> 6] = best;  // published by the next barrier
      }
    }

    // ---- magnitudes ----
    // Transpose through LDS: each lane scatters its E bins as dwords (consecutive lanes ->
    // consecutive bins, conflict-free), then every lane owns 4 consecutive bins and the row
    // leaves as global_store_dwordx4, 1 KiB contiguous per wavefront instruction, instead of E
    // dword stores with one stray element each (thread 0's self-paired bins).
    if constexpr (DIRECT) {
      if (want_rows) {
        // Wave-uniform row base (SGPR pair) + 32-bit lane offset + 13-bit immediate.  Slot s sits 4*NS3 bytes above slot
        // s - 1 (below, in the mirrored half): G = 8 KiB / (4*NS3) neighbouring slots share one lane offset and differ in the
        // store's immediate (-4096 .. +2048), so a frame's 2*R3 stores need 2*R3/G offset additions (N = 32768: 16 instead
        // of 32).  The compiler's own address matching does not find this form (it falls back to 64-bit per-lane
        // addresses): the store is spelled out.
        const float *row = a.mags + (size_t)f * (size_t)(N / 2);
        constexpr int H = C::R3 / 2;
        constexpr int STEP = 4 * C::NS3;  // bytes between a thread's consecutive slots
        constexpr int G = 8192 / STEP;    // slots per shared offset
        static_assert((STEP == 4096 || STEP == 2048) && H % G == 0, "slot groups share an offset through the store's immediate");
        const unsigned blo = 4u * (unsigned)out_lo, bhi = 4u * (unsigned)(out_hi + C::NS3 * H);
        const unsigned nlo = 4u * (unsigned)(C::M - out_lo), nhi = 4u * (unsigned)(C::M - out_hi - C::NS3 * H);
        static_for<0, C::R3>([&](auto ss) {
          constexpr int s = decltype(ss)::value;
          constexpr int sl = s < H ? s : s - H;
          // the slot of the group whose position carries the lane offset (going up / going down)
          constexpr int su = (sl & ~(G - 1)) + G / 2, sd = (sl & ~(G - 1)) + G / 2 - 1;
          const unsigned o0 = (s < H ? blo : bhi) + (unsigned)(su * STEP);
          unsigned o1 = (s < H ? nlo : nhi) - (unsigned)(sd * STEP);
          if constexpr (s == H) {  // thread 0's mirrored bin of this slot is bin M/2 instead of the Nyquist bin
            o1 = (t == 0) ? (unsigned)(4 * (C::M / 2)) + (unsigned)((sl - sd) * STEP) : o1;
          }
          st_row_nt<(sl - su) * STEP>(row, o0, mg[2 * s]);
          st_row_nt<-(sl - sd) * STEP, false>(row, o1, mg[2 * s + 1]);  // mirrored half: write-back, L2 merges the partial blocks
        });
      }
    } else if (want_rows) {
      float *plo = lout + out_lo, *phi = lout + out_hi;
      float *mlo = lout + (C::M - out_lo), *mhi = lout + (C::M - out_hi);
#pragma unroll
      for (int s = 0; s < C::R3; ++s) {
        constexpr int H = C::R3 / 2;
        (s < H ? plo : phi)[C::NS3 * s] = mg[2 * s];
        if (s == H) (t == 0 ? lout + C::M / 2 : mhi - C::NS3 * H)[0] = mg[2 * s + 1];
        else (s < H ? mlo : mhi)[-C::NS3 * s] = mg[2 * s + 1];
      }
    }
    if constexpr (!DEFER && !DIRECT) {
      if (want_rows) {
        MX_BARRIER();  // (also: every wave is past load_t2, so the image may be refilled)
        flush_row(f, t);
        MX_BARRIER();  // image free again
      } else {
        MX_BARRIER();
      }
      if constexpr (NW > 1) {
        flush_pitch(f, t);  // red[] is rewritten only after the next frame's T1/T2 barriers
      } else if (a.pitch && t == 0) {
        mx_pitch p;
        p.bin = 0x7fffffff - (int)(unsigned)(best & 0xffffffffull);
        p.mag = __uint_as_float((unsigned)(best >> 32));
        a.pitch[f] = p;
      }
    }
  }
  if constexpr (DEFER || DIRECT) {
    if (f0 < f1) {  // the last frame of this workgroup
      MX_BARRIER();
      flush_pitch(f1 - 1, t_);
      if constexpr (DEFER) flush_row(f1 - 1, t_);
    }
  }
}


}  // namespace mx
