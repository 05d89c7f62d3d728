// stft_kernel_ovl.h — the bulk STFT kernel of the 32-points-per-thread plans (N = 16384 / 32768) with the two LDS
// transpositions woven into the neighbouring passes (stft_overlap.h).  Same arithmetic as stft_kernel<..., DIRECT> of
// stft_kernel_impl.h on the same values in the same order — rows and pitch records are bit-identical —, another
// schedule: per frame
//   pass 1, stages 1-4                                (registers)
//   barrier: the image is free                        (every wave is past the previous frame's T2 gather)
//   pass 1, last stage  +  T1 scatter  +  the window ageing into the NEXT frame, woven butterfly by butterfly
//   barrier
//   T1 gather  +  pass-2 leaves, reads three leaves ahead behind counted lgkmcnt waits; pass 2, stages 2-4
//   barrier: every wave has gathered
//   pass 2, last stage  +  T2 scatter  +  the pass-3 twiddle products, woven
//   barrier
//   T2 gather  +  pass-3 leaves, six leaves ahead; pass 3, stages 2-4; split, magnitudes, pitch pick, row stores
// The carried window image Y is aged into frame f+1 while frame f's pass-1 outputs drain into the LDS: that is the
// "second frame in flight" the registers allow — Y exists anyway, the transform's own 64 registers are not duplicated.
#pragma once
#include "stft_kernel_impl.h"
#include "stft_overlap.h"

namespace mx {

// HOP > 0: the sliding window for that hop (Slide<P, HOP>::ok); HOP == 0: the circular window (Circ<P>::ok(a.hop)).
template <class P, int HOP, int WPE>
__global__ __launch_bounds__(P::T) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void stft_kernel_ovl(const StftArgs a0) {
#if defined(__HIP_DEVICE_COMPILE__)
  const StftArgs &a = a0;
  using C = P;
  constexpr int N = P::N;
  constexpr int NW = C::T / 64;
  static_assert(P::E == 32 && P::R1 == 32 && P::NB1 == 1 && P::R3 == 16 && NW > 1 && t1_padded<P>(), "32 points per thread");
  static_assert(P::NB2 * (P::R2 / 2) == P::R3, "one pass-3 twiddle product per last-stage butterfly of pass 2");
  constexpr bool kSlide = HOP > 0;
  static_assert(!kSlide || Slide<P, (HOP > 0 ? HOP : 2)>::ok, "hop the plan can slide by");
  constexpr int SD = Slide<P, (HOP > 0 ? HOP : 2)>::D;
  constexpr int kRed = ((NW + 1) / 2) * 2;
  constexpr int kTw2 = ((C::TW2 + 1) / 2) * 2;
  constexpr int IMG = t1_size<C>();
  constexpr float kSc = 0.5f / (float)N;
  __shared__ __attribute__((aligned(16))) float2 lds[IMG + kRed + kTw2];
  float2 *const ltw2 = lds + IMG + kRed;

  const int t_ = threadIdx.x;
  const bool wave0 = __builtin_amdgcn_readfirstlane(t_) < 64;  // wave-uniform
  cpx ulo, uhi_unused;
  post_bases<P>(t_, a.ubase, ulo, uhi_unused);
  cpx w3base[6];
  fetch_tw3_bases<P>(t_, a.tw3, w3base);
  const uint32_t bmask_ = band_mask<P>(t_, a.kmin, a.kmax);
  uint32_t umask = 0;
#pragma unroll
  for (int o = 0; o < P::E; ++o) umask |= (__ballot((bmask_ >> o) & 1u) != 0ull) ? (1u << o) : 0u;
  umask = __builtin_amdgcn_readfirstlane(umask);
  for (int i = t_; i < C::TW2; i += C::T) ltw2[i] = a.tw2[i];
  __syncthreads();

  // XCD-aware block -> frame-range map (stft_kernel_impl.h)
  unsigned lb = blockIdx.x;
  {
    const unsigned nb = gridDim.x, xcd = lb & 7u, q = nb >> 3, r = nb & 7u;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lb >> 3);
  }
  const int64_t f0 = (int64_t)lb * a.frames_per_block;
  const int64_t f1 = (f0 + a.frames_per_block < a.count) ? f0 + a.frames_per_block : a.count;
  if (f0 >= f1) return;  // (block-uniform; no barrier is pending)

  cpx Y[P::E];  // the windowed frame, carried from frame to frame
  cpx edge[SD], nx[SD];
  float cpx_[2 * Circ<P>::CS], cpw_[2 * Circ<P>::CS];
  if constexpr (kSlide) {
    const int64_t e0 = (a.first_frame + f0 + 1) * (int64_t)HOP;
    load_frame<P, 1, true>(t_, Y, a.audio + MX_AUDIO_PAD + (e0 - N), a.wtab);
    slide_edge<P, HOP>(t_, a.wtab, 2.0f * (float)N, edge);
  } else {
    const int64_t pe = (a.first_frame + f0 + 1) * (int64_t)a.hop;
    circ_load_first<P>(t_, Y, a.audio + MX_AUDIO_PAD + (pe - N), a.wtab, circ_geo<P>(pe, a.hop).o);
  }

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

  for (int64_t f = f0; f < f1; ++f) {
    // (thread-index invariants are re-materialised per frame: stft_kernel_impl.h)
    int t = t_;
    unsigned bmask = bmask_;
    int zoff = 0;
    asm volatile("" : "+v"(t), "+v"(bmask), "+s"(zoff));
    int out_lo, out_hi;
    out_bases<P>(t, out_lo, out_hi);

    // ---- the next frame's newest samples are requested now and land under pass 1 --------------------------------------
    // (also for the frame after the run's last: the image is aged once more than needed instead of branching; the reads
    // stay inside the padded image — at most two hops past the launch's last frame)
    CircGeo<P> geo{};
    if constexpr (kSlide) {
      slide_fetch<P, HOP>(t, a.audio + MX_AUDIO_PAD + (a.first_frame + f + 2) * (int64_t)HOP, nx);
    } else {
      const int64_t pe = (a.first_frame + f + 2) * (int64_t)a.hop;
      geo = circ_geo<P>(pe, a.hop);
      circ_fetch<P>(t, a.audio + MX_AUDIO_PAD + (pe - 2 * (int64_t)a.hop), a.wtab + zoff + (N - 2 * a.hop), geo, cpx_, cpw_);
    }

    // ---- pass 1, stages 1-4 ----------------------------------------------------------------------------------------
    cpx E1[P::R1 / 2], O1[P::R1 / 2];
    {
      cpx e[P::R1 / 2], o[P::R1 / 2];
#pragma unroll
      for (int q = 0; q < P::R1 / 2; ++q) {
        e[q] = Y[2 * q];
        o[q] = Y[2 * q + 1];
      }
      Dft<P::R1 / 2>::run(e, E1);
      Dft<P::R1 / 2>::run(o, O1);
    }
    lds_drain_barrier();  // the image is free: every wave is past the previous frame's T2 gather (and its red[] is out)

    // ---- pass 1, last stage + T1 scatter + ageing of Y into frame f+1 -------------------------------------------------
    {
      const uint32_t p1 = lds_addr(lds + (t * P::R1 + ((t * P::R1) >> 5)));  // store_t1's base (padded layout)
      if constexpr (kSlide) {
        using S = Slide<P, HOP>;
        const cpx gg = mk(a.decay, a.decay), ss = mk(kSc, kSc);
        final_stage_store<P::R1, 8>(E1, O1, p1, [&](auto qq) {  // slide_step, two slots per butterfly
          static_for<0, 2>([&](auto hh) {
            constexpr int e = 2 * decltype(qq)::value + decltype(hh)::value;
            if constexpr (e < P::E - 2 * S::D) Y[e] = pk_mul_xs(Y[e + S::D], gg);
            else if constexpr (e < P::E - S::D) Y[e] = pk_mul(Y[e + S::D], edge[e - (P::E - 2 * S::D)]);
            else Y[e] = pk_mul_xs(nx[e - (P::E - S::D)], ss);
          });
        });
      } else {
        const cpx gg = mk(a.decay, a.decay);
        auto weave = [&](auto sconst) {  // circ_age<S>, two slots per butterfly
          constexpr int S = decltype(sconst)::value;
          cpx head[S > 0 ? S : 1];
#pragma unroll
          for (int i = 0; i < S; ++i) head[i] = Y[i];
          final_stage_store<P::R1, 8>(E1, O1, p1, [&](auto qq) {
            static_for<0, 2>([&](auto hh) {
              constexpr int e = 2 * decltype(qq)::value + decltype(hh)::value;
              if constexpr (e + S < P::E) Y[e] = pk_mul_xs(Y[e + S], gg);
              else Y[e] = pk_mul_xs(head[e + S - P::E], gg);
            });
          });
        };
        constexpr int SMAX = (Circ<P>::W - 1 + (Circ<P>::CS - 1) * P::T) / Circ<P>::W;
        static_assert(SMAX <= 2, "rotations by 0, 1 or 2 slots");
        if (geo.s == 0) weave(std::integral_constant<int, 0>{});  // wave-uniform
        else if (SMAX == 1 || geo.s == 1) weave(std::integral_constant<int, 1>{});
        else weave(std::integral_constant<int, 2>{});
        // the newest 2*hop samples, each by its exact table weight (circ_step's second half)
#pragma unroll
        for (int k = 0; k < Circ<P>::CS; ++k) {
          const int e = Circ<P>::slot(k);
          const int c2 = 2 * (t + P::T * e);
          const int d0 = (c2 - geo.cr) & (P::N - 1), d1 = (c2 + 1 - geo.cr) & (P::N - 1);
          const float n0 = cpx_[2 * k] * cpw_[2 * k], n1 = cpx_[2 * k + 1] * cpw_[2 * k + 1];
          Y[e] = mk(d0 < geo.len ? n0 : Y[e].x, d1 < geo.len ? n1 : Y[e].y);
        }
      }
    }
    lds_drain_barrier();
    if (f > f0) flush_pitch(f - 1, t);

    // ---- T1 gather + pass 2 ---------------------------------------------------------------------------------------------
    cpx E2[P::NB2][P::R2 / 2], O2[P::NB2][P::R2 / 2];
    {
      cpx L[P::E];
      gather_t1_leaves<P>(t, lds, ltw2, L);
#pragma unroll
      for (int b = 0; b < P::NB2; ++b) {
        DftFromLeaves<P::R2 / 2, 2, 0>::run(L + b * P::R2, E2[b]);
        DftFromLeaves<P::R2 / 2, 2, 1>::run(L + b * P::R2, O2[b]);
      }
    }
    lds_drain_barrier();  // every wave has gathered: the image may be overwritten

    // ---- pass 2, last stage + T2 scatter + the pass-3 twiddle products -----------------------------------------------
    cpx w3[P::R3];  // w3[r] = gamma^r, r = 1..15 (w3[0] unused)
    static_for<0, P::NB2>([&](auto bb) {
      constexpr int b = decltype(bb)::value;
      const int j = t + P::T * b, k = j & (P::R1 - 1);
      const uint32_t p2 = lds_addr(lds + ((j - k) * P::R2 + k));  // store_t2's base
      final_stage_store<P::R2, P::R1 * 8>(E2[b], O2[b], p2, [&](auto qq) {
        constexpr int r = b * (P::R2 / 2) + decltype(qq)::value;  // one twiddle per butterfly (pass3_bases' products)
        if constexpr (r >= 1 && r < P::R3) {
          constexpr int hi = r >> 2, lo = r & 3;
          if constexpr (hi == 0) w3[r] = w3base[lo - 1];
          else if constexpr (lo == 0) w3[r] = w3base[2 + hi];
          else w3[r] = pk_cmul2(w3base[2 + hi], w3base[lo - 1]);
        }
      });
    });
    lds_drain_barrier();

    // ---- T2 gather + pass 3 + split ----------------------------------------------------------------------------------
    float mg[P::E];
    {
      cpx v[P::E], wq[P::R3];
      wq[0] = mk(1.0f, 0.0f);
#pragma unroll
      for (int r = 1; r < P::R3; ++r) wq[r] = w3[r];
      cpx LP[P::R3], LQ[P::R3];
      if (wave0) {  // wave-uniform: only the first wavefront contains thread 0 (whose P-butterfly takes twiddle 1)
        cpx wp[P::R3];
        wp[0] = mk(1.0f, 0.0f);
#pragma unroll
        for (int r = 1; r < P::R3; ++r) wp[r] = csel(t == 0, mk(1.0f, 0.0f), w3[r]);
        gather_t2_leaves<P>(t, lds, wp, wq, LP, LQ);
        DftFromLeaves<P::R3, 1, 0>::run(LP, v);
        DftFromLeaves<P::R3, 1, 0>::run(LQ, v + P::R3);
        post_fly<P, true>(t, v, ulo, csel(t == 0, mk(-1.0f, 0.0f), ulo), mg);
      } else {
        gather_t2_leaves<P>(t, lds, wq, wq, LP, LQ);
        DftFromLeaves<P::R3, 1, 0>::run(LP, v);
        DftFromLeaves<P::R3, 1, 0>::run(LQ, v + P::R3);
        post_fly<P, false>(t, v, ulo, ulo, mg);
      }
    }

    // ---- pitch pick (stft_kernel_impl.h: the same keys) -------------------------------------------------------------
    if (a.pitch) {
      unsigned long long best = 0ull;
      const unsigned klo = 0x7fffffffu - (unsigned)out_lo, khi = 0x7fffffffu - (unsigned)out_hi;
      const unsigned nlo = 0x7fffffffu - (unsigned)(C::M - out_lo), nhi = 0x7fffffffu - (unsigned)(C::M - out_hi);
#pragma unroll
      for (int s = 0; s < C::R3; ++s) {
        constexpr int H = C::R3 / 2;
        if (!((umask >> (2 * s)) & 3u)) continue;  // wave-uniform: no lane has these two bins in band
        const unsigned b0 = (s < H ? klo : khi) - (unsigned)(C::NS3 * s);
        unsigned b1 = (s < H ? nlo : nhi) + (unsigned)(C::NS3 * s);
        if (s == H) b1 = (t == 0) ? 0x7fffffffu - (unsigned)(C::M / 2) : b1;
        const unsigned v0 = (unsigned)((int)(bmask << (31 - 2 * s)) >> 31);
        const unsigned v1 = (unsigned)((int)(bmask << (31 - (2 * s + 1))) >> 31);
        const unsigned long long k0 = ((unsigned long long)(__float_as_uint(mg[2 * s]) & v0) << 32) | (b0 & v0);
        const unsigned long long k1 = ((unsigned long long)(__float_as_uint(mg[2 * s + 1]) & v1) << 32) | (b1 & v1);
        best = k0 > best ? k0 : best;
        best = k1 > best ? k1 : best;
      }
      best = wave_max_u64(best);
      if ((t & 63) == 0) red[t >> 6] = best;  // published by the next barrier
    }

    // ---- the row, straight from the registers (stft_kernel_impl.h, DIRECT) ---------------------------------------------
    if (a.mags != nullptr) {
      const float *row = a.mags + (size_t)f * (size_t)(N / 2);
      constexpr int H = C::R3 / 2;
      constexpr int STEP = 4 * C::NS3;
      constexpr int G = 8192 / STEP;
      static_assert((STEP == 4096 || STEP == 2048) && H % G == 0, "slot groups share an offset through the store's immediate");
      const unsigned blo = 4u * (unsigned)out_lo, bhi = 4u * (unsigned)(out_hi + C::NS3 * H);
      const unsigned nlo = 4u * (unsigned)(C::M - out_lo), nhi = 4u * (unsigned)(C::M - out_hi - C::NS3 * H);
      static_for<0, C::R3>([&](auto ss) {
        constexpr int s = decltype(ss)::value;
        constexpr int sl = s < H ? s : s - H;
        constexpr int su = (sl & ~(G - 1)) + G / 2, sd = (sl & ~(G - 1)) + G / 2 - 1;
        const unsigned o0 = (s < H ? blo : bhi) + (unsigned)(su * STEP);
        unsigned o1 = (s < H ? nlo : nhi) - (unsigned)(sd * STEP);
        if constexpr (s == H) {
          o1 = (t == 0) ? (unsigned)(4 * (C::M / 2)) + (unsigned)((sl - sd) * STEP) : o1;
        }
        st_row_nt<(sl - su) * STEP>(row, o0, mg[2 * s]);
        st_row_nt<-(sl - sd) * STEP>(row, o1, mg[2 * s + 1]);
      });
    }
  }
  lds_drain_barrier();
  flush_pitch(f1 - 1, t_);
#endif
}

}  // namespace mx
