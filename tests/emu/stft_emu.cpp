// stft_emu.cpp — TEST INFRASTRUCTURE: runs one STFT workgroup thread-by-thread
// on the CPU through the very templates the gfx950 kernel instantiates
// (melonix_amd/csrc/stft_core.h), so index maps, swizzles, twiddles and the
// real-FFT split can be checked against the oracle without a GPU.
// This is not a product path and is never loaded by melonix_amd.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../melonix_amd/csrc/stft_core.h"
#include "../../melonix_amd/csrc/stft_tables.h"

using namespace mx;

namespace {

// One frame from its windowed points Y[t][32] (the kernel's per-thread register image).
template <class C>
void emu_from_state(std::vector<cpx> &Yall, float *mags, std::vector<int> *collisions) {
  constexpr int E = C::E;
  static const std::vector<cpx_h> tw2 = make_tw2<C>();
  static const std::vector<cpx_h> tw3 = make_tw3<C>();
  static const std::vector<cpx_h> ub = make_ubase<C>();
  std::vector<cpx> lds((size_t)t1_size<C>());
  std::vector<cpx> regs((size_t)C::T * E);
  std::vector<char> written((size_t)t1_size<C>());
  auto V = [&](int t) -> cpx(&)[E] { return *reinterpret_cast<cpx(*)[E]>(&regs[(size_t)t * E]); };
  auto Y = [&](int t) -> cpx(&)[E] { return *reinterpret_cast<cpx(*)[E]>(&Yall[(size_t)t * E]); };

  for (int t = 0; t < C::T; ++t) pass1<C>(Y(t), V(t));
  std::fill(written.begin(), written.end(), 0);
  for (int t = 0; t < C::T; ++t) {
    store_t1<C>(t, V(t), lds.data());
    // the closed-form addresses of store_t1 must be the padded layout t1_index of the logical index
    for (int b = 0; b < C::NB1; ++b)
      for (int r = 0; r < C::R1; ++r) {
        const int a = t1_index<C>((t + C::T * b) * C::R1 + r);
        const cpx got = lds[(size_t)a], want = V(t)[b * C::R1 + r];
        if ((got.x != want.x || got.y != want.y) && collisions) collisions->push_back(-a - 1);
        if (written[(size_t)a]++ && collisions) collisions->push_back(a);
      }
  }
  for (int t = 0; t < C::T; ++t) load_t1<C>(t, V(t), lds.data());
  for (int t = 0; t < C::T; ++t) pass2<C>(t, V(t), reinterpret_cast<const cpx *>(tw2.data()));
  for (int t = 0; t < C::T; ++t) store_t2<C>(t, V(t), lds.data());
  for (int t = 0; t < C::T; ++t) load_t2<C>(t, V(t), lds.data());
  for (int t = 0; t < C::T; ++t) {
    float mg[E];
    if constexpr (C::R3 == 16) {
      // the 32-points-per-thread plans rebuild nine of the fifteen pass-3 twiddles and the post-split twiddles
      // from per-thread bases (stft_kernels.hip, twiddle placement 6): the emulation does the same arithmetic
      cpx wb[6], lo, hi;
      fetch_tw3_bases<C>(t, reinterpret_cast<const cpx *>(tw3.data()), wb);
      pass3_bases<C>(t, V(t), wb);
      post_bases<C>(t, reinterpret_cast<const cpx *>(ub.data()), lo, hi);
      post_fly<C>(t, V(t), lo, hi, mg);
    } else {
      cpx w3[C::R3 - 1];  // (the N = 4096 plan: pass-3 and post-split twiddles in registers, as in the kernel)
      fetch_tw3<C>(t, reinterpret_cast<const cpx *>(tw3.data()), w3);
      pass3_reg<C, true>(t, V(t), w3);
      cpx u[C::R3];
      post_twiddles<C>(t, reinterpret_cast<const cpx *>(ub.data()), u);
      post<C>(t, V(t), u, mg);
    }
    for (int o = 0; o < E; ++o) mags[out_bin<C>(t, o)] = mg[o];
  }
}

// WSTEP = -1: ranges mode (exact d-indexed weights); WSTEP = +1: bulk mode, whose weights the
// kernel derives from a per-thread seed (load_frame_geo; w then points at the table's 2nd section)
template <class C, int WSTEP>
void emu_frame(const float *x, const float *w, int hop, float *mags, std::vector<int> *collisions) {
  constexpr int E = C::E;
  std::vector<cpx> Yall((size_t)C::T * E);
  for (int t = 0; t < C::T; ++t) {
    auto &Y = *reinterpret_cast<cpx(*)[E]>(&Yall[(size_t)t * E]);
    if constexpr (WSTEP == 1) load_frame_geo<C, false>(t, Y, x, w, hop);
    else load_frame<C, WSTEP, false>(t, Y, x, w);
  }
  emu_from_state<C>(Yall, mags, collisions);
}

// Sliding mode: frames first..first+count-1 of the uniform-hop indexing, the first loaded in
// full, the rest slid through the register image exactly as the kernel does.
template <class C, int HOP>
int run_slide(const float *wav, long n, long first, long count, float *mags) {
  constexpr int N = C::N, E = C::E;
  using S = Slide<C, HOP>;
  static_assert(S::ok, "hop not slidable");
  const int PAD = 32768;
  std::vector<float> padded((size_t)n + 2 * PAD, 0.0f);
  std::memcpy(padded.data() + PAD, wav, sizeof(float) * (size_t)n);
  static const std::vector<float> wext = make_wext(fold_scale(N));
  const std::vector<float> wtab = make_wtab(N, HOP, wext);
  const float g = hop_decay(HOP), sc = fold_scale(N);
  std::vector<cpx> Yall((size_t)C::T * E);
  std::vector<int> coll;
  for (long f = 0; f < count; ++f) {
    const long e = (first + f + 1) * (long)HOP;
    if (f == 0) {
      for (int t = 0; t < C::T; ++t)
        load_frame<C, 1, true>(t, *reinterpret_cast<cpx(*)[E]>(&Yall[(size_t)t * E]),
                               padded.data() + PAD + (e - N), wtab.data());
    } else {
      for (int t = 0; t < C::T; ++t) {
        cpx edge[S::D], nx[S::D];
        slide_edge<C, HOP>(t, wtab.data(), 2.0f * (float)N, edge);
        slide_fetch<C, HOP>(t, padded.data() + PAD + e, nx);
        slide_step<C, HOP>(*reinterpret_cast<cpx(*)[E]>(&Yall[(size_t)t * E]), nx, edge, g, sc);
      }
    }
    emu_from_state<C>(Yall, mags + (size_t)f * (N / 2), &coll);
  }
  return (int)coll.size();
}

// Circular sliding window (any small uniform hop): frames first..first+count-1, the first loaded directly in the
// slot-aligned circular layout, the rest stepped exactly as the kernel does (circ_fetch for the coming frame, circ_step).
template <class C>
int run_circ(const float *wav, long n, int hop, long first, long count, float *mags) {
  constexpr int N = C::N, E = C::E;
  if (!Circ<C>::ok(hop)) return -2;
  const int PAD = 32768;
  std::vector<float> padded((size_t)n + 2 * PAD, 0.0f);
  std::memcpy(padded.data() + PAD, wav, sizeof(float) * (size_t)n);
  static const std::vector<float> wext = make_wext(fold_scale(N));
  const std::vector<float> wtab = make_wtab(N, hop, wext);
  const float g = hop_decay(hop);
  std::vector<cpx> Yall((size_t)C::T * E);
  std::vector<int> coll;
  for (long f = 0; f < count; ++f) {
    const long long pe = (long long)(first + f + 1) * hop;
    const CircGeo<C> geo = circ_geo<C>(pe, hop);
    for (int t = 0; t < C::T; ++t) {
      auto &Y = *reinterpret_cast<cpx(*)[E]>(&Yall[(size_t)t * E]);
      if (f == 0) {
        circ_load_first<C>(t, Y, padded.data() + PAD + (pe - N), wtab.data(), geo.o);
      } else {
        float px[2 * Circ<C>::CS], pw[2 * Circ<C>::CS];
        circ_fetch<C>(t, padded.data() + PAD + (pe - 2LL * hop), wtab.data() + (N - 2 * hop), geo, px, pw);
        circ_step<C>(t, Y, g, geo, px, pw);
      }
    }
    emu_from_state<C>(Yall, mags + (size_t)f * (N / 2), &coll);
  }
  return (int)coll.size();
}

template <class C>
int run(const float *wav, long n, int start, int end, int hop_mode, float *mags) {
  constexpr int N = C::N;
  // padded copy, as the device layout [PAD zeros][n][PAD zeros]
  const int PAD = 32768;
  std::vector<float> padded((size_t)n + 2 * PAD, 0.0f);
  std::memcpy(padded.data() + PAD, wav, sizeof(float) * (size_t)n);
  if ((long)end <= 0 || (long)end - N >= n) {
    std::fill(mags, mags + N / 2, 0.0f);
    return 0;
  }
  const float *x = padded.data() + PAD + ((long)end - N);
  static const std::vector<float> wext = make_wext(fold_scale(N));
  std::vector<int> coll;
  if (hop_mode) {
    const std::vector<float> wtab = make_wtab(N, end - start, wext);
    emu_frame<C, 1>(x, wtab.data() + N, end - start, mags, &coll);
  } else {
    long D0 = (long)N - ((long)end - (long)start);
    D0 = std::max<long>(D0, (long)N - 1 - kWOff);
    D0 = std::min<long>(D0, (long)kWDmax + kWTail);
    emu_frame<C, -1>(x, wext.data() + kWOff + D0, end - start, mags, &coll);
  }
  return (int)coll.size();
}

}  // namespace

// returns the number of LDS index collisions (must be 0), or <0 on bad N
extern "C" int emu_stft_frame(int N, int E, const float *wav, long n, int start, int end, int hop_mode,
                              float *mags) {
  if (N == 4096 && E == 16) return run<Plan<4096, 16>>(wav, n, start, end, hop_mode, mags);
  if (N == 16384 && E == 32) return run<Plan<16384, 32>>(wav, n, start, end, hop_mode, mags);
  if (N == 32768 && E == 32) return run<Plan<32768, 32>>(wav, n, start, end, hop_mode, mags);
  return -1;
}

// frames [first, first+count) with the sliding register image; mags = count x N/2
extern "C" int emu_stft_slide(int N, int E, int hop, const float *wav, long n, long first, long count, float *mags) {
  if (N == 4096 && E == 16 && hop == 256) return run_slide<Plan<4096, 16>, 256>(wav, n, first, count, mags);
  if (N == 16384 && E == 32 && hop == 512) return run_slide<Plan<16384, 32>, 512>(wav, n, first, count, mags);
  if (N == 4096 && E == 16 && hop == 512) return run_slide<Plan<4096, 16>, 512>(wav, n, first, count, mags);
  if (N == 16384 && E == 32 && hop == 1024) return run_slide<Plan<16384, 32>, 1024>(wav, n, first, count, mags);
  if (N == 32768 && E == 32 && hop == 1024) return run_slide<Plan<32768, 32>, 1024>(wav, n, first, count, mags);
  return -1;
}

// frames [first, first+count) with the circular sliding window; mags = count x N/2 (-2: hop outside its range)
extern "C" int emu_stft_circ(int N, int E, int hop, const float *wav, long n, long first, long count, float *mags) {
  if (N == 4096 && E == 16) return run_circ<Plan<4096, 16>>(wav, n, hop, first, count, mags);
  if (N == 16384 && E == 32) return run_circ<Plan<16384, 32>>(wav, n, hop, first, count, mags);
  if (N == 32768 && E == 32) return run_circ<Plan<32768, 32>>(wav, n, hop, first, count, mags);
  return -1;
}
