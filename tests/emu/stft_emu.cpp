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
template <int N>
void emu_from_state(std::vector<cpx> &Yall, float *mags, std::vector<int> *collisions) {
  using C = Cfg<N>;
  static const std::vector<cpx_h> tw2 = make_tw2<N>();
  static const std::vector<cpx_h> tw3 = make_tw3<N>();
  static const std::vector<cpx_h> ub = make_ubase<N>();
  std::vector<cpx> lds((size_t)C::M);
  std::vector<cpx> regs((size_t)C::T * 32);
  std::vector<char> written((size_t)C::M);
  auto V = [&](int t) -> cpx(&)[32] { return *reinterpret_cast<cpx(*)[32]>(&regs[(size_t)t * 32]); };
  auto Y = [&](int t) -> cpx(&)[32] { return *reinterpret_cast<cpx(*)[32]>(&Yall[(size_t)t * 32]); };

  for (int t = 0; t < C::T; ++t) pass1<N>(Y(t), V(t));
  std::fill(written.begin(), written.end(), 0);
  for (int t = 0; t < C::T; ++t) {
    store_t1<N>(t, V(t), lds.data());
    // the closed-form addresses of store_t1 must be the swizzle swz1 of the logical index
    for (int b = 0; b < C::NB1; ++b)
      for (int r = 0; r < C::R1; ++r) {
        const int a = swz1<N>((t + C::T * b) * C::R1 + r);
        const cpx got = lds[(size_t)a], want = V(t)[b * C::R1 + r];
        if ((got.x != want.x || got.y != want.y) && collisions) collisions->push_back(-a - 1);
        if (written[(size_t)a]++ && collisions) collisions->push_back(a);
      }
  }
  for (int t = 0; t < C::T; ++t) load_t1<N>(t, V(t), lds.data());
  for (int t = 0; t < C::T; ++t) pass2<N>(t, V(t), reinterpret_cast<const cpx *>(tw2.data()));
  for (int t = 0; t < C::T; ++t) store_t2<N>(t, V(t), lds.data());
  for (int t = 0; t < C::T; ++t) load_t2<N>(t, V(t), lds.data());
  for (int t = 0; t < C::T; ++t) {
    pass3<N>(t, V(t), reinterpret_cast<const cpx *>(tw3.data()));
    cpx u[16];
    post_twiddles<N>(t, reinterpret_cast<const cpx *>(ub.data()), u);
    float mg[32];
    post<N>(t, V(t), u, mg);
    for (int o = 0; o < 32; ++o) mags[out_bin<N>(t, o)] = mg[o];
  }
}

template <int N, int WSTEP>
void emu_frame(const float *x, const float *w, float *mags, std::vector<int> *collisions) {
  using C = Cfg<N>;
  std::vector<cpx> Yall((size_t)C::T * 32);
  for (int t = 0; t < C::T; ++t)
    load_frame<N, WSTEP, false>(t, *reinterpret_cast<cpx(*)[32]>(&Yall[(size_t)t * 32]), x, w);
  emu_from_state<N>(Yall, mags, collisions);
}

// Sliding mode: frames first..first+count-1 of the uniform-hop indexing, the first loaded in
// full, the rest slid through the register image exactly as the kernel does.
template <int N, int HOP>
int run_slide(const float *wav, long n, long first, long count, float *mags) {
  using C = Cfg<N>;
  using S = Slide<N, HOP>;
  static_assert(S::ok, "hop not slidable");
  const int PAD = 32768;
  std::vector<float> padded((size_t)n + 2 * PAD, 0.0f);
  std::memcpy(padded.data() + PAD, wav, sizeof(float) * (size_t)n);
  static const std::vector<float> wext = make_wext(fold_scale(N));
  const std::vector<float> wtab = make_wtab(N, HOP, wext);
  const float g = hop_decay(HOP), sc = fold_scale(N);
  std::vector<cpx> Yall((size_t)C::T * 32);
  std::vector<int> coll;
  for (long f = 0; f < count; ++f) {
    const long e = (first + f + 1) * (long)HOP;
    if (f == 0) {
      for (int t = 0; t < C::T; ++t)
        load_frame<N, 1, true>(t, *reinterpret_cast<cpx(*)[32]>(&Yall[(size_t)t * 32]),
                               padded.data() + PAD + (e - N), wtab.data());
    } else {
      for (int t = 0; t < C::T; ++t) {
        cpx edge[S::D], nx[S::D];
        slide_edge<N, HOP>(t, wtab.data(), 2.0f * (float)N, edge);
        slide_fetch<N, HOP>(t, padded.data() + PAD + e, nx);
        slide_step<N, HOP>(*reinterpret_cast<cpx(*)[32]>(&Yall[(size_t)t * 32]), nx, edge, g, sc);
      }
    }
    emu_from_state<N>(Yall, mags + (size_t)f * (N / 2), &coll);
  }
  return (int)coll.size();
}

template <int N>
int run(const float *wav, long n, int start, int end, int hop_mode, float *mags) {
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
    emu_frame<N, 1>(x, wtab.data(), mags, &coll);
  } else {
    long D0 = (long)N - ((long)end - (long)start);
    D0 = std::max<long>(D0, (long)N - 1 - kWOff);
    D0 = std::min<long>(D0, (long)kWDmax + kWTail);
    emu_frame<N, -1>(x, wext.data() + kWOff + D0, mags, &coll);
  }
  return (int)coll.size();
}

}  // namespace

// returns the number of LDS index collisions (must be 0), or <0 on bad N
extern "C" int emu_stft_frame(int N, const float *wav, long n, int start, int end, int hop_mode,
                              float *mags) {
  switch (N) {
    case 4096: return run<4096>(wav, n, start, end, hop_mode, mags);
    case 16384: return run<16384>(wav, n, start, end, hop_mode, mags);
    case 32768: return run<32768>(wav, n, start, end, hop_mode, mags);
  }
  return -1;
}

// frames [first, first+count) with the sliding register image; mags = count x N/2
extern "C" int emu_stft_slide(int N, int hop, const float *wav, long n, long first, long count, float *mags) {
  if (N == 4096 && hop == 256) return run_slide<4096, 256>(wav, n, first, count, mags);
  if (N == 16384 && hop == 512) return run_slide<16384, 512>(wav, n, first, count, mags);
  return -1;
}
