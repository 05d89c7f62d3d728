// stft_tables.h — host-side builders of the constant tables the STFT kernels
// read from HBM/L2: pass twiddles, the post-split bases and the window weights.
// Host code only (plain C++17); the tables are uploaded once per context.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "stft_core.h"

namespace mx {

struct cpx_h {  // layout-compatible with cpx / float2
  float x, y;
};

inline cpx_h unit_root(long long num, long long den) {  // exp(-2*pi*i*num/den)
  num %= den;
  if (num < 0) num += den;
  const long double a = 6.283185307179586476925286766559005768L * (long double)num / (long double)den;
  cpx_h r;
  r.x = (float)cosl(a);
  r.y = (float)(-sinl(a));
  // exact values at the quadrant points
  if ((4 * num) % den == 0) {
    const int q = (int)((4 * num) / den);
    r.x = (float)((q == 0) - (q == 2));
    r.y = (float)((q == 3) - (q == 1));
  }
  return r;
}

template <class C>
std::vector<cpx_h> make_tw2() {
  std::vector<cpx_h> t((size_t)C::TW2);
  for (int r = 1; r < C::R2; ++r)
    for (int k = 0; k < C::R1; ++k) t[(size_t)(r - 1) * C::R1 + k] = unit_root((long long)r * k, C::R1 * C::R2);
  return t;
}

template <class C>
std::vector<cpx_h> make_tw3() {
  std::vector<cpx_h> t((size_t)C::TW3);
  for (int r = 1; r < C::R3; ++r)
    for (int k = 0; k < C::NS3; ++k) t[(size_t)(r - 1) * C::NS3 + k] = unit_root((long long)r * k, C::M);
  return t;
}

// ubase[t] = i * exp(-2*pi*i*t/N), t = 0..T-1
template <class C>
std::vector<cpx_h> make_ubase() {
  std::vector<cpx_h> t((size_t)C::T);
  for (int k = 0; k < C::T; ++k) {
    const cpx_h w = unit_root(k, C::N);
    t[(size_t)k].x = -w.y;  // i*(a+ib) = -b + ia
    t[(size_t)k].y = w.x;
  }
  return t;
}

// Window weight by distance d = start - i (spec.cpp:55-58):
//   d <= 0 -> 1;  d >= 1 -> expf(-2.5e-4f * d)   (the host libm's expf, exactly
//   the call the reference makes; d converts to float exactly for d < 2^24).
// Stored as wext[d + kWOff], d in [-kWOff, kWDmax]; beyond kWDmax the weight
// underflows to exactly 0.f (expf(-105) == 0 in binary32), so the table is
// complete for every representable distance.  A frame's distances are
// d = D0 - p, p = 0..N-1, D0 = N - (end - start); clamping D0 into
// [N-1-kWOff, kWDmax+kWTail] keeps every lookup inside the table without
// changing any weight (below: all ones; above: all zeros).
constexpr int kWOff = 32768;
constexpr int kWDmax = 425984;  // 2.5e-4 * 425984 = 106.5 > 103.98 = ln(2^150)
constexpr int kWTail = 32768;   // zeros past kWDmax so a clamped frame never leaves the table

// sc: output scale folded into the weights (1/(2N), a power of two: w*sc is exact, and
// x*(w*sc) == (x*w)*sc bit for bit, so the kernel's magnitudes need no final multiply).
inline std::vector<float> make_wext(float sc = 1.0f) {
  std::vector<float> w((size_t)kWOff + kWDmax + kWTail + 1, 0.0f);
  for (int d = -kWOff; d <= 0; ++d) w[(size_t)(d + kWOff)] = sc;
  for (int d = 1; d <= kWDmax; ++d) w[(size_t)(d + kWOff)] = expf(-2.5e-4f * d) * sc;
  return w;
}
inline float fold_scale(int N) { return 0.5f / (float)N; }
// per-hop decay of the sliding window: exp(-2.5e-4*hop), rounded once
inline float hop_decay(int hop) { return (float)exp(-2.5e-4 * (double)hop); }

// Bulk mode (uniform hop): weight of frame position p is that of d = N-hop-p.
// Section 1, w[0..N): the exact weights (sliding kernels' first frame, slide_edge).
// Section 2, w[N..N+kWBase): the UNCLAMPED weights sc*exp(-2.5e-4f*d) of positions p < kWBase
// (no flat top: for d <= 0 they exceed sc) — the per-thread seed of load_frame_geo.
constexpr int kWBase = 1024;  // 2 * the largest T
inline std::vector<float> make_wtab(int N, int hop, const std::vector<float> &wext) {
  std::vector<float> w((size_t)N + kWBase);
  for (int p = 0; p < N; ++p) {
    long long d = (long long)N - hop - p;
    if (d < -kWOff) d = -kWOff;
    if (d > kWDmax) d = kWDmax;
    w[(size_t)p] = wext[(size_t)(d + kWOff)];
  }
  const double sc = (double)wext[0];  // the folded output scale (weight of d <= 0)
  for (int p = 0; p < kWBase; ++p) {
    const double d = (double)N - (double)hop - (double)p;
    w[(size_t)N + p] = (float)(sc * exp(-(double)2.5e-4f * d));  // 0 once it underflows (huge d)
  }
  return w;
}

}  // namespace mx
