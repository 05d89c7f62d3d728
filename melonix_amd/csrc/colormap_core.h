// colormap_core.h — SpecCache::populateTex's colormap (reference spec-cache.cpp:77-96) for one texel,
// operation by operation:
//   v = clamp(mag*k, 0, 255)                      binary32
//   v < 85 (=255/3):   (uchar)v, 0, 0
//   v < 170 (=2*255/3): a = (v-85)/85 [binary32] * 3.141592 / 2 [binary64];
//                       (uchar)(v*cos(a)), (uchar)(v*sin(a)), 0   [binary64 products, truncation]
//   else:               l = (uchar)((v-170)*3); l, (uchar)v, l
// Translation units that include this are built with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mx {

__device__ __forceinline__ void texel(float mag, float k, unsigned &r, unsigned &g, unsigned &b) {
  float v = mag * k;
  v = v < 0.f ? 0.f : (255.f < v ? 255.f : v);  // std::clamp(v, 0.f, 255.f)
  if (v < 85.f) {
    r = (unsigned)(unsigned char)v; g = 0; b = 0;
  } else if (v < 170.f) {
    const double a = (double)((v - 85.f) / 85.f) * 3.141592 / 2;
    r = (unsigned)(unsigned char)((double)v * cos(a));
    g = (unsigned)(unsigned char)((double)v * sin(a));
    b = 0;
  } else {
    const unsigned l = (unsigned)(unsigned char)((v - 170.f) * 3.f);
    r = l; g = (unsigned)(unsigned char)v; b = l;
  }
}

// four consecutive texels -> 12 bytes: r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
__device__ __forceinline__ void texel4(float m0, float m1, float m2, float m3, float k, uint32_t (&o)[3]) {
  unsigned r0, g0, b0, r1, g1, b1, r2, g2, b2, r3, g3, b3;
  texel(m0, k, r0, g0, b0);
  texel(m1, k, r1, g1, b1);
  texel(m2, k, r2, g2, b2);
  texel(m3, k, r3, g3, b3);
  o[0] = r0 | (g0 << 8) | (b0 << 16) | (r1 << 24);
  o[1] = g1 | (b1 << 8) | (r2 << 16) | (g2 << 24);
  o[2] = b2 | (r3 << 8) | (g3 << 16) | (b3 << 24);
}

}  // namespace mx
