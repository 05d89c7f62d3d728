// tools/ubench_dft.hip — SIMD pipe cost and single-wave latency of the in-register DFT-16 butterflies,
// scalar f32 (round-1 code, kept here as tools/scalar_dft.inc) against the packed-f32 form of
// melonix_amd/csrc/stft_core.h, at 1..8 wavefronts per SIMD.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffp-contract=off -I melonix_amd/csrc tools/ubench_dft.hip -o tools/bin/ubench_dft
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "stft_core.h"

namespace sc {  // the scalar butterflies of round 1
using mx::cpx; using mx::mk; using mx::fma_; using mx::kCos64; using mx::kSin64;
MX_HD cpx cadd(cpx a, cpx b) { return mk(a.x + b.x, a.y + b.y); }
MX_HD cpx csub(cpx a, cpx b) { return mk(a.x - b.x, a.y - b.y); }
template <int K>
MX_HD cpx mulw64(cpx a) {
  constexpr int k = ((K % 64) + 64) % 64;
  if constexpr (k == 0) return a;
  else if constexpr (k == 16) return mk(a.y, -a.x);
  else if constexpr (k == 32) return mk(-a.x, -a.y);
  else return mk(-a.y, a.x);
}
template <int K>
MX_HD void bfly_w64(cpx E, cpx O, cpx &out0, cpx &out1) {
  constexpr int k = ((K % 64) + 64) % 64;
  constexpr float h = 0.707106781187f;
  if constexpr (k % 16 == 0) {
    const cpx t = mulw64<k>(O);
    out0 = cadd(E, t);
    out1 = csub(E, t);
  } else {
    if constexpr (k == 8) out0 = mk(fma_(h, O.x + O.y, E.x), fma_(h, O.y - O.x, E.y));
    else if constexpr (k == 24) out0 = mk(fma_(h, O.y - O.x, E.x), fma_(-h, O.x + O.y, E.y));
    else {
      constexpr float c = kCos64[k], sn = kSin64[k];
      out0 = mk(fma_(O.x, c, fma_(O.y, sn, E.x)), fma_(O.y, c, fma_(-O.x, sn, E.y)));
    }
    out1 = mk(fma_(2.0f, E.x, -out0.x), fma_(2.0f, E.y, -out0.y));
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
    for (int q = 0; q < R / 2; ++q) { e[q] = in[2 * q]; o[q] = in[2 * q + 1]; }
    Dft<R / 2>::run(e, E);
    Dft<R / 2>::run(o, O);
    Combine<R, 0>::run(E, O, out);
  }
};
template <>
struct Dft<2> {
  static MX_HD void run(const cpx *in, cpx *out) { out[0] = cadd(in[0], in[1]); out[1] = csub(in[0], in[1]); }
};
}  // namespace sc

struct Rec { unsigned long long cyc, real; };

template <int KIND>  // 0 scalar, 1 packed
__global__ void bench(Rec *rec, float2 *out, int iters) {
  extern __shared__ char dyn[];
  mx::cpx v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = mx::mk(1e-3f * (threadIdx.x + i), 1e-3f * (i * 7 - (int)threadIdx.x));
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    mx::cpx o[16];
    if constexpr (KIND == 0) sc::Dft<16>::run(v, o);
    else mx::Dft<16>::run(v, o);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = o[(i * 5) & 15];  // free renaming; keeps values bounded? (scaled below)
    // keep magnitudes bounded without adding VALU work per element: nothing — 1000 iterations of x16 growth
    // overflow to inf, which costs the same to compute
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    Rec r; r.cyc = t1 - t0; r.real = r1 - r0;
    rec[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r;
  }
  float2 s = make_float2(0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 16; ++i) { s.x += v[i].x; s.y += v[i].y; }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(int bpc, int iters, Rec *d_rec, float2 *d_out, int ninstr) {
  const int blocks = 256 * bpc;
  size_t lds = (160 * 1024 / bpc) & ~(size_t)255;
  if (lds > 160 * 1024) lds = 160 * 1024;
  if (bpc > 8) lds = 0;
  hipFuncSetAttribute((const void *)bench<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((bench<KIND>), dim3(blocks), dim3(256), lds, 0, d_rec, d_out, iters);
    hipDeviceSynchronize();
  }
  std::vector<Rec> h((size_t)blocks * 4);
  hipMemcpy(h.data(), d_rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
  double c = 0, rt = 0;
  for (auto &r : h) { c += r.cyc; rt += r.real; }
  c /= h.size(); rt /= h.size();
  printf("%s DFT16 %d waves/SIMD: %8.1f cycles per DFT16 per wave, %7.1f per SIMD  (%d VALU instr: %.2f cyc/instr/wave, %.2f pipe)  clk %.2f GHz\n",
         KIND ? "packed" : "scalar", bpc, c / iters, c / iters / bpc, ninstr, c / iters / ninstr, c / iters / bpc / ninstr, c / (rt * 10.0));
}

int main(int argc, char **argv) {
  Rec *d_rec; float2 *d_out;
  hipMalloc(&d_rec, sizeof(Rec) * 256 * 8 * 4);
  hipMalloc(&d_out, sizeof(float2) * 256 * 8 * 256);
  const int ns = argc > 1 ? atoi(argv[1]) : 148, np = argc > 2 ? atoi(argv[2]) : 80;
  for (int bpc : {1, 2, 3, 4, 6, 8}) run<0>(bpc, 2000, d_rec, d_out, ns);
  for (int bpc : {1, 2, 3, 4, 6, 8}) run<1>(bpc, 2000, d_rec, d_out, np);
  return 0;
}
