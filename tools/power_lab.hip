// tools/power_lab.hip — runs ONE variant of the headline STFT kernel (N = 4096, hop 256, 675 000 frames) back to back for
// a few seconds so that tools/power_lab.sh can sample the package power and the shader clock next to it: the kernel
// is power-limited (1400 W cap), so what a piece of it costs is its ENERGY per frame, = power x time / frames.
// Variants (compile-time, results wrong by construction): -DMX_ABL_NOLDS, -DMX_ABL_NOGSTORE, -DMX_ABL_NOOUTLDS,
// -DMX_ABL_NOVALU, -DLAB_PITCH_ONLY (no magnitude rows at all), -DLAB_IDLE (a kernel that only sleeps).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "lab/stft_kernel_lab.h"
using namespace mx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
#ifndef ABLNAME
#define ABLNAME "base"
#endif
__global__ void idle_kernel(int n) { for (int i = 0; i < n; ++i) asm volatile("s_sleep 64"); }

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  using C = Plan<4096, 16>;
  constexpr int N = 4096, HOP = 256;
  const int64_t n = 60LL * 60 * 48000, F = (n + HOP - 1) / HOP;
  std::vector<float> h((size_t)n + 2 * MX_AUDIO_PAD, 0.f);
  for (int64_t i = 0; i < n; ++i) { const double t = (double)i / 48000.0; h[(size_t)i + MX_AUDIO_PAD] = (float)(0.5 * sin(2 * 3.14159265358979 * (110.0 * t + 1650.0 * t * t / (2 * 3600.0)))); }
  float *d_audio; CK(hipMalloc(&d_audio, h.size() * 4)); CK(hipMemcpy(d_audio, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const auto tw2 = make_tw2<C>(); const auto tw3 = make_tw3<C>(); const auto ub = make_ubase<C>();
  const auto wext = make_wext(fold_scale(N)); const auto wtab = make_wtab(N, HOP, wext);
  float2 *d_tw2, *d_tw3, *d_ub; float *d_wext, *d_wtab;
  CK(hipMalloc(&d_tw2, tw2.size() * 8)); CK(hipMemcpy(d_tw2, tw2.data(), tw2.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_tw3, tw3.size() * 8)); CK(hipMemcpy(d_tw3, tw3.data(), tw3.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_ub, ub.size() * 8)); CK(hipMemcpy(d_ub, ub.data(), ub.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_wext, wext.size() * 4)); CK(hipMemcpy(d_wext, wext.data(), wext.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_wtab, wtab.size() * 4)); CK(hipMemcpy(d_wtab, wtab.data(), wtab.size() * 4, hipMemcpyHostToDevice));
  float *d_mags; mx_pitch *d_pitch;
  CK(hipMalloc(&d_mags, (size_t)F * (N / 2) * 4)); CK(hipMalloc(&d_pitch, (size_t)F * sizeof(mx_pitch)));
  StftArgs a{};
  a.audio = d_audio; a.n = n; a.wtab = d_wtab; a.wext = d_wext; a.decay = hop_decay(HOP); a.tw2 = d_tw2; a.tw3 = d_tw3; a.ubase = d_ub;
  a.hop = HOP; a.first_frame = 0; a.count = F; a.kmin = 5; a.kmax = 150; a.pitch = d_pitch; a.frames_per_block = 32;
#ifdef LAB_PITCH_ONLY
  a.mags = nullptr;
#else
  a.mags = d_mags;
#endif
  const unsigned blocks = (unsigned)((F + 31) / 32);
  auto launch = [&]() {
#ifdef LAB_IDLE
    hipLaunchKernelGGL(idle_kernel, dim3(256 * 8), dim3(256), 0, 0, 500);
#else
    hipLaunchKernelGGL((mxlab::stft_kernel<C, kBulkAligned, HOP, 3, true, true, 2, true, true, false, true>), dim3(blocks), dim3(C::T), 0, 0, a);
#endif
  };
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0; double ms_sum = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 50; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms_sum += ms; launches += 50;
  }
  printf("LAB %-12s %8.4f ms per launch (%ld launches)\n", ABLNAME, ms_sum / launches, launches);
  return 0;
}
