// tools/timeline_lab.hip — one instantiation of the LAB copy of the STFT kernel (tools/lab/stft_kernel_lab.h), timed over the
// hour-long workload, and — with -DMX_TIMELINE — the phase timeline of one workgroup's waves for one frame (s_memtime at the
// phase boundaries, all outstanding memory operations drained at each stamp).  Not part of the product.
//   -DLAB_N=32768 -DLAB_E=32 -DLAB_MODE=1 -DLAB_THOP=0 -DLAB_HOP=375 -DLAB_WPE=2 -DLAB_TWREG=2 -DLAB_OUTSEP=0 -DLAB_DEFER=0
//   -DLAB_PREFETCH=0 -DLAB_EARLYBAR=0 -DLAB_G=8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef MX_TL_BLOCK
#define MX_TL_BLOCK 1000
#endif
#ifndef MX_TL_FRAME
#define MX_TL_FRAME 3
#endif
#ifdef LAB_PRODUCT  // the PRODUCT template (no timeline hooks): variants of its options
#include "stft_kernel_impl.h"
namespace mxlab = mx;
#else
#include "lab/stft_kernel_lab.h"
#endif
using namespace mx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
#ifndef LAB_NAME
#define LAB_NAME "variant"
#endif
int main(int argc, char **argv) {
  using C = Plan<LAB_N, LAB_E>;
  constexpr int N = LAB_N, HOP = LAB_HOP;
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
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
  a.hop = HOP; a.first_frame = 0; a.count = F; a.kmin = (int)(55.0 * N / 48000.0 + 0.999); a.kmax = (int)(1760.0 * N / 48000.0); a.pitch = d_pitch; a.mags = d_mags;
  a.frames_per_block = LAB_G;
  const unsigned blocks = (unsigned)((F + LAB_G - 1) / LAB_G);
  auto launch = [&]() {
    hipLaunchKernelGGL((mxlab::stft_kernel<C, LAB_MODE, LAB_THOP, LAB_WPE, true, true, LAB_TWREG, (bool)LAB_OUTSEP, (bool)LAB_DEFER, LAB_PREFETCH, (bool)LAB_EARLYBAR>),
                       dim3(blocks), dim3(C::T), 0, 0, a);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double balg = 4.0 * HOP + 4.0 * (N / 2) + 8.0;
  // fingerprint of sampled rows, so variants can be compared
  unsigned long long fp = 1469598103934665603ull;
  std::vector<float> row(N / 2);
  for (int64_t f : {(int64_t)0, (int64_t)1, (int64_t)17, F / 3, F / 2 + 5, F - 2, F - 1}) {
    CK(hipMemcpy(row.data(), d_mags + (size_t)f * (N / 2), (N / 2) * 4, hipMemcpyDeviceToHost));
    const unsigned char *c = (const unsigned char *)row.data();
    for (size_t i = 0; i < (size_t)(N / 2) * 4; ++i) { fp ^= c[i]; fp *= 1099511628211ull; }
  }
  printf("LAB %-26s N=%d hop=%d G=%d: %8.3f ms  %7.2f Mframes/s  frac %.3f  fp=%016llx\n", LAB_NAME, N, HOP, LAB_G, ms, F / ms / 1e3,
         balg * F / (ms * 1e-3) / 8e12, fp);
#ifdef MX_TIMELINE
  unsigned long long tl[16][24];
  CK(hipMemcpyFromSymbol(tl, HIP_SYMBOL(mxlab::g_tl), sizeof tl));
  static const char *names[16] = {"frame assembly (loads/slide)", "pass 1", "T1 store", "barrier", "T1 read (+tw2, prev row flush)", "barrier",
                                  "pass 2", "T2 store", "barrier", "T2 read", "barrier", "pass 3 + split + sqrt", "pitch pick", "row scatter",
                                  "row out (barrier, flush, stores)", ""};
  const int NW = C::T / 64;
  printf("phase timeline, block %d frame +%d (cycles; mean over %d waves | wave 0 | last wave):\n", MX_TL_BLOCK, MX_TL_FRAME, NW);
  double tot = 0;
  for (int i = 0; i < 15; ++i) {
    double m = 0;
    for (int w = 0; w < NW; ++w) m += (double)(tl[w][i + 1] - tl[w][i]);
    m /= NW;
    tot += m;
    printf("  %2d %-34s %8.0f | %8llu | %8llu\n", i, names[i], m, tl[0][i + 1] - tl[0][i], tl[NW - 1][i + 1] - tl[NW - 1][i]);
  }
  printf("  frame total %.0f cycles (with the drains at the stamps)\n", tot);
#endif
  return 0;
}
