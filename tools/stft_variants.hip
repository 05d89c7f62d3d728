// tools/stft_variants.hip — tuning harness (not part of the product): times
// register/occupancy variants of the shipped STFT kernel template on one MI355X
// and a few VALU/LDS microbenchmarks that calibrate the roofline model in DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I melonix_amd/csrc tools/stft_variants.hip -o gpurun_out/stft_variants
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "stft_kernel_impl.h"

using namespace mx;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

template <class C, int MODE, int HOP, int WPE, bool NH, bool XM = true, int TR = 0, bool OS = false, bool DF = false, bool PF = false, bool DU = false>
float time_variant(const StftArgs &a, int reps, const char *name) {
  constexpr int N = C::N;
  const int64_t blocks = (a.count + a.frames_per_block - 1) / a.frames_per_block;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stft_kernel<C, MODE, HOP, WPE, NH, XM, TR, OS, DF, PF, DU>), dim3((unsigned)blocks), dim3(C::T), 0, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stft_kernel<C, MODE, HOP, WPE, NH, XM, TR, OS, DF, PF, DU>), dim3((unsigned)blocks), dim3(C::T), 0, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double fps = a.count / (ms * 1e-3);
  const double balg = 4.0 * a.hop + (a.mags ? 4.0 * (N / 2) : 0.0) + 8.0;
  // fingerprint of the outputs (sampled rows + the whole pitch track) so variants can be compared
  uint64_t fp = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t nb) { const unsigned char *c = (const unsigned char *)p; for (size_t i = 0; i < nb; ++i) { fp ^= c[i]; fp *= 1099511628211ull; } };
  if (a.mags) {
    std::vector<float> row(N / 2);
    for (int64_t f : {(int64_t)0, (int64_t)1, (int64_t)17, a.count / 3, a.count / 2 + 5, a.count - 2, a.count - 1}) {
      CK(hipMemcpy(row.data(), a.mags + (size_t)f * (N / 2), (N / 2) * 4, hipMemcpyDeviceToHost));
      mix(row.data(), (N / 2) * 4);
    }
  }
  if (a.pitch) {
    std::vector<mx_pitch> pt((size_t)a.count);
    CK(hipMemcpy(pt.data(), a.pitch, pt.size() * sizeof(mx_pitch), hipMemcpyDeviceToHost));
    for (auto &q : pt) { mix(&q.bin, 4); mix(&q.mag, 4); }
  }
  printf("%-30s E=%2d N=%5d G=%3d %8.3f ms  %8.2f Mframes/s  %7.1f GB/s alg (%.1f%% of 8 TB/s) fp=%016llx\n", name, C::E, N,
         a.frames_per_block, ms, fps / 1e6, fps * balg / 1e9, fps * balg / 8e12 * 100, (unsigned long long)fp);
  if (a.mags) CK(hipMemset(a.mags, 0xff, (size_t)a.count * (N / 2) * 4));
  if (a.pitch) CK(hipMemset(a.pitch, 0xff, (size_t)a.count * sizeof(mx_pitch)));
  fflush(stdout);
  return ms;
}

// ---- microbenchmarks ------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void valu_bench(float *out, int iters) {
  float2 a = make_float2(threadIdx.x * 1e-3f, 1.0f), b = make_float2(0.5f, 0.25f), c = make_float2(1.0001f, 0.9999f);
  float2 d = make_float2(0.3f, 0.7f), e = make_float2(0.1f, 0.2f), g = make_float2(0.9f, 1.1f);
  float2 h = make_float2(0.4f, 0.6f), k = make_float2(0.8f, 1.2f);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) {  // scalar adds (8 independent chains x2 components)
        a.x += c.x; a.y += c.y; b.x += c.x; b.y += c.y; d.x += c.x; d.y += c.y; e.x += c.x; e.y += c.y;
        g.x += c.x; g.y += c.y; h.x += c.x; h.y += c.y; k.x += c.x; k.y += c.y;
      } else if (KIND == 1) {  // packed adds
        asm volatile("v_pk_add_f32 %0, %0, %7\n v_pk_add_f32 %1, %1, %7\n v_pk_add_f32 %2, %2, %7\n v_pk_add_f32 %3, %3, %7\n"
                     "v_pk_add_f32 %4, %4, %7\n v_pk_add_f32 %5, %5, %7\n v_pk_add_f32 %6, %6, %7\n"
                     : "+v"(a), "+v"(b), "+v"(d), "+v"(e), "+v"(g), "+v"(h), "+v"(k) : "v"(c));
      } else if (KIND == 2) {  // scalar fma
        a.x = fmaf(a.x, c.x, c.y); a.y = fmaf(a.y, c.x, c.y); b.x = fmaf(b.x, c.x, c.y); b.y = fmaf(b.y, c.x, c.y);
        d.x = fmaf(d.x, c.x, c.y); d.y = fmaf(d.y, c.x, c.y); e.x = fmaf(e.x, c.x, c.y); e.y = fmaf(e.y, c.x, c.y);
        g.x = fmaf(g.x, c.x, c.y); g.y = fmaf(g.y, c.x, c.y); h.x = fmaf(h.x, c.x, c.y); h.y = fmaf(h.y, c.x, c.y);
        k.x = fmaf(k.x, c.x, c.y); k.y = fmaf(k.y, c.x, c.y);
      } else {  // packed fma
        asm volatile("v_pk_fma_f32 %0, %0, %7, %7\n v_pk_fma_f32 %1, %1, %7, %7\n v_pk_fma_f32 %2, %2, %7, %7\n"
                     "v_pk_fma_f32 %3, %3, %7, %7\n v_pk_fma_f32 %4, %4, %7, %7\n v_pk_fma_f32 %5, %5, %7, %7\n"
                     "v_pk_fma_f32 %6, %6, %7, %7\n"
                     : "+v"(a), "+v"(b), "+v"(d), "+v"(e), "+v"(g), "+v"(h), "+v"(k) : "v"(c));
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a.x + a.y + b.x + b.y + d.x + d.y + e.x + e.y + g.x + g.y + h.x + h.y + k.x + k.y;
}

template <int KIND>
void run_valu(const char *name, float *out) {
  const int blocks = 256 * 8, iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(valu_bench<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(valu_bench<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // 14 float lanes-ops per unrolled step (7 chains x 2 components), 8 steps per iteration
  const double ops = (double)blocks * 256 * iters * 8 * 14;
  printf("ubench %-18s %7.3f ms  %7.2f T lane-ops/s (fma counts as 1)\n", name, ms, ops / (ms * 1e-3) / 1e12);
}

// LDS write/read rate: each wave streams 16 KiB through its own LDS region
template <int KIND>
__global__ __launch_bounds__(64) void lds_bench(float *out, int iters) {
  __shared__ __attribute__((aligned(16))) float2 buf[2048];
  const int t = threadIdx.x;
  float2 v[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) v[r] = make_float2(t + r, r);
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0 || KIND == 2) {
#pragma unroll
      for (int r = 0; r < 32; ++r) buf[t + 64 * r] = v[r];
    }
    __syncthreads();
    if (KIND == 1 || KIND == 2) {
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        float2 q = buf[((t + i) & 63) + 64 * r];
        v[r].x += q.x;
        v[r].y += q.y;
      }
    }
    __syncthreads();
  }
  float acc = 0;
#pragma unroll
  for (int r = 0; r < 32; ++r) acc += v[r].x + v[r].y;
  out[blockIdx.x * 64 + t] = acc;
}

template <int KIND>
void run_lds(const char *name, float *out) {
  const int blocks = 256 * 8, iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(lds_bench<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(lds_bench<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)blocks * iters * 16384.0 * (KIND == 2 ? 2 : 1);
  printf("ubench %-18s %7.3f ms  %7.2f TB/s chip  (%.1f B/clk/CU @2.4GHz)\n", name, ms, bytes / (ms * 1e-3) / 1e12,
         bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main(int argc, char **argv) {
  const int minutes = argc > 1 ? atoi(argv[1]) : 60;
  constexpr int N = 4096;
  const int hop = 256;
  const int64_t n = (int64_t)minutes * 60 * 48000;
  const int64_t F = (n + hop - 1) / hop;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d MHz  LDS/block=%zu\n", prop.gcnArchName, prop.multiProcessorCount,
         prop.clockRate / 1000, prop.sharedMemPerBlock);

  // audio: deterministic pseudo-random + tone
  std::vector<float> h((size_t)n + 2 * MX_AUDIO_PAD, 0.f);
  uint32_t st = 12345;
  for (int64_t i = 0; i < n; ++i) {
    st = st * 1664525u + 1013904223u;
    h[(size_t)i + MX_AUDIO_PAD] = 0.4f * sinf(6.2831853f * 440.f * (float)(i % 48000) / 48000.f) + 0.1f * ((st >> 8) * (1.0f / 8388608.f) - 1.0f);
  }
  float *d_audio, *d_mags, *d_wtab, *d_out;
  mx_pitch *d_pitch;
  float2 *d_tw2, *d_tw3, *d_ub;
  CK(hipMalloc(&d_audio, h.size() * 4));
  CK(hipMemcpy(d_audio, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_mags, (size_t)F * (N / 2) * 4));
  CK(hipMalloc(&d_pitch, (size_t)F * sizeof(mx_pitch)));
  CK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
  auto wext = make_wext(fold_scale(N));
  auto wtab = make_wtab(N, hop, wext);
  using P32 = Plan<N, 32>;
  using P16 = Plan<N, 16>;
  auto up = [&](const std::vector<cpx_h> &v) { float2 *d; CK(hipMalloc(&d, v.size() * 8)); CK(hipMemcpy(d, v.data(), v.size() * 8, hipMemcpyHostToDevice)); return d; };
  CK(hipMalloc(&d_wtab, wtab.size() * 4));
  CK(hipMemcpy(d_wtab, wtab.data(), wtab.size() * 4, hipMemcpyHostToDevice));
  float2 *tw2_32 = up(make_tw2<P32>()), *tw3_32 = up(make_tw3<P32>()), *ub_32 = up(make_ubase<P32>());
  float2 *tw2_16 = up(make_tw2<P16>()), *tw3_16 = up(make_tw3<P16>()), *ub_16 = up(make_ubase<P16>());
  (void)d_tw2; (void)d_tw3; (void)d_ub;

  if (argc > 2 && atoi(argv[2])) {
  run_valu<0>("v_add_f32", d_out);
  run_valu<1>("v_pk_add_f32", d_out);
  run_valu<2>("v_fma_f32", d_out);
  run_valu<3>("v_pk_fma_f32", d_out);
  run_lds<0>("lds write b64", d_out);
  run_lds<1>("lds read b64", d_out);
  run_lds<2>("lds write+read b64", d_out);
  }

  StftArgs a{};
  a.audio = d_audio; a.n = n; a.wtab = d_wtab;
  a.decay = hop_decay(hop);
  a.hop = hop; a.first_frame = 0; a.count = F; a.kmin = 5; a.kmax = 150; a.mags = d_mags; a.pitch = d_pitch;
  StftArgs a32 = a, a16 = a;
  a32.tw2 = tw2_32; a32.tw3 = tw3_32; a32.ubase = ub_32;
  a16.tw2 = tw2_16; a16.tw3 = tw3_16; a16.ubase = ub_16;
  const int reps = 10;
  const bool ubench = argc > 2 && atoi(argv[2]);
  (void)ubench;
#ifdef ABLNAME
  a16.frames_per_block = 32;
  time_variant<P16, kBulkAligned, 256, 3, true, true, 2, true, true>(a16, reps, ABLNAME);
  time_variant<P16, kBulkAligned, 256, 3, true, true, 2, true, true, false, true>(a16, reps, ABLNAME " earlybar");
  a16.mags = nullptr;
  time_variant<P16, kBulkAligned, 256, 3, true, true, 2, true, true>(a16, reps, ABLNAME " pitch-only");
  return 0;
#endif
#ifdef OCC4
  for (int g : {16, 32}) {
    a16.frames_per_block = g;
    time_variant<P16, kBulkAligned, 256, 3, true, true, 2, true, true, false, true>(a16, reps, "shipped (3 waves/SIMD)");
    time_variant<P16, kBulkAligned, 256, 4, true, true, 3, false, false>(a16, reps, "wpe4 tw2 lds tw3 L2, out via image");
    time_variant<P16, kBulkAligned, 256, 4, true, true, 3, true, false>(a16, reps, "wpe4 tw2 lds tw3 L2, outsep");
    time_variant<P16, kBulkAligned, 256, 4, true, true, 3, true, true, false, true>(a16, reps, "wpe4 tw2 lds tw3 L2, defer earlybar");
    time_variant<P16, kBulkAligned, 256, 4, true, true, 2, false, false>(a16, reps, "wpe4 tw3 reg, out via image");
    time_variant<P16, kBulkAligned, 256, 3, true, true, 3, false, false>(a16, reps, "wpe3 tw3 L2, out via image");
  }
  return 0;
#endif
#ifdef GSWEEP
  for (int rep = 0; rep < 2; ++rep)
  for (int g : {24, 32, 40, 48, 64, 80, 96, 128, 192}) {
    a16.frames_per_block = g;
    time_variant<P16, kBulkAligned, 256, 3, true, true, 2, true, true, false, true>(a16, reps, "shipped E=16");
  }
  return 0;
#endif
#ifdef P32SWEEP
  for (int g : {32, 64}) {
    a32.frames_per_block = a16.frames_per_block = g;
    time_variant<P16, kBulkAligned, 256, 3, true, true, 2, true, true, false, true>(a16, reps, "shipped E=16");
    time_variant<P32, kBulkAligned, 256, 2, true, true, 0, false, false>(a32, reps, "E=32 tw L2, out via image");
    time_variant<P32, kBulkAligned, 256, 2, true, true, 3, false, false>(a32, reps, "E=32 tw2 lds, out via image");
    time_variant<P32, kBulkAligned, 256, 2, true, true, 3, true, false>(a32, reps, "E=32 tw2 lds, outsep");
    time_variant<P32, kBulkAligned, 256, 2, true, true, 3, true, true>(a32, reps, "E=32 tw2 lds, defer");
    time_variant<P32, kBulkAligned, 256, 2, true, true, 3, true, true, false, true>(a32, reps, "E=32 tw2 lds, defer earlybar");
    time_variant<P32, kBulkAligned, 256, 2, true, true, 2, true, true>(a32, reps, "E=32 tw2 lds tw3 reg, defer");
  }
  return 0;
#endif
  for (int g : {8, 16, 32}) {
    a32.frames_per_block = a16.frames_per_block = g;
    time_variant<P16, kBulkAligned, 256, 3, true, true, 2, true, true>(a16, reps, "shipped: tw2lds defer wpe3");
    time_variant<P16, kBulkAligned, 256, 3, true, true, 2, true, true, false, true>(a16, reps, "early barrier wpe3");
  }
  a16.frames_per_block = 16;
  time_variant<P16, kBulkAligned, 0, 3, true, true, 2, true, true>(a16, reps, "direct: tw2lds defer");
  time_variant<P16, kBulkAligned, 0, 3, true, true, 2, true, true, false, true>(a16, reps, "direct: early barrier");
  a16.mags = nullptr;
  time_variant<P16, kBulkAligned, 256, 3, true, true, 2, true, true>(a16, reps, "pitch-only shipped");
  time_variant<P16, kBulkAligned, 256, 3, true, true, 2, true, true, false, true>(a16, reps, "pitch-only early barrier");
  return 0;
}
