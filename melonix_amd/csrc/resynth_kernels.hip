// resynth_kernels.hip — gfx950 kernels for the granular pitch-shift resynthesis
// (reference App::process, app.cpp:332-343, and the float->int16 step of
// App::exportWav, app.cpp:1209-1212) plus the zero-crossing predicate bitmaps
// of the grain scan (app.cpp:167-181, 202-216).
//
// This file is built with -ffp-contract=off: the reference evaluates
//   (1.f - f) * g[idx] + f * g[idx+1]
// as separate binary32 multiplies and an add (g++ -O2 on baseline x86-64 has no
// FMA), and the PCM must match bit for bit, so no v_fma/v_fmac may be formed
// here (tests/test_build.py greps the ISA for it).
//
// Parallelisation: the export loop is serial only in its scalar cursor
// (host_logic.cpp: build_schedule).  Given the schedule, every output sample
// is independent: one workgroup per step stages nothing but reads the step's
// grain (contiguous source samples, L2/HBM-coalesced) and writes a contiguous
// run of the PCM stream at out_offset.
#include <hip/hip_runtime.h>

#include "kernels.h"

#pragma clang fp contract(off)

namespace mx {

namespace {

constexpr int kResynthThreads = 256;

__global__ __launch_bounds__(kResynthThreads) void resynth_kernel(const ResynthArgs a) {
  const mx_step st = a.steps[blockIdx.x];
  const float *g = a.audio + MX_AUDIO_PAD + st.grain_start;  // grain = wav[grain_start, grain_start+L)
  const int L = st.grain_len;
  const float rate = st.rate;
  const float next = st.next_first;
  float *of = a.pcm_f32 ? a.pcm_f32 + st.out_offset : nullptr;
  int16_t *oi = a.pcm_i16 ? a.pcm_i16 + st.out_offset : nullptr;
  for (int i = threadIdx.x; i < st.sz; i += kResynthThreads) {
    const float x = (float)i * rate;      // i * rate + bias, bias == 0.f (app.cpp:336, app.hpp:66)
    const float ip = __builtin_truncf(x); // std::modf integral part
    const float f = x - ip;               // std::modf fractional part (exact)
    const int idx = (int)ip;
    const float a0 = g[idx];
    const float a1 = (idx + 1 < L) ? g[idx + 1] : next;
    const float v = (1.f - f) * a0 + f * a1;  // app.cpp:340-341, no contraction
    if (of) of[i] = v;
    if (oi) oi[i] = (int16_t)((double)v * 32767.);  // app.cpp:1212: double multiply, truncation
  }
}

// zc bitmaps: one thread per sample, one 64-bit ballot per wavefront.
__global__ __launch_bounds__(256) void zc_kernel(const float *__restrict__ wav /* unpadded base */,
                                                 int64_t n, uint64_t *__restrict__ zc7,
                                                 uint64_t *__restrict__ zc3) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool z7 = false, z3 = false;
  if (i + 1 < n) {
    // padded image: indices -PAD..n+PAD-1 are readable; the reference's bounds
    // (idx >= k, idx < n-k-1) are applied explicitly so the pads never decide.
    bool a3 = true, a7 = true;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      // app.cpp:175-178: "wav[idx-j] >= 0 -> reject", "wav[idx+1+j] < 0 -> reject"
      const bool ok = !(wav[i - j] >= 0.f) && !(wav[i + 1 + j] < 0.f);
      a7 = a7 && ok;
      if (j < 3) a3 = a3 && ok;
    }
    z7 = a7 && (i >= 7) && (i < n - 7 - 1);
    z3 = a3 && (i >= 3) && (i < n - 3 - 1);
  }
  const unsigned long long b7 = __ballot(z7);
  const unsigned long long b3 = __ballot(z3);
  if ((threadIdx.x & 63) == 0 && i < n) {
    zc7[i >> 6] = b7;
    zc3[i >> 6] = b3;
  }
}

}  // namespace

hipError_t launch_resynth(const ResynthArgs &a, hipStream_t s) {
  if (a.nsteps > 0x7fffffffLL) return hipErrorInvalidValue;
  // The terminating process() call's 1500 zeros (app.cpp:303-309) are not a step:
  // capi.cpp clears the tail [sum(sz), nsamples) with hipMemsetAsync before this launch.
  if (a.nsteps > 0) {
    hipLaunchKernelGGL(resynth_kernel, dim3((unsigned)a.nsteps), dim3(kResynthThreads), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_zc_bitmaps(const float *audio_padded, int64_t n, uint64_t *zc7, uint64_t *zc3,
                             hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(zc_kernel, dim3((unsigned)blocks), dim3(256), 0, s, audio_padded + MX_AUDIO_PAD, n,
                     zc7, zc3);
  return hipGetLastError();
}

}  // namespace mx
