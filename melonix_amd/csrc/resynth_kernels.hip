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
// is independent: one workgroup per step.  resynth_kernel_v stages the step's
// grain in LDS with coalesced 16-byte loads (each source sample leaves HBM/L2
// once), gathers the two interpolation taps from LDS (one ds_read2_b32), and
// writes the PCM run as aligned 16-byte vectors (8 outputs per thread), with
// scalar head/tail elements up to the first/after the last 8-aligned output.
// resynth_kernel (scalar) remains for output pointers that are not 16-byte aligned.
#include <hip/hip_runtime.h>

#include "kernels.h"

#pragma clang fp contract(off)

namespace mx {

namespace {

constexpr int kResynthThreads = 128;  // a step emits ~1260 samples = ~158 groups of 8: 128 threads keep the lanes busy (measured 64/128/192/256)

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

constexpr int kMaxGrain = 2304;  // good grains are 751..2249 samples (app.cpp:164-166); longer ones take the scalar kernel

__device__ __forceinline__ float lerp_tap(const float *lg, int i, float rate) {
  const float x = (float)i * rate;       // app.cpp:336
  const float ip = __builtin_truncf(x);  // std::modf
  const float f = x - ip;
  const int idx = (int)ip;
  return (1.f - f) * lg[idx] + f * lg[idx + 1];  // lg[L] holds nextGrainFirstSample (app.cpp:340-341)
}

__global__ __launch_bounds__(kResynthThreads) void resynth_kernel_v(const ResynthArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kMaxGrain + 8];
  const mx_step st = a.steps[blockIdx.x];
  const int L = st.grain_len, sz = st.sz, tid = threadIdx.x;
  const float *g = a.audio + MX_AUDIO_PAD + st.grain_start;
  if (L > kMaxGrain) {
    // a fallback grain (app.cpp:198-228) beyond the LDS stage: straight from L2, the arithmetic of resynth_kernel
    // (block-uniform branch; rare — one launch serves both kinds of steps)
    float *of = a.pcm_f32 ? a.pcm_f32 + st.out_offset : nullptr;
    int16_t *oi = a.pcm_i16 ? a.pcm_i16 + st.out_offset : nullptr;
    for (int i = tid; i < sz; i += kResynthThreads) {
      const float x = (float)i * st.rate;
      const float ip = __builtin_truncf(x);
      const float f = x - ip;
      const int idx = (int)ip;
      const float a1 = (idx + 1 < L) ? g[idx + 1] : st.next_first;
      const float v = (1.f - f) * g[idx] + f * a1;
      if (of) of[i] = v;
      if (oi) oi[i] = (int16_t)((double)v * 32767.);
    }
    return;
  }
  // stage [g - shift, g + L] with aligned float4 loads (the padded image makes g - 3 readable)
  const int shift = (int)((reinterpret_cast<uintptr_t>(g) >> 2) & 3);
  const float4 *ga = reinterpret_cast<const float4 *>(g - shift);
  float4 *l4 = reinterpret_cast<float4 *>(lds);
  for (int j = tid; j < (L + shift + 4) / 4; j += kResynthThreads) l4[j] = ga[j];
  __syncthreads();
  float *lg = lds + shift;
  if (tid == 0) lg[L] = st.next_first;  // may differ from wav[grain_start + L] when grains repeat or skip
  __syncthreads();

  float *of = a.pcm_f32 ? a.pcm_f32 + st.out_offset : nullptr;
  int16_t *oi = a.pcm_i16 ? a.pcm_i16 + st.out_offset : nullptr;
  const float rate = st.rate;
  // outputs [head, head + 8*nv) are 8-aligned in the PCM stream
  const int head = (int)((8 - (st.out_offset & 7)) & 7) < sz ? (int)((8 - (st.out_offset & 7)) & 7) : sz;
  const int nv = (sz - head) / 8;
  for (int q = tid; q < nv; q += kResynthThreads) {
    const int i0 = head + 8 * q;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = lerp_tap(lg, i0 + k, rate);
    if (of) {
      float4 *o4 = reinterpret_cast<float4 *>(of + i0);
      o4[0] = make_float4(v[0], v[1], v[2], v[3]);
      o4[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (oi) {
      unsigned w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned lo = (unsigned)(unsigned short)(int16_t)((double)v[2 * k] * 32767.);  // app.cpp:1212
        const unsigned hi = (unsigned)(unsigned short)(int16_t)((double)v[2 * k + 1] * 32767.);
        w[k] = lo | (hi << 16);
      }
      *reinterpret_cast<uint4 *>(oi + i0) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  // scalar head and tail
  const int tail0 = head + 8 * nv;
  for (int i = tid; i < head + (sz - tail0); i += kResynthThreads) {
    const int ii = i < head ? i : tail0 + (i - head);
    const float v = lerp_tap(lg, ii, rate);
    if (of) of[ii] = v;
    if (oi) oi[ii] = (int16_t)((double)v * 32767.);
  }
}

// zc bitmaps.  A wavefront owns 64 consecutive bitmap words (4096 samples): every sample is loaded once
// (64 coalesced 256-byte reads), two ballots turn each row into the sign words
//   A = { !(x >= 0) }  (app.cpp:175 "wav[idx-j] >= 0 -> reject")   B = { !(x < 0) }  (app.cpp:177)
// which lane w keeps for row w; the predicates are then bit-parallel on 64 samples per lane:
//   zc_k[i] = AND_{j<k} A[i-j]  &  AND_{j<k} B[i+1+j],   k = 7 and 3,
// with the carries from the neighbouring words (previous lane's A, next lane's B; the rows before and
// after the wavefront's block come from two extra reads).  The reference's index bounds (idx >= k,
// idx < n-k-1) are masks, so the pads around the audio never decide a bit.
__device__ __forceinline__ uint64_t bit_range(int64_t base, int64_t lo, int64_t hi) {  // bits i: lo <= base+i < hi
  const int64_t s = lo - base > 0 ? lo - base : 0;
  const int64_t e = hi - base < 64 ? hi - base : 64;
  if (e <= s) return 0ull;
  const uint64_t upto_e = e >= 64 ? ~0ull : ((1ull << e) - 1ull);
  return upto_e & ~((1ull << s) - 1ull);
}

__global__ __launch_bounds__(256) void zc_kernel(const float *__restrict__ wav /* unpadded base */,
                                                 int64_t n, uint64_t *__restrict__ zc7,
                                                 uint64_t *__restrict__ zc3) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;  // first word of this wavefront
  const int64_t nwords = (n + 63) >> 6;
  if (w0 >= nwords) return;  // wavefront-uniform
  // rows -1 .. 64 of the block; MX_AUDIO_PAD readable samples lie on either side of the audio
  const float *p = wav + w0 * 64 + lane;
  const float xp = p[-64], xn = p[64 * 64];
  const uint64_t Aprev = __ballot(!(xp >= 0.f)), Bnext = __ballot(!(xn < 0.f));
  uint64_t A = 0, B = 0;
#pragma unroll 16
  for (int w = 0; w < 64; ++w) {
    const float x = p[64 * w];
    const uint64_t a = __ballot(!(x >= 0.f)), b = __ballot(!(x < 0.f));
    if (lane == w) {
      A = a;
      B = b;
    }
  }
  uint64_t Ap = ((uint64_t)(unsigned)__shfl_up((int)(A >> 32), 1) << 32) | (unsigned)__shfl_up((int)A, 1);
  uint64_t Bn = ((uint64_t)(unsigned)__shfl_down((int)(B >> 32), 1) << 32) | (unsigned)__shfl_down((int)B, 1);
  if (lane == 0) Ap = Aprev;
  if (lane == 63) Bn = Bnext;
  uint64_t ra3 = A, ra7 = A, rb3 = ~0ull, rb7 = ~0ull;
#pragma unroll
  for (int j = 1; j <= 7; ++j) {
    const uint64_t bj = (B >> j) | (Bn << (64 - j));  // B[i + j]
    rb7 &= bj;
    if (j <= 3) rb3 &= bj;
    if (j <= 6) {
      const uint64_t aj = (A << j) | (Ap >> (64 - j));  // A[i - j]
      ra7 &= aj;
      if (j <= 2) ra3 &= aj;
    }
  }
  const int64_t wi = w0 + lane;
  if (wi < nwords) {
    zc7[wi] = ra7 & rb7 & bit_range(wi * 64, 7, n - 7 - 1);
    zc3[wi] = ra3 & rb3 & bit_range(wi * 64, 3, n - 3 - 1);
  }
}

// Samples past the last step's run: the zeros of the terminating process() call(s) (app.cpp:303-309) are not steps.
// The last step record says where the steps end, so a caller holding only device pointers needs no host-side sum.
__global__ __launch_bounds__(256) void resynth_tail_kernel(const ResynthArgs a) {
  const mx_step last = a.steps[a.nsteps - 1];
  const int64_t covered = last.out_offset + (int64_t)last.sz;
  for (int64_t i = covered + (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.nsamples; i += (int64_t)gridDim.x * 256) {
    if (a.pcm_f32) a.pcm_f32[i] = 0.f;
    if (a.pcm_i16) a.pcm_i16[i] = 0;
  }
}

}  // namespace

hipError_t launch_resynth(const ResynthArgs &a, hipStream_t s) {
  if (a.nsteps > 0x7fffffffLL) return hipErrorInvalidValue;
  if (a.nsteps > 0) {
    hipLaunchKernelGGL(resynth_tail_kernel, dim3(64), dim3(256), 0, s, a);
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.pcm_f32) | reinterpret_cast<uintptr_t>(a.pcm_i16) |
                           reinterpret_cast<uintptr_t>(a.audio)) & 15) == 0;
    if (aligned) {
      hipLaunchKernelGGL(resynth_kernel_v, dim3((unsigned)a.nsteps), dim3(kResynthThreads), 0, s, a);
    } else {
      hipLaunchKernelGGL(resynth_kernel, dim3((unsigned)a.nsteps), dim3(kResynthThreads), 0, s, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_zc_bitmaps(const float *audio_padded, int64_t n, uint64_t *zc7, uint64_t *zc3,
                             hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int64_t blocks = (((n + 63) >> 6) + 255) / 256;  // 4 wavefronts x 64 words per workgroup
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  static_assert(MX_AUDIO_PAD >= 64 * 65, "zc_kernel reads one block row past either end of the audio");
  hipLaunchKernelGGL(zc_kernel, dim3((unsigned)blocks), dim3(256), 0, s, audio_padded + MX_AUDIO_PAD, n,
                     zc7, zc3);
  return hipGetLastError();
}

}  // namespace mx
