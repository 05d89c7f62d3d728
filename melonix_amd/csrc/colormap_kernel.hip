// colormap_kernel.hip — the two small UI-feeding kernels next to the STFT: the waveform min/max
// pyramid (App::calcPicks, app.cpp:347-378; see below) and SpecCache::populateTex's colormap (reference spec-cache.cpp:77-96) on the
// GPU: magnitude rows (HBM/L2-resident, straight out of the STFT kernel) -> RGB8 texture rows, so a
// batch of columns leaves the device as 3 bytes per bin instead of 4 and the UI thread's per-column
// loop disappears.  Arithmetic follows the reference operation by operation:
//   v = clamp(mag*k, 0, 255)                      binary32
//   v < 85 (=255/3):   (uchar)v, 0, 0
//   v < 170 (=2*255/3): a = (v-85)/85 [binary32] * 3.141592 / 2 [binary64];
//                       (uchar)(v*cos(a)), (uchar)(v*sin(a)), 0   [binary64 products, truncation]
//   else:               l = (uchar)((v-170)*3); l, (uchar)v, l
// Built with -ffp-contract=off.  Four texels per thread -> three dword stores (coalesced).
#include <hip/hip_runtime.h>

#include "colormap_core.h"
#include "kernels.h"

namespace mx {
namespace {

__global__ __launch_bounds__(256) void colormap_kernel(const float4 *__restrict__ mags, uint32_t *__restrict__ rgb,
                                                       int64_t n4, float k) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 m = mags[i];
  uint32_t w[3];
  texel4(m.x, m.y, m.z, m.w, k, w);
  uint32_t *o = rgb + 3 * i;
  o[0] = w[0];
  o[1] = w[1];
  o[2] = w[2];
}

// ---- waveform min/max pyramid (App::calcPicks, app.cpp:347-378) -------------------------------
// Level l holds floor(n / 2^(l+1)) {min,max} pairs over blocks of 2^(l+1) samples.  The comparisons
// are the reference's std::min / std::max forms ((b < a) ? b : a and (a < b) ? b : a), not v_min/v_max,
// so signed zeros and NaNs come out bit-identical, and every pair is combined from the same two pairs of
// the level below as in the reference's loops (app.cpp:363-376), so the tree order is the reference's too.
//
// picks_fused: one workgroup owns 4096 consecutive samples and produces their part of levels 0..11 from
// ONE read of the audio — levels 0-1 in registers (4 samples per thread), 2-7 across the lanes of a
// wavefront, 8-11 across the sixteen wavefronts through LDS.  picks_tail then builds the (tiny) upper levels,
// one workgroup walking level after level.  Two launches instead of one per level, and the lower levels
// are never read back: 2.07 GB of traffic for 60 min of audio instead of 3.45 GB.
struct PicksArgs {
  const float *wav;       // unpadded base; MX_AUDIO_PAD readable samples follow the audio
  int64_t n;
  float2 *level[64];      // start of each level inside the caller's buffer
  int nlevels;
};
constexpr int kFusedLevels = 12;  // 4 samples/thread (2) + 64 lanes (6) + 16 wavefronts (4)

__device__ __forceinline__ float2 pick2(float a, float b) { return make_float2(b < a ? b : a, a < b ? b : a); }
__device__ __forceinline__ float2 pick_up(float2 p0, float2 p1) {  // pairs 2i, 2i+1 -> pair i of the next level
  return make_float2(p1.x < p0.x ? p1.x : p0.x, p0.y < p1.y ? p1.y : p0.y);
}

__global__ __launch_bounds__(1024) void picks_fused(const PicksArgs a) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t blk = blockIdx.x;
  // four consecutive samples per thread: every load and every level-0/1 store is lane-contiguous
  const float4 s0 = reinterpret_cast<const float4 *>(a.wav + blk * 4096)[t];
  // pair i of level l exists iff i < n >> (l+1); it then only depends on samples below n
  auto cnt = [&](int l) { return a.n >> (l + 1); };
  const float2 pa = pick2(s0.x, s0.y), pb = pick2(s0.z, s0.w);
  float2 p = pick_up(pa, pb);
  {
    const int64_t i0 = blk * 2048 + t * 2;
    if (a.nlevels > 0) {
      if (i0 + 2 <= cnt(0)) *reinterpret_cast<float4 *>(a.level[0] + i0) = make_float4(pa.x, pa.y, pb.x, pb.y);
      else if (i0 < cnt(0)) a.level[0][i0] = pa;
    }
    if (a.nlevels > 1 && blk * 1024 + t < cnt(1)) a.level[1][blk * 1024 + t] = p;
  }
  // levels 2..7: lanes 2^k apart (every lane computes, the lanes that own a pair write it)
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int d = 1 << k;
    const float2 q = make_float2(__shfl_down(p.x, d), __shfl_down(p.y, d));
    p = pick_up(p, q);
    const int lvl = 2 + k;
    const int64_t idx = (blk * 1024 + t) >> (k + 1);
    if (lvl < a.nlevels && (lane & (2 * d - 1)) == 0 && idx < cnt(lvl)) a.level[lvl][idx] = p;
  }
  // levels 8..11: the sixteen wavefronts' level-7 pairs, one more wavefront-level tree
  __shared__ float2 w7[16];
  if (lane == 0) w7[wave] = p;
  __syncthreads();
  if (t < 16) {
    float2 u = w7[t];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int d = 1 << k;
      const float2 q = make_float2(__shfl_down(u.x, d), __shfl_down(u.y, d));
      u = pick_up(u, q);
      const int lvl = 8 + k;
      const int64_t idx = (blk * 16 + t) >> (k + 1);
      if (lvl < a.nlevels && (t & (2 * d - 1)) == 0 && idx < cnt(lvl)) a.level[lvl][idx] = u;
    }
  }
}

// Upper levels (from `first` on): a single workgroup, level after level; each level is a few thousand
// pairs at most.  The workgroup's own stores are visible to it after the barrier.
__global__ __launch_bounds__(1024) void picks_tail(const PicksArgs a, int first) {
  for (int lvl = first; lvl < a.nlevels; ++lvl) {
    const int64_t cnt = a.n >> (lvl + 1);
    const float2 *prev = a.level[lvl - 1];
    float2 *cur = a.level[lvl];
    for (int64_t i = threadIdx.x; i < cnt; i += 1024) cur[i] = pick_up(prev[2 * i], prev[2 * i + 1]);
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace

hipError_t launch_picks(const float *audio_padded, int64_t n, float *d_out, int64_t *counts, int *nlevels, hipStream_t s) {
  *nlevels = 0;
  if (n <= 2) return hipSuccess;  // app.cpp:352: no level at all
  PicksArgs a{};
  a.wav = audio_padded + MX_AUDIO_PAD;
  a.n = n;
  float2 *cur = reinterpret_cast<float2 *>(d_out);
  int lvl = 0;
  for (; lvl < 62 && n > (int64_t)(1ll << (lvl + 1)); ++lvl) {  // app.cpp:352, :366
    counts[lvl] = n >> (lvl + 1);
    a.level[lvl] = cur;
    cur += counts[lvl];
  }
  a.nlevels = lvl;
  *nlevels = lvl;
  static_assert(MX_AUDIO_PAD >= 4096, "picks_fused reads whole 4096-sample blocks");
  const int64_t blocks = (n + 4095) / 4096;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(picks_fused, dim3((unsigned)blocks), dim3(1024), 0, s, a);
  if (lvl > kFusedLevels) hipLaunchKernelGGL(picks_tail, dim3(1), dim3(1024), 0, s, a, kFusedLevels);
  return hipGetLastError();
}

hipError_t launch_colormap(const float *mags, uint8_t *rgb, int64_t nbins_total, float k, hipStream_t s) {
  if (nbins_total <= 0) return hipSuccess;
  if (nbins_total % 4) return hipErrorInvalidValue;  // rows are N/2 bins, N/2 % 4 == 0
  const int64_t n4 = nbins_total / 4;
  const int64_t blocks = (n4 + 255) / 256;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(colormap_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4 *>(mags),
                     reinterpret_cast<uint32_t *>(rgb), n4, k);
  return hipGetLastError();
}

}  // namespace mx
