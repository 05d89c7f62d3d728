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
// so signed zeros and NaNs come out bit-identical.
__global__ __launch_bounds__(256) void picks_level0(const float2 *__restrict__ wav2, float2 *__restrict__ out, int64_t cnt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cnt) return;
  const float2 s = wav2[i];  // samples 2i, 2i+1 (the padded image keeps this 8-byte aligned)
  out[i] = make_float2(s.y < s.x ? s.y : s.x, s.x < s.y ? s.y : s.x);
}
__global__ __launch_bounds__(256) void picks_levelN(const float4 *__restrict__ prev, float2 *__restrict__ out, int64_t cnt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cnt) return;
  const float4 p = prev[i];  // {min0, max0, min1, max1} = previous level's pairs 2i, 2i+1
  out[i] = make_float2(p.z < p.x ? p.z : p.x, p.y < p.w ? p.w : p.y);
}

}  // namespace

hipError_t launch_picks(const float *audio_padded, int64_t n, float *d_out, int64_t *counts, int *nlevels, hipStream_t s) {
  *nlevels = 0;
  int lvl = 0;
  if (n <= (int64_t)(1ll << (lvl + 1))) return hipSuccess;  // app.cpp:352
  const float2 *wav2 = reinterpret_cast<const float2 *>(audio_padded + MX_AUDIO_PAD);
  float2 *cur = reinterpret_cast<float2 *>(d_out);
  int64_t cnt = n >> 1;
  hipLaunchKernelGGL(picks_level0, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, wav2, cur, cnt);
  counts[0] = cnt;
  const float2 *prev = cur;
  cur += cnt;
  for (;;) {
    ++lvl;
    if (n <= (int64_t)(1ll << (lvl + 1)) || lvl >= 62) break;  // app.cpp:366
    cnt = n >> (lvl + 1);
    hipLaunchKernelGGL(picks_levelN, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4 *>(prev), cur, cnt);
    counts[lvl] = cnt;
    prev = cur;
    cur += cnt;
  }
  *nlevels = lvl;
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
