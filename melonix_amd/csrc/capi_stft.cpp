// capi_stft.cpp — STFT + pitch entry points: bulk hop mode, ranges mode, host-staged and texel variants.
// One unit of the C-ABI implementation behind include/melonix_amd.h (see capi_internal.h).  There is no CPU compute path:
// every transform entry point needs a live gfx950 device and fails with MX_ERR_DEVICE otherwise.
#include "capi_internal.h"

using namespace mx;

extern "C" {

int mx_stft_run_length(int N, int hop, int64_t count) {
  return mx_guard([&]() -> int {
    if ((N != 4096 && N != 16384 && N != 32768) || hop <= 0 || count < 0) return fail(MX_ERR_INVALID, "bad argument");
    return default_frames_per_block(N, (hop & 1) ? kBulkAny : kBulkAligned, hop, count);
  });
}

// ---- STFT ---------------------------------------------------------------------
void mx_pitch_band(int N, int sampleRate, int *kmin, int *kmax) {
  mx_guard_void([&] {
    // notes 24..84 of the default view (app.hpp:45-46): f = 55*2^((note-24)/12), bin = f*N/sr (app.cpp:499-516)
    const double lo = 55.0 * N / sampleRate, hi = 1760.0 * N / sampleRate;
    int a = (int)lo;
    if ((double)a < lo) ++a;
    if (kmin) *kmin = a;
    if (kmax) *kmax = (int)hi;
  });
}

double mx_bin_note(int bin, int N, int sampleRate) {
  return mx_guard_or<double>(std::nan(""), [&]() -> double {
    if (bin <= 0 || N <= 0 || sampleRate <= 0) return -HUGE_VAL;
    return 24. + 12. * std::log2((double)bin * sampleRate / N / 55.);
  });
}
double mx_note_bin(double note, int N, int sampleRate) {
  return mx_guard_or<double>(std::nan(""), [&]() -> double {
    if (N <= 0 || sampleRate <= 0) return 0.;
    return 55. * std::pow(2., (note - 24.) / 12.) * N / sampleRate;  // app.cpp:498
  });
}

int64_t mx_frame_count(int64_t n, int hop) {
  return mx_guard([&]() -> int64_t {
    return hop > 0 && n >= 0 ? (n + hop - 1) / hop : -1;
  });
}

static int stft_hop_dev_run(mx_ctx *ctx, const mx_audio *a, int N, int hop, int64_t first_frame, int64_t count,
                            int kmin, int kmax, float *d_mags, mx_pitch *d_pitch, int run_length) {
  int rc = check_common(ctx, a, N, count, kmin, kmax);
  if (rc) return rc;
  if (hop <= 0 || hop > MX_AUDIO_PAD) return fail(MX_ERR_INVALID, "hop %d out of range [1,%d]", hop, MX_AUDIO_PAD);
  if (first_frame < 0 || (count > 0 && (first_frame + count - 1) * (int64_t)hop >= a->n))
    return fail(MX_ERR_INVALID, "frames [%lld,%lld) exceed ceil(n/hop)", (long long)first_frame,
                (long long)(first_frame + count));
  return stft_launch(ctx, a, N, (hop % 2 == 0) ? kBulkAligned : kBulkAny, hop, first_frame, nullptr, count, kmin,
                     kmax, d_mags, d_pitch, nullptr, 0.f, run_length);
}

int mx_stft_hop_dev(mx_ctx *ctx, const mx_audio *a, int N, int hop, int64_t first_frame, int64_t count,
                    int kmin, int kmax, float *d_mags, mx_pitch *d_pitch) {
  return mx_guard([&]() -> int {
    return stft_hop_dev_run(ctx, a, N, hop, first_frame, count, kmin, kmax, d_mags, d_pitch, 0);
  });
}

int mx_stft_ranges_dev(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *d_ranges, int64_t count, int kmin,
                       int kmax, float *d_mags, mx_pitch *d_pitch) {
  return mx_guard([&]() -> int {
    int rc = check_common(ctx, a, N, count, kmin, kmax);
    if (rc) return rc;
    if (count > 0 && !d_ranges) return fail(MX_ERR_INVALID, "ranges is null");
    return stft_launch(ctx, a, N, kRanges, 0, 0, d_ranges, count, kmin, kmax, d_mags, d_pitch, nullptr, 0.f);
  });
}

static int stft_host_common(mx_ctx *ctx, const mx_audio *a, int N, bool ranges_mode, int hop, int64_t first_frame,
                            const int32_t *ranges, int64_t count, int kmin, int kmax, float *mags_out,
                            mx_pitch *pitch_out) {
  int rc = check_common(ctx, a, N, count, kmin, kmax);
  if (rc) return rc;
  if (count == 0) return MX_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  const int64_t chunk = std::min<int64_t>(count, chunk_frames(N));
  float *d_mags = nullptr;
  mx_pitch *d_pitch = nullptr;
  int32_t *d_ranges = nullptr;
  const size_t row = (size_t)(N / 2);
  hipError_t e = hipSuccess;
  std::lock_guard<std::mutex> slk(ctx->stage_mu);
  if (mags_out) e = stage_get(ctx, 0, (size_t)chunk * row * sizeof(float), (void **)&d_mags);
  if (e == hipSuccess && pitch_out) e = stage_get(ctx, 1, (size_t)chunk * sizeof(mx_pitch), (void **)&d_pitch);
  if (e == hipSuccess && ranges_mode) e = stage_get(ctx, 2, (size_t)chunk * 2 * sizeof(int32_t), (void **)&d_ranges);
  if (e != hipSuccess) return fail(MX_ERR_NOMEM, "device staging buffers: %s", hipGetErrorString(e));
  rc = MX_OK;
  // one run length for the whole call, whatever its staging chunks are (chunks are multiples of 32 frames): the rows
  // are those of a single launch of `count` frames
  const int run = ranges_mode ? 0 : default_frames_per_block(N, (hop % 2 == 0) ? kBulkAligned : kBulkAny, hop, count);
  for (int64_t done = 0; done < count && rc == MX_OK; done += chunk) {
    const int64_t c = std::min(chunk, count - done);
    if (ranges_mode) {
      e = hipMemcpyAsync(d_ranges, ranges + 2 * done, (size_t)c * 2 * sizeof(int32_t), hipMemcpyHostToDevice,
                         ctx->stream);
      if (e != hipSuccess) { rc = fail(MX_ERR_DEVICE, "ranges upload: %s", hipGetErrorString(e)); break; }
      rc = mx_stft_ranges_dev(ctx, a, N, d_ranges, c, kmin, kmax, d_mags, d_pitch);
    } else {
      rc = stft_hop_dev_run(ctx, a, N, hop, first_frame + done, c, kmin, kmax, d_mags, d_pitch, run);
    }
    if (rc) break;
    if (mags_out)
      e = hipMemcpyAsync(mags_out + (size_t)done * row, d_mags, (size_t)c * row * sizeof(float),
                         hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && pitch_out)
      e = hipMemcpyAsync(pitch_out + done, d_pitch, (size_t)c * sizeof(mx_pitch), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(MX_ERR_DEVICE, "result download: %s", hipGetErrorString(e));
  }
  stage_trim(ctx);
  return rc;
}

int mx_stft_hop(mx_ctx *ctx, const mx_audio *a, int N, int hop, int64_t first_frame, int64_t count, int kmin,
                int kmax, float *mags_out, mx_pitch *pitch_out) {
  return mx_guard([&]() -> int {
    if (hop <= 0) return fail(MX_ERR_INVALID, "hop must be positive");
    return stft_host_common(ctx, a, N, false, hop, first_frame, nullptr, count, kmin, kmax, mags_out, pitch_out);
  });
}

int mx_stft_ranges(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *ranges, int64_t count, int kmin, int kmax,
                   float *mags_out, mx_pitch *pitch_out) {
  return mx_guard([&]() -> int {
    if (count > 0 && !ranges) return fail(MX_ERR_INVALID, "ranges is null");
    return stft_host_common(ctx, a, N, true, 0, 0, ranges, count, kmin, kmax, mags_out, pitch_out);
  });
}

int mx_stft_ranges_rgb_dev(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *d_ranges, int64_t count, float k,
                           float *d_mags, uint8_t *d_rgb) {
  return mx_guard([&]() -> int {
    int kmin = -1, kmax = -1;
    int rc = check_common(ctx, a, N, count, kmin, kmax);
    if (rc) return rc;
    if (count > 0 && (!d_ranges || !d_rgb)) return fail(MX_ERR_INVALID, "ranges / rgb is null");
    if (count == 0) return MX_OK;
    // one launch: the STFT kernel's epilogue writes the texels (and, if asked, the magnitudes as well)
    return stft_launch(ctx, a, N, kRanges, 0, 0, d_ranges, count, kmin, kmax, d_mags, nullptr, d_rgb, k);
  });
}

int mx_stft_ranges_rgb_mags(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *ranges, int64_t count, float k,
                            float *mags_out, uint8_t *rgb_out) {
  return mx_guard([&]() -> int {
    int kmin = -1, kmax = -1;
    int rc = check_common(ctx, a, N, count, kmin, kmax);
    if (rc) return rc;
    if (count == 0) return MX_OK;
    if (!ranges || !rgb_out) return fail(MX_ERR_INVALID, "ranges / rgb_out is null");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t row = (size_t)(N / 2);
    const int64_t chunk = std::min<int64_t>(count, chunk_frames(N));
    float *d_mags = nullptr;
    uint8_t *d_rgb = nullptr;
    int32_t *d_ranges = nullptr;
    std::lock_guard<std::mutex> slk(ctx->stage_mu);
    hipError_t e = stage_get(ctx, 3, (size_t)chunk * row * 3, (void **)&d_rgb);
    if (e == hipSuccess && mags_out) e = stage_get(ctx, 0, (size_t)chunk * row * sizeof(float), (void **)&d_mags);
    if (e == hipSuccess) e = stage_get(ctx, 2, (size_t)chunk * 2 * sizeof(int32_t), (void **)&d_ranges);
    if (e != hipSuccess) return fail(MX_ERR_NOMEM, "device staging buffers: %s", hipGetErrorString(e));
    for (int64_t done = 0; done < count && rc == MX_OK; done += chunk) {
      const int64_t c = std::min(chunk, count - done);
      e = hipMemcpyAsync(d_ranges, ranges + 2 * done, (size_t)c * 2 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
      if (e != hipSuccess) { rc = fail(MX_ERR_DEVICE, "ranges upload: %s", hipGetErrorString(e)); break; }
      rc = mx_stft_ranges_rgb_dev(ctx, a, N, d_ranges, c, k, d_mags, d_rgb);
      if (rc) break;
      e = hipMemcpyAsync(rgb_out + (size_t)done * row * 3, d_rgb, (size_t)c * row * 3, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess && mags_out)
        e = hipMemcpyAsync(mags_out + (size_t)done * row, d_mags, (size_t)c * row * sizeof(float), hipMemcpyDeviceToHost,
                           ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) rc = fail(MX_ERR_DEVICE, "texel download: %s", hipGetErrorString(e));
    }
    stage_trim(ctx);
    return rc;
  });
}

int mx_stft_ranges_rgb(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *ranges, int64_t count, float k,
                       uint8_t *rgb_out) {
  return mx_guard([&]() -> int {
    return mx_stft_ranges_rgb_mags(ctx, a, N, ranges, count, k, nullptr, rgb_out);
  });
}

}  // extern "C"
