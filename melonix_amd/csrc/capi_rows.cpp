// capi_rows.cpp — device-resident magnitude rows (Spec's device-side row cache) and the colormap.
// One unit of the C-ABI implementation behind include/melonix_amd.h (see capi_internal.h).  There is no CPU compute path:
// every transform entry point needs a live gfx950 device and fails with MX_ERR_DEVICE otherwise.
#include "capi_internal.h"

using namespace mx;

extern "C" {

// ---- device-resident magnitude rows (Spec's device-side row cache) ---------------------------------
struct mx_rows {
  int N = 0;
  int device = 0;
  int64_t count = 0;
  float *d = nullptr;  // count x N/2
};

int mx_stft_ranges_keep(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *ranges, int64_t count, float k,
                        float *mags_out, uint8_t *rgb_out, mx_rows **rows_out) {
  return mx_guard([&]() -> int {
    if (rows_out) *rows_out = nullptr;
    int kmin = -1, kmax = -1;
    int rc = check_common(ctx, a, N, count, kmin, kmax);
    if (rc) return rc;
    if (!rows_out) return fail(MX_ERR_INVALID, "rows_out is null");
    if (count == 0) return MX_OK;
    if (!ranges) return fail(MX_ERR_INVALID, "ranges is null");
    const bool want_rgb = rgb_out != nullptr && k != 0.f;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t row = (size_t)(N / 2);
    std::unique_ptr<mx_rows> keep(new (std::nothrow) mx_rows);
    if (!keep) return fail(MX_ERR_NOMEM, "host memory");
    keep->N = N;
    keep->device = ctx->device;
    keep->count = count;
    hipError_t e = hipMalloc((void **)&keep->d, (size_t)count * row * sizeof(float));
    if (e != hipSuccess) return fail(MX_ERR_NOMEM, "device rows (%lld x %zu floats): %s", (long long)count, row, hipGetErrorString(e));
    const int64_t chunk = std::min<int64_t>(count, chunk_frames(N));
    uint8_t *d_rgb = nullptr;
    int32_t *d_ranges = nullptr;
    {
      std::lock_guard<std::mutex> slk(ctx->stage_mu);
      e = hipSuccess;
      if (want_rgb) e = stage_get(ctx, 3, (size_t)chunk * row * 3, (void **)&d_rgb);
      if (e == hipSuccess) e = stage_get(ctx, 2, (size_t)chunk * 2 * sizeof(int32_t), (void **)&d_ranges);
      if (e != hipSuccess) rc = fail(MX_ERR_NOMEM, "device staging buffers: %s", hipGetErrorString(e));
      for (int64_t done = 0; done < count && rc == MX_OK; done += chunk) {
        const int64_t c = std::min(chunk, count - done);
        float *dm = keep->d + (size_t)done * row;
        e = hipMemcpyAsync(d_ranges, ranges + 2 * done, (size_t)c * 2 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { rc = fail(MX_ERR_DEVICE, "ranges upload: %s", hipGetErrorString(e)); break; }
        rc = want_rgb ? mx_stft_ranges_rgb_dev(ctx, a, N, d_ranges, c, k, dm, d_rgb)
                      : mx_stft_ranges_dev(ctx, a, N, d_ranges, c, -1, -1, dm, nullptr);
        if (rc) break;
        if (want_rgb)
          e = hipMemcpyAsync(rgb_out + (size_t)done * row * 3, d_rgb, (size_t)c * row * 3, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && mags_out)
          e = hipMemcpyAsync(mags_out + (size_t)done * row, dm, (size_t)c * row * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(MX_ERR_DEVICE, "row download: %s", hipGetErrorString(e));
      }
      stage_trim(ctx);
    }
    if (rc) {
      hipFree(keep->d);
      return rc;
    }
    *rows_out = keep.release();
    return MX_OK;
  });
}

int64_t mx_rows_count(const mx_rows *rows) {
  return mx_guard([&]() -> int64_t {
    return rows ? rows->count : 0;
  });
}

void mx_rows_free(mx_ctx *ctx, mx_rows *rows) {
  mx_guard_void([&] {
    if (!rows) return;
    if (ctx) hipSetDevice(ctx->device);
    if (rows->d) hipFree(rows->d);
    delete rows;
  });
}

namespace {
int rows_span_ok(mx_ctx *ctx, const mx_rows *rows, int64_t first, int64_t count, const void *out) {
  if (!ctx || !rows) return fail(MX_ERR_INVALID, "null context / rows");
  if (rows->device != ctx->device) return fail(MX_ERR_INVALID, "rows belong to another device");
  if (first < 0 || count < 0 || first + count > rows->count) return fail(MX_ERR_INVALID, "row span outside the batch");
  if (count > 0 && !out) return fail(MX_ERR_INVALID, "output is null");
  return MX_OK;
}
}  // namespace

int mx_rows_fetch(mx_ctx *ctx, const mx_rows *rows, int64_t first, int64_t count, float *mags_out) {
  return mx_guard([&]() -> int {
    int rc = rows_span_ok(ctx, rows, first, count, mags_out);
    if (rc || count == 0) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t row = (size_t)(rows->N / 2);
    HIP_TRY(hipMemcpyAsync(mags_out, rows->d + (size_t)first * row, (size_t)count * row * sizeof(float), hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MX_OK;
  });
}

int mx_rows_colormap(mx_ctx *ctx, const mx_rows *rows, int64_t first, int64_t count, float k, uint8_t *rgb_out) {
  return mx_guard([&]() -> int {
    int rc = rows_span_ok(ctx, rows, first, count, rgb_out);
    if (rc || count == 0) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t row = (size_t)(rows->N / 2);
    const int64_t chunk = std::min<int64_t>(count, chunk_frames(rows->N));
    std::lock_guard<std::mutex> slk(ctx->stage_mu);
    uint8_t *d_rgb = nullptr;
    hipError_t e = stage_get(ctx, 3, (size_t)chunk * row * 3, (void **)&d_rgb);
    if (e != hipSuccess) return fail(MX_ERR_NOMEM, "device staging buffer: %s", hipGetErrorString(e));
    for (int64_t done = 0; done < count && rc == MX_OK; done += chunk) {
      const int64_t c = std::min(chunk, count - done);
      e = launch_colormap(rows->d + (size_t)(first + done) * row, d_rgb, (int64_t)c * (int64_t)row, k, ctx->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(rgb_out + (size_t)done * row * 3, d_rgb, (size_t)c * row * 3, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) rc = fail(MX_ERR_DEVICE, "re-colouring rows: %s", hipGetErrorString(e));
    }
    stage_trim(ctx);
    return rc;
  });
}

int mx_colormap_dev(mx_ctx *ctx, const float *d_mags, int64_t nbins_total, float k, uint8_t *d_rgb) {
  return mx_guard([&]() -> int {
    if (!ctx || nbins_total < 0 || (nbins_total > 0 && (!d_mags || !d_rgb))) return fail(MX_ERR_INVALID, "bad argument");
    if (nbins_total % 4) return fail(MX_ERR_INVALID, "bin count must be a multiple of 4 (whole rows)");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_colormap(d_mags, d_rgb, nbins_total, k, ctx->stream));
    return MX_OK;
  });
}

}  // extern "C"
