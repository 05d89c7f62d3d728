// capi_pyramid.cpp — the waveform min/max pyramid (App::calcPicks, app.cpp:347-378).
// One unit of the C-ABI implementation behind include/melonix_amd.h (see capi_internal.h).  There is no CPU compute path:
// every transform entry point needs a live gfx950 device and fails with MX_ERR_DEVICE otherwise.
#include "capi_internal.h"

using namespace mx;

extern "C" {

// ---- waveform pyramid ---------------------------------------------------------------------
int mx_minmax_pyramid_dev(mx_ctx *ctx, const mx_audio *a, float *d_picks, int64_t *counts_out, int *nlevels) {
  return mx_guard([&]() -> int {
    if (!ctx || !a || !counts_out || !nlevels || (a->n > 2 && !d_picks)) return fail(MX_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_picks(a->d_padded, a->n, d_picks, counts_out, nlevels, ctx->stream));
    return MX_OK;
  });
}

int mx_minmax_pyramid(mx_ctx *ctx, const mx_audio *a, float *picks_out, int64_t *counts_out, int *nlevels) {
  return mx_guard([&]() -> int {
    if (!ctx || !a || !counts_out || !nlevels || (a->n > 2 && !picks_out)) return fail(MX_ERR_INVALID, "bad argument");
    *nlevels = 0;
    if (a->n <= 2) return MX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    float *d = nullptr;
    HIP_TRY(hipMalloc(&d, (size_t)a->n * 2 * sizeof(float)));
    int rc = mx_minmax_pyramid_dev(ctx, a, d, counts_out, nlevels);
    if (rc == MX_OK) {
      int64_t pairs = 0;
      for (int l = 0; l < *nlevels; ++l) pairs += counts_out[l];
      hipError_t e = hipMemcpyAsync(picks_out, d, (size_t)pairs * 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) rc = fail(MX_ERR_DEVICE, "pyramid download: %s", hipGetErrorString(e));
    }
    hipFree(d);
    return rc;
  });
}

void mx_minmax_range(const float *host_wav, int64_t n, const float *picks, const int64_t *counts, int nlevels, int start,
                     int end, float *mn, float *mx) {
  mx_guard_void([&] {
    float a = 0.f, b = 0.f;
    if (host_wav && picks && counts) minmax_from_range(host_wav, n, picks, counts, nlevels, start, end, a, b);
    if (mn) *mn = a;
    if (mx) *mx = b;
  });
}
}  // extern "C"
