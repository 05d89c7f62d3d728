// capi_internal.h — what the units of the C-ABI implementation (capi_*.cpp) share: the context and audio handle
// types, the error slot behind mx_last_error, the per-N table cache.  Not installed; include/melonix_amd.h is the boundary.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/melonix_amd.h"
#include "host_logic.h"
#include "kernels.h"

namespace mx {

// records the message mx_last_error returns on this thread and hands `code` back (a fixed thread-local buffer: recording an
// error allocates nothing and cannot throw — it is what the handlers below call)
int fail(int code, const char *fmt, ...) noexcept __attribute__((format(printf, 2, 3)));

// NOTHING THROWN CROSSES extern "C" (SURVEY 8b "Errors": the reference's callers get empty results, never exceptions; an
// exception that reached the C boundary would std::terminate the editor).  By construction: the body of EVERY entry point of
// capi_*.cpp is a lambda run by one of these three — tests/test_abi.py checks that of the sources, tests/cpp/fault_sweep.cpp makes
// operator new fail at the k-th call inside the library for every k and watches the statuses.
//   mx_guard       entry points that return a status (int) or a count / size that is negative on error (int64_t)
//   mx_guard_or    entry points that return a value (time maps: double, float, int): `fallback` + the error recorded
//   mx_guard_void  entry points that return nothing (destructors, pure out-parameter helpers)
template <class T, class F>
T mx_guard_or(T fallback, F &&body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    fail(MX_ERR_NOMEM, "out of host memory");
  } catch (const std::exception &e) {
    fail(MX_ERR_INVALID, "%s", e.what());
  } catch (...) {
    fail(MX_ERR_INVALID, "unknown exception");
  }
  return fallback;
}
template <class F>
auto mx_guard(F &&body) noexcept -> decltype(body()) {
  using R = decltype(body());
  try {
    return body();
  } catch (const std::bad_alloc &) {
    return (R)fail(MX_ERR_NOMEM, "out of host memory");
  } catch (const std::exception &e) {
    return (R)fail(MX_ERR_INVALID, "%s", e.what());
  } catch (...) {
    return (R)fail(MX_ERR_INVALID, "unknown exception");
  }
}
template <class F>
void mx_guard_void(F &&body) noexcept {
  try {
    body();
  } catch (const std::bad_alloc &) {
    fail(MX_ERR_NOMEM, "out of host memory");
  } catch (const std::exception &e) {
    fail(MX_ERR_INVALID, "%s", e.what());
  } catch (...) {
    fail(MX_ERR_INVALID, "unknown exception");
  }
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return ::mx::fail(MX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct NTables {
  float2 *tw2 = nullptr, *tw3 = nullptr, *ubase = nullptr;
  float *wext = nullptr;  // d-indexed window weights, pre-scaled by 1/(2N)
  std::vector<float> wext_host;
};

struct PvPipe;  // capi_pv.cpp: the phase vocoder's bounded work arena, streams and events

}  // namespace mx

struct mx_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::map<int, mx::NTables> tables;
  std::map<std::pair<int, int>, float *> wtabs;  // (N, hop) -> forward weights
  int frames_per_block = 0;  // 0 = per-N default
  std::mutex mu;
  // host landing zone of the zero-crossing bitmaps (mx_grains_dev), kept between calls: a copy into
  // pages that are already mapped runs at PCIe rate, a fresh 2 x n/8-byte buffer pays ~3 ms of faults
  std::mutex zc_mu;
  mx::ZcBitmaps zc_scratch;
  // device staging of the host-pointer entry points (mx_stft_ranges, mx_stft_hop, mx_stft_ranges_rgb*):
  // grow-only buffers kept between calls — a screen-sized batch otherwise spends more time in
  // hipMalloc/hipFree than in the kernel.  One host-staged call per context at a time.
  std::mutex stage_mu;
  struct Stage {
    void *p = nullptr;
    size_t cap = 0;
  } stage[4];
  // device work buffers of the grain chain (mx_grains_dev): the two predicate bitmaps, the rank tables, the lifting
  // tables — kept between calls like the staging buffers (guarded by zc_mu)
  Stage chain[4];
  // the phase vocoder's bounded work arena, second stream and events (capi_pv.cpp): built on first use, kept for the next
  // call, released by mx_ctx_release_scratch / mx_ctx_destroy; pv_chunk_frames = 0: the default chunk length
  std::mutex pv_mu;
  mx::PvPipe *pv = nullptr;
  int64_t pv_chunk_frames = 0;
  // the arena's budget: set (mx_pv_set_arena_budget; 0 = not set) and the automatic one (a quarter of the free device memory when
  // the context first needed an arena; forgotten by mx_ctx_release_scratch)
  int64_t pv_budget_bytes = 0, pv_budget_auto = 0;
  // a signal overflowed the compact record regions once: the context's arenas are laid out with full-size regions from then on
  // (until mx_ctx_release_scratch)
  bool pv_rec_full = false;
};

struct mx_audio {
  float *d_padded = nullptr;
  int64_t n = 0;
  bool owned = false;
};

namespace mx {

int default_frames_per_block(int N, int mode, int hop, int64_t count);
int get_tables(mx_ctx *ctx, int N, NTables &out);
int get_wtab(mx_ctx *ctx, int N, int hop, const NTables &nt, const float **out);
int check_common(mx_ctx *ctx, const mx_audio *a, int N, int64_t count, int &kmin, int &kmax);
int stft_launch(mx_ctx *ctx, const mx_audio *a, int N, int mode, int hop, int64_t first_frame, const int32_t *d_ranges,
                int64_t count, int kmin, int kmax, float *d_mags, mx_pitch *d_pitch, uint8_t *d_rgb, float cmap_k,
                int run_length = 0);
// frames per host-staging chunk: keep the device staging buffer <= ~1 GiB
int64_t chunk_frames(int N);
// Staging slot `i` with room for `bytes` (contents undefined).  Caller holds ctx->stage_mu.
hipError_t stage_get(mx_ctx *ctx, int i, size_t bytes, void **out);
// Bulk jobs stage up to 1 GiB per buffer: give those back, keep what a screen of columns needs.
void stage_trim(mx_ctx *ctx);
// gives the phase vocoder's arena, stream and events back (capi_pv.cpp); the caller holds ctx->pv_mu or owns the context
// outright (mx_ctx_destroy)
void pv_release(mx_ctx *ctx);

}  // namespace mx
