// capi.cpp — implementation of the C-ABI boundary declared in
// include/melonix_amd.h.  Built by hipcc into libmelonix_amd.so together with
// the gfx950 kernels.  There is no CPU compute path here: every transform
// entry point needs a live gfx950 device and fails with MX_ERR_DEVICE otherwise.
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
#include "stft_tables.h"

using namespace mx;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(MX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct NTables {
  float2 *tw2 = nullptr, *tw3 = nullptr, *ubase = nullptr;
  float *wext = nullptr;  // d-indexed window weights, pre-scaled by 1/(2N)
  std::vector<float> wext_host;
};

}  // namespace

struct mx_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::map<int, NTables> tables;
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
  // work buffers of the phase vocoder (tens of GB for an hour of audio): hipMalloc of that size takes of the
  // order of a second, so the arena is kept for the next call; mx_ctx_destroy releases it
  std::mutex pv_mu;
  Stage pv_arena;
  // the staged (one rank of a multi-GPU run) phase-vocoder job between mx_pv_shard_analyze and _finish
  mx::PvArgs pv_job{};
  bool pv_job_active = false, pv_job_last = false;
  char *pv_slot_carry = nullptr, *pv_slot_prev_tail = nullptr, *pv_slot_next_head = nullptr;
};

struct mx_audio {
  float *d_padded = nullptr;
  int64_t n = 0;
  bool owned = false;
};

namespace {

// Consecutive frames one workgroup walks.  Long runs amortise the per-workgroup setup (full window
// load, twiddle fetch) over 16-32 frames; short batches (a screen of 1280 columns) use fewer frames per
// workgroup so that every CU still gets work (>= ~8 workgroups per CU when there are enough frames).
int default_frames_per_block(int N, int mode, int hop, int64_t count) {
  if (const char *e = getenv("MELONIX_FRAMES_PER_BLOCK")) {
    const int v = atoi(e);
    if (v > 0) return v;
  }
  const int cap = stft_frames_per_block_cap(N, mode, hop);
  const int64_t want_blocks = 2048;
  const int64_t want = std::max<int64_t>(1, std::min<int64_t>(cap, count / want_blocks));
  // a power of two: the sliding / circular-window kernels restart their decay chains at the head of every run, so
  // rows are a function of where the runs start; with run lengths 1, 2, 4 .. cap (a power of two itself) a launch
  // that starts on a multiple of `cap` frames is cut on the same run heads as any longer launch with the same run
  // length (mx_stft_run_length + mx_ctx_set_frames_per_block pin that length for the shards of a multi-GPU job)
  int g = 1;
  while (2 * g <= want) g *= 2;
  return g;
}

template <class P>
int build_tables(NTables &t) {
  constexpr int N = P::N;
  const auto tw2 = make_tw2<P>();
  const auto tw3 = make_tw3<P>();
  const auto ub = make_ubase<P>();
  HIP_TRY(hipMalloc(&t.tw2, tw2.size() * sizeof(float2)));
  HIP_TRY(hipMalloc(&t.tw3, tw3.size() * sizeof(float2)));
  HIP_TRY(hipMalloc(&t.ubase, ub.size() * sizeof(float2)));
  HIP_TRY(hipMemcpy(t.tw2, tw2.data(), tw2.size() * sizeof(float2), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(t.tw3, tw3.data(), tw3.size() * sizeof(float2), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(t.ubase, ub.data(), ub.size() * sizeof(float2), hipMemcpyHostToDevice));
  t.wext_host = make_wext(fold_scale(N));
  HIP_TRY(hipMalloc(&t.wext, t.wext_host.size() * sizeof(float)));
  HIP_TRY(hipMemcpy(t.wext, t.wext_host.data(), t.wext_host.size() * sizeof(float), hipMemcpyHostToDevice));
  return MX_OK;
}

int get_tables(mx_ctx *ctx, int N, NTables &out) {
  std::lock_guard<std::mutex> lk(ctx->mu);
  auto it = ctx->tables.find(N);
  if (it != ctx->tables.end()) {
    out = it->second;
    return MX_OK;
  }
  NTables t;
  int rc;
  switch (N) {
    case 4096: rc = build_tables<Plan<4096, kPlan4096E>>(t); break;
    case 16384: rc = build_tables<Plan<16384, 32>>(t); break;
    case 32768: rc = build_tables<Plan<32768, 32>>(t); break;
    default: return fail(MX_ERR_INVALID, "unsupported FFT size %d (supported: 4096, 16384, 32768)", N);
  }
  if (rc) return rc;
  ctx->tables[N] = t;
  out = t;
  return MX_OK;
}

int get_wtab(mx_ctx *ctx, int N, int hop, const NTables &nt, const float **out) {
  std::lock_guard<std::mutex> lk(ctx->mu);
  const auto key = std::make_pair(N, hop);
  auto it = ctx->wtabs.find(key);
  if (it != ctx->wtabs.end()) {
    *out = it->second;
    return MX_OK;
  }
  const std::vector<float> w = make_wtab(N, hop, nt.wext_host);
  float *d = nullptr;
  HIP_TRY(hipMalloc(&d, w.size() * sizeof(float)));
  HIP_TRY(hipMemcpy(d, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
  ctx->wtabs[key] = d;
  *out = d;
  return MX_OK;
}

int check_common(mx_ctx *ctx, const mx_audio *a, int N, int64_t count, int &kmin, int &kmax) {
  if (!ctx || !a) return fail(MX_ERR_INVALID, "null context or audio handle");
  if (N != 4096 && N != 16384 && N != 32768)
    return fail(MX_ERR_INVALID, "unsupported FFT size %d (supported: 4096, 16384, 32768)", N);
  if (count < 0) return fail(MX_ERR_INVALID, "negative frame count");
  if (kmin < 0 && kmax < 0) mx_pitch_band(N, 48000, &kmin, &kmax);
  kmin = std::max(kmin, 0);
  kmax = std::min(kmax, N / 2 - 1);
  if (kmin > kmax) return fail(MX_ERR_INVALID, "empty pitch band [%d,%d]", kmin, kmax);
  return MX_OK;
}

int stft_launch(mx_ctx *ctx, const mx_audio *a, int N, int mode, int hop, int64_t first_frame,
                const int32_t *d_ranges, int64_t count, int kmin, int kmax, float *d_mags,
                mx_pitch *d_pitch, uint8_t *d_rgb, float cmap_k, int run_length = 0) {
  // HIP's current device is per thread: the tables below must land on the context's GPU whichever
  // thread makes the first call
  HIP_TRY(hipSetDevice(ctx->device));
  NTables t;
  int rc = get_tables(ctx, N, t);
  if (rc) return rc;
  StftArgs s{};
  s.audio = a->d_padded;
  s.n = a->n;
  s.wext = t.wext;
  s.decay = hop_decay(hop);
  s.tw2 = t.tw2;
  s.tw3 = t.tw3;
  s.ubase = t.ubase;
  s.ranges = d_ranges;
  s.hop = hop;
  s.first_frame = first_frame;
  s.count = count;
  s.kmin = kmin;
  s.kmax = kmax;
  s.mags = d_mags;
  s.pitch = d_pitch;
  s.rgb = d_rgb;
  s.cmap_k = cmap_k;
  // the context's pinned run length (mx_ctx_set_frames_per_block: the shards of a multi-GPU job, bench sweeps) is a
  // property of BULK launches — ranges mode has no run heads (every column loads the exact table) and a screen-sized
  // batch must keep its short runs so that every CU gets work
  s.frames_per_block = (mode != kRanges && ctx->frames_per_block > 0) ? ctx->frames_per_block
                       : run_length > 0                              ? run_length
                                                                     : default_frames_per_block(N, mode, hop, count);
  if (mode != kRanges) {
    rc = get_wtab(ctx, N, hop, t, &s.wtab);
    if (rc) return rc;
  }
  HIP_TRY(launch_stft(N, mode, s, ctx->stream));
  return MX_OK;
}

// frames per host-staging chunk: keep the device staging buffer <= ~1 GiB
int64_t chunk_frames(int N) { return std::max<int64_t>(1, (int64_t)(1ull << 30) / ((int64_t)(N / 2) * 4)); }

// Staging slot `i` with room for `bytes` (contents undefined).  Caller holds ctx->stage_mu.
hipError_t stage_get(mx_ctx *ctx, int i, size_t bytes, void **out) {
  mx_ctx::Stage &st = ctx->stage[i];
  if (st.cap < bytes) {
    if (st.p) hipFree(st.p);
    st.p = nullptr;
    st.cap = 0;
    const hipError_t e = hipMalloc(&st.p, bytes);
    if (e != hipSuccess) return e;
    st.cap = bytes;
  }
  *out = st.p;
  return hipSuccess;
}
// Bulk jobs stage up to 1 GiB per buffer: give those back, keep what a screen of columns needs.
void stage_trim(mx_ctx *ctx) {
  for (auto &st : ctx->stage)
    if (st.cap > ((size_t)256 << 20)) {
      hipFree(st.p);
      st.p = nullptr;
      st.cap = 0;
    }
}

}  // namespace

// ===========================================================================
extern "C" {

const char *mx_last_error(void) { return g_err.c_str(); }
const char *mx_version(void) { return "melonix_amd 0.1.0 gfx950"; }

int mx_ctx_create(int device, mx_ctx **out) {
  if (!out) return fail(MX_ERR_INVALID, "out is null");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(MX_ERR_DEVICE, "no HIP device visible (%s); melonix_amd has no CPU path",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail(MX_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(MX_ERR_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code objects only", device,
                prop.gcnArchName);
  HIP_TRY(hipSetDevice(device));
  mx_ctx *c = new (std::nothrow) mx_ctx();
  if (!c) return fail(MX_ERR_NOMEM, "out of host memory");
  c->device = device;
  e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return fail(MX_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  *out = c;
  return MX_OK;
}

void mx_ctx_destroy(mx_ctx *ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (auto &kv : ctx->tables) {
    hipFree(kv.second.tw2);
    hipFree(kv.second.tw3);
    hipFree(kv.second.ubase);
    hipFree(kv.second.wext);
  }
  for (auto &kv : ctx->wtabs) hipFree(kv.second);
  for (auto &st : ctx->stage) hipFree(st.p);
  for (auto &st : ctx->chain) hipFree(st.p);
  hipFree(ctx->pv_arena.p);
  hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int mx_ctx_set_stream(mx_ctx *ctx, void *hip_stream) {
  if (!ctx) return fail(MX_ERR_INVALID, "null context");
  ctx->stream = (hipStream_t)hip_stream;  // NULL is the HIP null stream (torch's default stream)
  return MX_OK;
}

int mx_ctx_use_own_stream(mx_ctx *ctx) {
  if (!ctx) return fail(MX_ERR_INVALID, "null context");
  ctx->stream = ctx->own_stream;
  return MX_OK;
}

int mx_ctx_synchronize(mx_ctx *ctx) {
  if (!ctx) return fail(MX_ERR_INVALID, "null context");
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MX_OK;
}

int mx_ctx_release_scratch(mx_ctx *ctx) {
  if (!ctx) return fail(MX_ERR_INVALID, "null context");
  HIP_TRY(hipSetDevice(ctx->device));
  {
    std::lock_guard<std::mutex> lk(ctx->stage_mu);
    for (auto &st : ctx->stage) {
      hipFree(st.p);
      st = {};
    }
  }
  {
    std::lock_guard<std::mutex> lk(ctx->pv_mu);
    hipFree(ctx->pv_arena.p);
    ctx->pv_arena = {};
    ctx->pv_job_active = false;  // a staged phase-vocoder job lives in that arena
  }
  {
    std::lock_guard<std::mutex> lk(ctx->zc_mu);
    ctx->zc_scratch = ZcBitmaps{};
    for (auto &st : ctx->chain) {
      hipFree(st.p);
      st = {};
    }
  }
  return MX_OK;
}

int mx_pinned_alloc(mx_ctx *ctx, size_t bytes, void **out) {
  if (!ctx || !out) return fail(MX_ERR_INVALID, "null context / out");
  *out = nullptr;
  if (bytes == 0) return MX_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    *out = nullptr;
    return fail(MX_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
  }
  return MX_OK;
}

void mx_pinned_free(mx_ctx *ctx, void *p) {
  if (!p) return;
  if (ctx) hipSetDevice(ctx->device);
  hipHostFree(p);
}

int mx_stft_run_length(int N, int hop, int64_t count) {
  if ((N != 4096 && N != 16384 && N != 32768) || hop <= 0 || count < 0) return fail(MX_ERR_INVALID, "bad argument");
  return default_frames_per_block(N, (hop & 1) ? kBulkAny : kBulkAligned, hop, count);
}

int mx_ctx_set_frames_per_block(mx_ctx *ctx, int g) {  // tuning knob (bench sweeps)
  if (!ctx || g < 0) return fail(MX_ERR_INVALID, "bad argument");
  ctx->frames_per_block = g;
  return MX_OK;
}

// ---- audio ------------------------------------------------------------------
int mx_audio_upload(mx_ctx *ctx, const float *host_wav, int64_t n, mx_audio **out) {
  if (!ctx || !out || n < 0 || (n > 0 && !host_wav)) return fail(MX_ERR_INVALID, "bad argument");
  if (n > 0x7fffffffLL - 2 * MX_AUDIO_PAD)
    return fail(MX_ERR_INVALID, "audio longer than the reference's int sample indices allow");
  HIP_TRY(hipSetDevice(ctx->device));
  mx_audio *a = new (std::nothrow) mx_audio();
  if (!a) return fail(MX_ERR_NOMEM, "out of host memory");
  const size_t total = (size_t)n + 2 * (size_t)MX_AUDIO_PAD;
  hipError_t e = hipMalloc(&a->d_padded, total * sizeof(float));
  if (e == hipSuccess) e = hipMemsetAsync(a->d_padded, 0, total * sizeof(float), ctx->stream);
  if (e == hipSuccess && n > 0)
    e = hipMemcpyAsync(a->d_padded + MX_AUDIO_PAD, host_wav, (size_t)n * sizeof(float), hipMemcpyHostToDevice,
                       ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    if (a->d_padded) hipFree(a->d_padded);
    delete a;
    return fail(MX_ERR_DEVICE, "audio upload: %s", hipGetErrorString(e));
  }
  a->n = n;
  a->owned = true;
  *out = a;
  return MX_OK;
}

int mx_audio_wrap_device(mx_ctx *ctx, const float *d_padded, int64_t n, mx_audio **out) {
  if (!ctx || !out || !d_padded || n < 0) return fail(MX_ERR_INVALID, "bad argument");
  if (reinterpret_cast<uintptr_t>(d_padded) & 15)  // the kernels use 8- and 16-byte loads of the samples
    return fail(MX_ERR_INVALID, "device audio buffer must be 16-byte aligned");
  mx_audio *a = new (std::nothrow) mx_audio();
  if (!a) return fail(MX_ERR_NOMEM, "out of host memory");
  a->d_padded = const_cast<float *>(d_padded);
  a->n = n;
  a->owned = false;
  *out = a;
  return MX_OK;
}

int64_t mx_audio_length(const mx_audio *a) { return a ? a->n : -1; }

int mx_audio_free(mx_ctx *ctx, mx_audio *a) {
  if (!a) return MX_OK;
  if (a->owned) {
    if (ctx) {
      hipSetDevice(ctx->device);
      hipStreamSynchronize(ctx->stream);
    }
    hipFree(a->d_padded);
  }
  delete a;
  return MX_OK;
}

// ---- STFT ---------------------------------------------------------------------
void mx_pitch_band(int N, int sampleRate, int *kmin, int *kmax) {
  // notes 24..84 of the default view (app.hpp:45-46): f = 55*2^((note-24)/12), bin = f*N/sr (app.cpp:499-516)
  const double lo = 55.0 * N / sampleRate, hi = 1760.0 * N / sampleRate;
  int a = (int)lo;
  if ((double)a < lo) ++a;
  if (kmin) *kmin = a;
  if (kmax) *kmax = (int)hi;
}

double mx_bin_note(int bin, int N, int sampleRate) {
  if (bin <= 0 || N <= 0 || sampleRate <= 0) return -HUGE_VAL;
  return 24. + 12. * std::log2((double)bin * sampleRate / N / 55.);
}
double mx_note_bin(double note, int N, int sampleRate) {
  if (N <= 0 || sampleRate <= 0) return 0.;
  return 55. * std::pow(2., (note - 24.) / 12.) * N / sampleRate;  // app.cpp:498
}

int64_t mx_frame_count(int64_t n, int hop) { return hop > 0 && n >= 0 ? (n + hop - 1) / hop : -1; }

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
  return stft_hop_dev_run(ctx, a, N, hop, first_frame, count, kmin, kmax, d_mags, d_pitch, 0);
}

int mx_stft_ranges_dev(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *d_ranges, int64_t count, int kmin,
                       int kmax, float *d_mags, mx_pitch *d_pitch) {
  int rc = check_common(ctx, a, N, count, kmin, kmax);
  if (rc) return rc;
  if (count > 0 && !d_ranges) return fail(MX_ERR_INVALID, "ranges is null");
  return stft_launch(ctx, a, N, kRanges, 0, 0, d_ranges, count, kmin, kmax, d_mags, d_pitch, nullptr, 0.f);
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
  if (hop <= 0) return fail(MX_ERR_INVALID, "hop must be positive");
  return stft_host_common(ctx, a, N, false, hop, first_frame, nullptr, count, kmin, kmax, mags_out, pitch_out);
}

int mx_stft_ranges(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *ranges, int64_t count, int kmin, int kmax,
                   float *mags_out, mx_pitch *pitch_out) {
  if (count > 0 && !ranges) return fail(MX_ERR_INVALID, "ranges is null");
  return stft_host_common(ctx, a, N, true, 0, 0, ranges, count, kmin, kmax, mags_out, pitch_out);
}

int mx_stft_ranges_rgb_dev(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *d_ranges, int64_t count, float k,
                           float *d_mags, uint8_t *d_rgb) {
  int kmin = -1, kmax = -1;
  int rc = check_common(ctx, a, N, count, kmin, kmax);
  if (rc) return rc;
  if (count > 0 && (!d_ranges || !d_rgb)) return fail(MX_ERR_INVALID, "ranges / rgb is null");
  if (count == 0) return MX_OK;
  // one launch: the STFT kernel's epilogue writes the texels (and, if asked, the magnitudes as well)
  return stft_launch(ctx, a, N, kRanges, 0, 0, d_ranges, count, kmin, kmax, d_mags, nullptr, d_rgb, k);
}

int mx_stft_ranges_rgb_mags(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *ranges, int64_t count, float k,
                            float *mags_out, uint8_t *rgb_out) {
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
}

int mx_stft_ranges_rgb(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *ranges, int64_t count, float k,
                       uint8_t *rgb_out) {
  return mx_stft_ranges_rgb_mags(ctx, a, N, ranges, count, k, nullptr, rgb_out);
}

// ---- device-resident magnitude rows (Spec's device-side row cache) ---------------------------------
struct mx_rows {
  int N = 0;
  int device = 0;
  int64_t count = 0;
  float *d = nullptr;  // count x N/2
};

int mx_stft_ranges_keep(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *ranges, int64_t count, float k,
                        float *mags_out, uint8_t *rgb_out, mx_rows **rows_out) {
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
}

int64_t mx_rows_count(const mx_rows *rows) { return rows ? rows->count : 0; }

void mx_rows_free(mx_ctx *ctx, mx_rows *rows) {
  if (!rows) return;
  if (ctx) hipSetDevice(ctx->device);
  if (rows->d) hipFree(rows->d);
  delete rows;
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
  int rc = rows_span_ok(ctx, rows, first, count, mags_out);
  if (rc || count == 0) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t row = (size_t)(rows->N / 2);
  HIP_TRY(hipMemcpyAsync(mags_out, rows->d + (size_t)first * row, (size_t)count * row * sizeof(float), hipMemcpyDeviceToHost,
                         ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MX_OK;
}

int mx_rows_colormap(mx_ctx *ctx, const mx_rows *rows, int64_t first, int64_t count, float k, uint8_t *rgb_out) {
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
}

int mx_colormap_dev(mx_ctx *ctx, const float *d_mags, int64_t nbins_total, float k, uint8_t *d_rgb) {
  if (!ctx || nbins_total < 0 || (nbins_total > 0 && (!d_mags || !d_rgb))) return fail(MX_ERR_INVALID, "bad argument");
  if (nbins_total % 4) return fail(MX_ERR_INVALID, "bin count must be a multiple of 4 (whole rows)");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(launch_colormap(d_mags, d_rgb, nbins_total, k, ctx->stream));
  return MX_OK;
}

// ---- build-defined phase-vocoder pitch shift (no reference counterpart) ---------------------------
namespace {
constexpr int kPvN = 4096, kPvM = kPvN / 2, kPvHs = 256, kPvSeam = kPvN - kPvHs;

int64_t pv_frame_count(int64_t n, double r) { return (int64_t)std::ceil((double)n * r / kPvHs) + 1; }
// smallest output sample whose interpolation base floor(i*r + N/2) reaches stretched sample q (same binary64
// expression as pv_resample evaluates)
int64_t pv_first_output_at(int64_t q, double r, int64_t n) {
  int64_t i = (int64_t)std::ceil(((double)q - kPvN / 2) / r);
  if (i < 0) i = 0;
  while (i > 0 && (int64_t)std::floor((double)(i - 1) * r + (double)(kPvN / 2)) >= q) --i;
  while (i < n && (int64_t)std::floor((double)i * r + (double)(kPvN / 2)) < q) ++i;
  return i < n ? i : n;
}

// Lays out the work arena for frames [F_lo, F_hi) of the signal's F frames (plus, when F_lo > 0, the frame before
// them as local row 0) and fills every PvArgs field but the output pointers.  Caller holds ctx->pv_mu.
// `plan` (marker-driven variant, whole signal only): the analysis positions come from it instead of floor(f*Hs/r).
int pv_prepare(mx_ctx *ctx, const mx_audio *a, double semitones, int64_t F_lo, int64_t F_hi, bool want_totals,
               PvArgs &p, const PvPlan *plan = nullptr) {
  constexpr int N = kPvN, M = kPvM, Hs = kPvHs;
  HIP_TRY(hipSetDevice(ctx->device));  // before any table allocation: HIP's current device is per thread
  NTables t;
  int rc = get_tables(ctx, N, t);
  if (rc) return rc;
  const double r = std::pow(2.0, semitones / 12.0);
  const int64_t first = F_lo > 0 ? 1 : 0;
  const int64_t Fl = F_hi - F_lo + first;  // local rows
  // the constant-ratio plan (analysis positions, hops, Hs/hop) is written on the device; a marker plan comes from the host
  std::vector<uint32_t> hop;
  std::vector<double> hratio;
  if (plan) {
    hop.assign((size_t)Fl, 0u);
    hratio.assign((size_t)Fl, 0.0);
    for (int64_t j = 1; j < Fl; ++j) {
      const int64_t h = plan->apos[(size_t)j] - plan->apos[(size_t)j - 1];
      if (h >= 1 && h <= 0x7fffffffLL) {
        hop[(size_t)j] = (uint32_t)h;
        hratio[(size_t)j] = (double)Hs / (double)h;
      }
    }
  }
  std::vector<float> hann((size_t)N), hann_sc((size_t)N);
  for (int j = 0; j < N; ++j) {
    hann[(size_t)j] = (float)(0.5 - 0.5 * std::cos(2.0 * 3.14159265358979323846 * j / N));
    hann_sc[(size_t)j] = hann[(size_t)j] * fold_scale(N);  // exact: a power of two
  }
  std::vector<float2> wsplit((size_t)M);
  for (int c = 0; c < M; ++c) {
    const double ang = 2.0 * 3.14159265358979323846 * c / N;
    wsplit[(size_t)c] = make_float2((float)std::cos(ang), (float)std::sin(ang));
  }
  p = PvArgs{};
  p.audio = a->d_padded;
  p.n = a->n;
  p.ratio = r;
  p.frames = Fl;
  p.first = first;
  p.global_first = F_lo == 0;
  p.tw2 = t.tw2;
  p.tw3 = t.tw3;
  p.ubase = t.ubase;
  // chunks of the frame axis for the scan: about 1536 of them (their maps are composed in groups of 32), at least 64 frames each
  p.scan_chunk = (int)std::max<int64_t>(64, (Fl - first + 1535) / 1536);  // (one round of row-walking workgroups, six per CU)
  p.s_len = (Fl - first) * Hs + N;
  p.s_origin = F_lo * Hs;
  const int64_t nchunks = (Fl - first + p.scan_chunk - 1) / p.scan_chunk;
  // one arena: apos, the two windows, the complex spectra, the peak records, the peaks' synthesis offsets, chunk sums,
  // boundary halos, s (+1 for s[m+1]), split twiddles, source bins of the chunk maps, peak maps and counts, this rank's
  // total map, carry-in, the neighbours' seams.  (Records and offsets have room for a peak in every bin — silence, an
  // impulse — but only a frame's first pkcount entries are ever touched.)
  const size_t rowsz = (size_t)Fl * M;
  size_t off = 0;
  auto take = [&off](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_apos = take((size_t)Fl * 8), o_h = take(N * 4), o_hs = take(N * 4), o_m = take(rowsz * 8),
               o_p = take(rowsz * 8), o_i = take(rowsz * 4), o_pc = take((size_t)Fl * 4), o_ft = take((size_t)Fl * 4), o_c = take((size_t)nchunks * M * 4),
               o_gs = take((size_t)((nchunks + 31) / 32) * M * 4), o_go = take((size_t)((nchunks + 31) / 32) * M * 2),
               o_f = take((size_t)pv_halo_floats(Fl - first) * 4), o_s = take(((size_t)p.s_len + 1) * 4),
               o_w = take((size_t)M * 8), o_a = take((size_t)nchunks * M * 2), o_ow = take((size_t)Fl * (M / 32) * 4),
               o_ts = take((size_t)M * 4), o_ta = take((size_t)M * 2), o_ci = take((size_t)M * 4),
               o_pt = take((size_t)kPvSeam * 4), o_nh = take((size_t)kPvSeam * 4),
               o_tf = take(plan ? (size_t)Fl * 8 : 0), o_rf = take(plan ? (size_t)Fl * 8 : 0),
               o_i0 = take(plan ? ((size_t)Fl + 1) * 8 : 0), o_hp = take((size_t)Fl * 4), o_hr = take((size_t)Fl * 8);
  if (ctx->pv_arena.cap < off) {
    if (ctx->pv_arena.p) hipFree(ctx->pv_arena.p);
    ctx->pv_arena = {};
    const hipError_t em = hipMalloc(&ctx->pv_arena.p, off);
    if (em != hipSuccess) {
      ctx->pv_arena = {};
      return fail(MX_ERR_NOMEM, "phase-vocoder work buffers (%zu MiB): %s", off >> 20, hipGetErrorString(em));
    }
    ctx->pv_arena.cap = off;
  }
  char *arena = static_cast<char *>(ctx->pv_arena.p);
  hipError_t e = plan ? hipMemcpyAsync(arena + o_apos, plan->apos.data(), (size_t)Fl * 8, hipMemcpyHostToDevice, ctx->stream)
                      : launch_pv_plan_const(reinterpret_cast<int64_t *>(arena + o_apos), reinterpret_cast<uint32_t *>(arena + o_hp),
                                             reinterpret_cast<double *>(arena + o_hr), Fl, F_lo - first, r, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(arena + o_h, hann.data(), N * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(arena + o_hs, hann_sc.data(), N * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(arena + o_w, wsplit.data(), (size_t)M * 8, hipMemcpyHostToDevice, ctx->stream);
  // the last hop of s is beyond every frame, and s[s_len] backs the interpolation's m+1
  if (e == hipSuccess) e = hipMemsetAsync(arena + o_s + (size_t)(p.s_len - Hs) * 4, 0, (size_t)(Hs + 1) * 4, ctx->stream);
  if (plan) {
    if (e == hipSuccess) e = hipMemcpyAsync(arena + o_hp, hop.data(), (size_t)Fl * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(arena + o_hr, hratio.data(), (size_t)Fl * 8, hipMemcpyHostToDevice, ctx->stream);
  }
  if (plan) {
    if (e == hipSuccess) e = hipMemcpyAsync(arena + o_tf, plan->tf.data(), (size_t)Fl * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(arena + o_rf, plan->rf.data(), (size_t)Fl * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(arena + o_i0, plan->i0.data(), ((size_t)Fl + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the host tables above die with this frame
  if (e != hipSuccess) return fail(MX_ERR_DEVICE, "phase vocoder setup: %s", hipGetErrorString(e));
  p.chunk_org = reinterpret_cast<uint16_t *>(arena + o_a);
  p.pkmap = reinterpret_cast<uint32_t *>(arena + o_ow);
  p.apos = reinterpret_cast<const int64_t *>(arena + o_apos);
  p.hop = reinterpret_cast<const uint32_t *>(arena + o_hp);
  p.hratio = reinterpret_cast<const double *>(arena + o_hr);
  p.hann = reinterpret_cast<const float *>(arena + o_h);
  p.hann_scaled = reinterpret_cast<const float *>(arena + o_hs);
  p.xrows = reinterpret_cast<float2 *>(arena + o_m);
  p.recs = reinterpret_cast<uint2 *>(arena + o_p);
  p.cvals = reinterpret_cast<uint32_t *>(arena + o_i);
  p.pkcount = reinterpret_cast<uint32_t *>(arena + o_pc);
  p.fthr = reinterpret_cast<float *>(arena + o_ft);
  p.chunk_sums = reinterpret_cast<uint32_t *>(arena + o_c);
  p.group_sums = reinterpret_cast<uint32_t *>(arena + o_gs);
  p.group_org = reinterpret_cast<uint16_t *>(arena + o_go);
  p.halo = reinterpret_cast<float *>(arena + o_f);
  p.wsplit = reinterpret_cast<const float2 *>(arena + o_w);
  p.s = reinterpret_cast<float *>(arena + o_s);
  if (plan) {
    p.tf = reinterpret_cast<const double *>(arena + o_tf);
    p.rf = reinterpret_cast<const double *>(arena + o_rf);
    p.i0 = reinterpret_cast<const int64_t *>(arena + o_i0);
  }
  if (want_totals) {
    p.tot_sums = reinterpret_cast<uint32_t *>(arena + o_ts);
    p.tot_org = reinterpret_cast<uint16_t *>(arena + o_ta);
  }
  // slots the staged (multi-GPU) entry points fill from host data
  ctx->pv_slot_carry = arena + o_ci;
  ctx->pv_slot_prev_tail = arena + o_pt;
  ctx->pv_slot_next_head = arena + o_nh;
  return MX_OK;
}
}  // namespace

int mx_pv_pitch_shift_dev(mx_ctx *ctx, const mx_audio *a, double semitones, float *d_pcm_f32, int16_t *d_pcm_i16) {
  if (!ctx || !a) return fail(MX_ERR_INVALID, "null context or audio handle");
  if (!(semitones >= -48.0 && semitones <= 48.0)) return fail(MX_ERR_INVALID, "semitones out of range [-48, 48]");
  if (a->n == 0 || (!d_pcm_f32 && !d_pcm_i16)) return MX_OK;
  std::lock_guard<std::mutex> plk(ctx->pv_mu);
  ctx->pv_job_active = false;
  PvArgs p;
  const int rc = pv_prepare(ctx, a, semitones, 0, pv_frame_count(a->n, std::pow(2.0, semitones / 12.0)), false, p);
  if (rc) return rc;
  p.out_lo = 0;
  p.out_hi = a->n;
  p.pcm_f32 = d_pcm_f32;
  p.pcm_i16 = d_pcm_i16;
  hipError_t e = launch_pv(p, ctx->stream);
  const hipError_t es = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = es;
  if (e != hipSuccess) return fail(MX_ERR_DEVICE, "phase vocoder: %s", hipGetErrorString(e));
  return MX_OK;
}

// Marker-driven variant: the vocoder steered by the editor's markers as App::exportWav is (warped time, pitch bend).
int64_t mx_pv_render_length(int64_t n, int sampleRate, const mx_marker *markers, int nmarkers) {
  if (n < 0 || nmarkers < 0 || (nmarkers > 0 && !markers)) return fail(MX_ERR_INVALID, "bad argument");
  try {
    PvPlan plan;
    std::string err;
    const int rc = build_pv_plan(markers, nmarkers, sampleRate, n, plan, err);
    if (rc) return fail(rc, "%s", err.c_str());
    return plan.n_out;
  } catch (const std::exception &e) {  // nothing may propagate across the C boundary
    return fail(MX_ERR_NOMEM, "phase-vocoder plan: %s", e.what());
  }
}

int mx_pv_plan(int64_t n, int sampleRate, const mx_marker *markers, int nmarkers, int64_t **apos, double **tf,
               double **rf, int64_t **i0, int64_t *frames, int64_t *nsamples) {
  if (n < 0 || nmarkers < 0 || (nmarkers > 0 && !markers) || !apos || !tf || !rf || !i0 || !frames || !nsamples)
    return fail(MX_ERR_INVALID, "bad argument");
  try {
    PvPlan plan;
    std::string err;
    const int rc = build_pv_plan(markers, nmarkers, sampleRate, n, plan, err);
    if (rc) return fail(rc, "%s", err.c_str());
    const size_t F = plan.apos.size();
    int64_t *pa = (int64_t *)malloc(F * 8), *pi = (int64_t *)malloc((F + 1) * 8);
    double *pt = (double *)malloc(F * 8), *pr = (double *)malloc(F * 8);
    if (!pa || !pi || !pt || !pr) {
      free(pa); free(pi); free(pt); free(pr);
      return fail(MX_ERR_NOMEM, "out of host memory");
    }
    memcpy(pa, plan.apos.data(), F * 8);
    memcpy(pi, plan.i0.data(), (F + 1) * 8);
    memcpy(pt, plan.tf.data(), F * 8);
    memcpy(pr, plan.rf.data(), F * 8);
    *apos = pa; *i0 = pi; *tf = pt; *rf = pr;
    *frames = (int64_t)F;
    *nsamples = plan.n_out;
    return MX_OK;
  } catch (const std::bad_alloc &) {
    return fail(MX_ERR_NOMEM, "out of host memory");
  }
}

int mx_pv_render_dev(mx_ctx *ctx, const mx_audio *a, int sampleRate, const mx_marker *markers, int nmarkers,
                     float *d_pcm_f32, int16_t *d_pcm_i16) {
  if (!ctx || !a || nmarkers < 0 || (nmarkers > 0 && !markers)) return fail(MX_ERR_INVALID, "bad argument");
  if (a->n == 0 || (!d_pcm_f32 && !d_pcm_i16)) return MX_OK;
  try {
    PvPlan plan;
    std::string err;
    int rc = build_pv_plan(markers, nmarkers, sampleRate, a->n, plan, err);
    if (rc) return fail(rc, "%s", err.c_str());
    if (plan.n_out == 0) return MX_OK;
    for (int64_t c : plan.apos)
      if (c < -(int64_t)MX_AUDIO_PAD / 2 || c > a->n + (int64_t)MX_AUDIO_PAD / 2)
        return fail(MX_ERR_INVALID, "a marker maps warped time outside the audio");
    std::lock_guard<std::mutex> plk(ctx->pv_mu);
    ctx->pv_job_active = false;
    PvArgs p;
    rc = pv_prepare(ctx, a, 0.0, 0, (int64_t)plan.apos.size(), false, p, &plan);
    if (rc) return rc;
    p.sample_rate = sampleRate;
    p.pcm_f32 = d_pcm_f32;
    p.pcm_i16 = d_pcm_i16;
    hipError_t e = launch_pv(p, ctx->stream);
    const hipError_t es = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) return fail(MX_ERR_DEVICE, "phase vocoder: %s", hipGetErrorString(e));
    return MX_OK;
  } catch (const std::bad_alloc &) {
    return fail(MX_ERR_NOMEM, "out of host memory");
  }
}

int mx_pv_render(mx_ctx *ctx, const mx_audio *a, int sampleRate, const mx_marker *markers, int nmarkers,
                 float *pcm_f32_out, int16_t *pcm_i16_out) {
  if (!ctx || !a) return fail(MX_ERR_INVALID, "null context or audio handle");
  const int64_t m = mx_pv_render_length(a->n, sampleRate, markers, nmarkers);
  if (m < 0) return (int)m;
  if (m == 0 || (!pcm_f32_out && !pcm_i16_out)) return MX_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  float *d_f = nullptr;
  int16_t *d_i = nullptr;
  hipError_t e = hipSuccess;
  if (pcm_f32_out) e = hipMalloc(&d_f, (size_t)m * sizeof(float));
  if (e == hipSuccess && pcm_i16_out) e = hipMalloc(&d_i, (size_t)m * sizeof(int16_t));
  if (e != hipSuccess) {
    hipFree(d_f); hipFree(d_i);
    return fail(MX_ERR_NOMEM, "device PCM buffers: %s", hipGetErrorString(e));
  }
  int rc = mx_pv_render_dev(ctx, a, sampleRate, markers, nmarkers, d_f, d_i);
  if (rc == MX_OK) {
    if (d_f) e = hipMemcpy(pcm_f32_out, d_f, (size_t)m * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && d_i) e = hipMemcpy(pcm_i16_out, d_i, (size_t)m * sizeof(int16_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(MX_ERR_DEVICE, "PCM download: %s", hipGetErrorString(e));
  }
  hipFree(d_f); hipFree(d_i);
  return rc;
}

// ---- one rank of a multi-GPU phase-vocoder run (the overlap-add seams) ------------------------
// Every rank holds the whole input and takes a contiguous range of the frame axis (boundaries on multiples of 32
// frames = the synthesis workgroups, so the float sums group exactly as in a single-GPU run).  Two small exchanges
// happen outside this library (RCCL / gloo all-gathers in the caller): after stage 1 the per-rank phase totals
// (2048 x {restart, phase}), after stage 2 the seams (2 x 3840 raw partial sums).
int mx_pv_shard_frames(int64_t n, double semitones, int rank, int world, int64_t *frame_lo, int64_t *frame_hi,
                       int64_t *out_lo, int64_t *out_hi) {
  if (n <= 0 || world < 1 || rank < 0 || rank >= world || !(semitones >= -48.0 && semitones <= 48.0))
    return fail(MX_ERR_INVALID, "bad argument");
  const double r = std::pow(2.0, semitones / 12.0);
  const int64_t F = pv_frame_count(n, r);
  int64_t per = (F + world - 1) / world;
  per = (per + 31) / 32 * 32;
  if (per * (world - 1) >= F) return fail(MX_ERR_INVALID, "signal too short for %d ranks (%lld frames)", world, (long long)F);
  const int64_t lo = (int64_t)rank * per, hi = rank == world - 1 ? F : lo + per;
  if (frame_lo) *frame_lo = lo;
  if (frame_hi) *frame_hi = hi;
  if (out_lo) *out_lo = rank == 0 ? 0 : pv_first_output_at(lo * kPvHs, r, n);
  if (out_hi) *out_hi = rank == world - 1 ? n : pv_first_output_at(hi * kPvHs, r, n);
  return MX_OK;
}

int mx_pv_shard_analyze(mx_ctx *ctx, const mx_audio *a, double semitones, int rank, int world, uint32_t *tot_sums_out,
                        uint16_t *tot_org_out) {
  if (!ctx || !a || !tot_sums_out || !tot_org_out) return fail(MX_ERR_INVALID, "bad argument");
  int64_t lo, hi, olo, ohi;
  int rc = mx_pv_shard_frames(a->n, semitones, rank, world, &lo, &hi, &olo, &ohi);
  if (rc) return rc;
  std::lock_guard<std::mutex> plk(ctx->pv_mu);
  ctx->pv_job_active = false;
  rc = pv_prepare(ctx, a, semitones, lo, hi, true, ctx->pv_job);
  if (rc) return rc;
  PvArgs &p = ctx->pv_job;
  p.out_lo = olo;
  p.out_hi = ohi;
  ctx->pv_job_last = rank == world - 1;
  hipError_t e = launch_pv_analyze(p, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(tot_sums_out, p.tot_sums, kPvM * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(tot_org_out, p.tot_org, kPvM * 2, hipMemcpyDeviceToHost, ctx->stream);
  const hipError_t es = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = es;
  if (e != hipSuccess) return fail(MX_ERR_DEVICE, "phase vocoder (analysis): %s", hipGetErrorString(e));
  ctx->pv_job_active = true;
  return MX_OK;
}

int mx_pv_shard_synthesize(mx_ctx *ctx, const uint32_t *carry_in, float *head_out, float *tail_out) {
  if (!ctx || !head_out || !tail_out) return fail(MX_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> plk(ctx->pv_mu);
  if (!ctx->pv_job_active) return fail(MX_ERR_INVALID, "mx_pv_shard_analyze has not run on this context");
  PvArgs &p = ctx->pv_job;
  if (!p.global_first && !carry_in) return fail(MX_ERR_INVALID, "carry_in is required on every rank but the first");
  HIP_TRY(hipSetDevice(ctx->device));
  hipError_t e = hipSuccess;
  p.carry_in = nullptr;
  if (carry_in) {
    e = hipMemcpyAsync(ctx->pv_slot_carry, carry_in, kPvM * 4, hipMemcpyHostToDevice, ctx->stream);
    p.carry_in = reinterpret_cast<const uint32_t *>(ctx->pv_slot_carry);
  }
  if (e == hipSuccess) e = launch_pv_synthesize(p, ctx->stream);
  // the seams, raw: this rank's sums into the N - Hs samples before its first complete hop (halo of workgroup 0;
  // all zero on the first rank, whose first hops are complete) and after its last hop
  if (e == hipSuccess) {
    if (p.global_first) memset(head_out, 0, kPvSeam * 4);
    else e = hipMemcpyAsync(head_out, p.halo, kPvSeam * 4, hipMemcpyDeviceToHost, ctx->stream);
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(tail_out, p.s + (p.frames - p.first) * kPvHs, kPvSeam * 4, hipMemcpyDeviceToHost, ctx->stream);
  const hipError_t es = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = es;
  if (e != hipSuccess) return fail(MX_ERR_DEVICE, "phase vocoder (synthesis): %s", hipGetErrorString(e));
  return MX_OK;
}

int mx_pv_shard_finish(mx_ctx *ctx, const float *prev_tail, const float *next_head, float *pcm_f32_out,
                       int16_t *pcm_i16_out) {
  if (!ctx) return fail(MX_ERR_INVALID, "null context");
  std::lock_guard<std::mutex> plk(ctx->pv_mu);
  if (!ctx->pv_job_active) return fail(MX_ERR_INVALID, "mx_pv_shard_analyze has not run on this context");
  PvArgs &p = ctx->pv_job;
  if ((!p.global_first && !prev_tail) || (!ctx->pv_job_last && !next_head))
    return fail(MX_ERR_INVALID, "a neighbour's seam is missing");
  HIP_TRY(hipSetDevice(ctx->device));
  const int64_t cnt = p.out_hi - p.out_lo;
  float *d_f = nullptr;
  int16_t *d_i = nullptr;
  hipError_t e = hipSuccess;
  if (pcm_f32_out && cnt) e = hipMalloc(&d_f, (size_t)cnt * 4);
  if (e == hipSuccess && pcm_i16_out && cnt) e = hipMalloc(&d_i, (size_t)cnt * 2);
  p.prev_tail = p.next_head = nullptr;
  if (e == hipSuccess && !p.global_first) {
    e = hipMemcpyAsync(ctx->pv_slot_prev_tail, prev_tail, kPvSeam * 4, hipMemcpyHostToDevice, ctx->stream);
    p.prev_tail = reinterpret_cast<const float *>(ctx->pv_slot_prev_tail);
  }
  if (e == hipSuccess && !ctx->pv_job_last) {
    e = hipMemcpyAsync(ctx->pv_slot_next_head, next_head, kPvSeam * 4, hipMemcpyHostToDevice, ctx->stream);
    p.next_head = reinterpret_cast<const float *>(ctx->pv_slot_next_head);
  }
  p.pcm_f32 = d_f;
  p.pcm_i16 = d_i;
  if (e == hipSuccess) e = launch_pv_finish(p, ctx->stream);
  if (e == hipSuccess && d_f) e = hipMemcpyAsync(pcm_f32_out, d_f, (size_t)cnt * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && d_i) e = hipMemcpyAsync(pcm_i16_out, d_i, (size_t)cnt * 2, hipMemcpyDeviceToHost, ctx->stream);
  const hipError_t es = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = es;
  hipFree(d_f);
  hipFree(d_i);
  ctx->pv_job_active = false;
  if (e != hipSuccess) return fail(MX_ERR_DEVICE, "phase vocoder (finish): %s", hipGetErrorString(e));
  return MX_OK;
}

int mx_pv_pitch_shift(mx_ctx *ctx, const mx_audio *a, double semitones, float *pcm_f32_out, int16_t *pcm_i16_out) {
  if (!ctx || !a) return fail(MX_ERR_INVALID, "null context or audio handle");
  if (a->n == 0 || (!pcm_f32_out && !pcm_i16_out)) return MX_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  float *d_f = nullptr;
  int16_t *d_i = nullptr;
  hipError_t e = hipSuccess;
  if (pcm_f32_out) e = hipMalloc(&d_f, (size_t)a->n * sizeof(float));
  if (e == hipSuccess && pcm_i16_out) e = hipMalloc(&d_i, (size_t)a->n * sizeof(int16_t));
  if (e != hipSuccess) {
    hipFree(d_f); hipFree(d_i);
    return fail(MX_ERR_NOMEM, "device PCM buffers: %s", hipGetErrorString(e));
  }
  int rc = mx_pv_pitch_shift_dev(ctx, a, semitones, d_f, d_i);
  if (rc == MX_OK) {
    if (d_f) e = hipMemcpy(pcm_f32_out, d_f, (size_t)a->n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && d_i) e = hipMemcpy(pcm_i16_out, d_i, (size_t)a->n * sizeof(int16_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(MX_ERR_DEVICE, "PCM download: %s", hipGetErrorString(e));
  }
  hipFree(d_f); hipFree(d_i);
  return rc;
}

// ---- time maps -----------------------------------------------------------------
double mx_sample2time(const mx_marker *m, int nm, int sr, int val) { return TimeMap(m, nm, sr, 0).sample2time(val); }
int mx_time2sample(const mx_marker *m, int nm, int sr, double val) { return TimeMap(m, nm, sr, 0).time2sample(val); }
double mx_duration(const mx_marker *m, int nm, int sr, int64_t n) { return TimeMap(m, nm, sr, n).duration(); }
float mx_time2pitchbend(const mx_marker *m, int nm, int sr, int64_t n, double val) {
  return TimeMap(m, nm, sr, n).time2pitchbend(val);
}
void mx_column_range(const mx_marker *m, int nm, int sr, double time, int width, double rangeTime, int *key,
                     int *start, int *end) {
  const TimeMap tm(m, nm, sr, 0);
  const int k = static_cast<int>(time * width / rangeTime);  // spec-cache.cpp:12
  const double st = k * rangeTime / width;                   // spec-cache.cpp:63
  const double pixelSize = rangeTime / width;                // spec-cache.cpp:64
  if (key) *key = k;
  if (start) *start = tm.time2sample(st);                    // spec-cache.cpp:65
  if (end) *end = tm.time2sample(st + pixelSize);
}

// ---- grains + schedule -----------------------------------------------------------
static int export_vectors(const std::vector<int32_t> &s, const std::vector<int32_t> &l, int32_t **starts,
                          int32_t **lens, int64_t *count) {
  const size_t n = s.size();
  int32_t *ps = (int32_t *)malloc(sizeof(int32_t) * std::max<size_t>(n, 1));
  int32_t *pl = (int32_t *)malloc(sizeof(int32_t) * std::max<size_t>(n, 1));
  if (!ps || !pl) { free(ps); free(pl); return fail(MX_ERR_NOMEM, "out of host memory"); }
  if (n) { memcpy(ps, s.data(), n * sizeof(int32_t)); memcpy(pl, l.data(), n * sizeof(int32_t)); }
  *starts = ps; *lens = pl; *count = (int64_t)n;
  return MX_OK;
}

int mx_grains(const float *host_wav, int64_t n, int32_t **starts, int32_t **lens, int64_t *count) {
  if (!starts || !lens || !count || n < 0 || (n > 0 && !host_wav)) return fail(MX_ERR_INVALID, "bad argument");
  try {
    ZcBitmaps zc;
    zc_bitmaps_host(host_wav, n, zc);
    std::vector<int32_t> s, l;
    grains_from_bitmaps(zc, s, l);
    return export_vectors(s, l, starts, lens, count);
  } catch (const std::bad_alloc &) {
    return fail(MX_ERR_NOMEM, "out of host memory");
  }
}

// grow-only device buffer `slot` of the grain chain; caller holds ctx->zc_mu
static hipError_t chain_buf(mx_ctx *ctx, int slot, size_t bytes, void **out) {
  mx_ctx::Stage &st = ctx->chain[slot];
  if (st.cap < bytes) {
    if (st.p) hipFree(st.p);
    st = {};
    const hipError_t e = hipMalloc(&st.p, bytes);
    if (e != hipSuccess) return e;
    st.cap = bytes;
  }
  *out = st.p;
  return hipSuccess;
}

int mx_grain_table_dev(mx_ctx *ctx, const mx_audio *a, int32_t **starts, int32_t **lens, float **firsts, int64_t *count) {
  if (!ctx || !a || !starts || !lens || !count) return fail(MX_ERR_INVALID, "bad argument");
  *starts = *lens = nullptr;
  if (firsts) *firsts = nullptr;
  *count = 0;
  HIP_TRY(hipSetDevice(ctx->device));
  const bool tr = getenv("MELONIX_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto x, auto y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
  const auto t0 = now();
  std::lock_guard<std::mutex> zlk(ctx->zc_mu);
  const int64_t n = a->n;
  const size_t words = (size_t)((n + 63) >> 6);
  uint32_t ngr = 0;
  int32_t *d_s = nullptr, *d_l = nullptr;
  float *d_f = nullptr;
  auto t1 = t0, t2 = t0;
  if (words && n >= 1501) {  // (the reference's size_t arithmetic wraps below 1501 samples, app.cpp:161: no grains)
    void *d7 = nullptr, *d3 = nullptr, *rk = nullptr, *ch = nullptr;
    hipError_t e = chain_buf(ctx, 0, words * 8, &d7);
    if (e == hipSuccess) e = chain_buf(ctx, 1, words * 8, &d3);
    if (e == hipSuccess) e = chain_buf(ctx, 2, grain_rank_scratch_bytes(n), &rk);
    if (e != hipSuccess) return fail(MX_ERR_NOMEM, "grain chain buffers: %s", hipGetErrorString(e));
    HIP_TRY(launch_zc_bitmaps(a->d_padded, n, (uint64_t *)d7, (uint64_t *)d3, ctx->stream));
    HIP_TRY(launch_grain_rank(a->d_padded, n, (const uint64_t *)d7, (const uint64_t *)d3, rk, ctx->stream));
    uint32_t hdr[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(hdr, rk, sizeof hdr, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the node count sizes the lifting tables
    t1 = now();
    int levels;
    uint32_t out_cap;
    size_t bytes;
    grain_chain_sizes(n, hdr[2], &levels, &out_cap, &bytes);
    e = chain_buf(ctx, 3, bytes, &ch);
    if (e != hipSuccess) return fail(MX_ERR_NOMEM, "grain chain tables (%zu bytes): %s", bytes, hipGetErrorString(e));
    HIP_TRY(launch_grain_chain(a->d_padded, n, (const uint64_t *)d7, (const uint64_t *)d3, rk, hdr[2], ch, &d_s, &d_l, &d_f,
                               ctx->stream));
    HIP_TRY(hipMemcpyAsync(hdr, rk, sizeof hdr, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    t2 = now();
    ngr = hdr[1];
    if (ngr > out_cap) return fail(MX_ERR_DEVICE, "grain chain: %u grains exceed the bound %u", ngr, out_cap);
  }
  const size_t m = std::max<size_t>(ngr, 1);
  int32_t *ps = (int32_t *)malloc(m * 4), *pl = (int32_t *)malloc(m * 4);
  float *pf = firsts ? (float *)malloc(m * 4) : nullptr;
  if (!ps || !pl || (firsts && !pf)) {
    free(ps); free(pl); free(pf);
    return fail(MX_ERR_NOMEM, "out of host memory");
  }
  if (ngr) {
    hipError_t e = hipMemcpyAsync(ps, d_s, (size_t)ngr * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pl, d_l, (size_t)ngr * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && pf) e = hipMemcpyAsync(pf, d_f, (size_t)ngr * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      free(ps); free(pl); free(pf);
      return fail(MX_ERR_DEVICE, "grain table download: %s", hipGetErrorString(e));
    }
  }
  if (tr)
    fprintf(stderr, "mx_grain_table_dev: bitmaps + ranks %.2f ms, chain %.2f, download of %u grains %.2f\n", ms(t0, t1),
            ms(t1, t2), ngr, ms(t2, now()));
  *starts = ps;
  *lens = pl;
  if (firsts) *firsts = pf;
  *count = (int64_t)ngr;
  return MX_OK;
}

int mx_grains_dev(mx_ctx *ctx, const mx_audio *a, int32_t **starts, int32_t **lens, int64_t *count) {
  return mx_grain_table_dev(ctx, a, starts, lens, nullptr, count);
}

int mx_schedule_build(const float *host_wav, int64_t n, int sampleRate, const int32_t *grain_starts,
                      const int32_t *grain_lens, int64_t ngrains, const mx_marker *markers, int nmarkers,
                      mx_step **steps, int64_t *nsteps, int64_t *nsamples) {
  return mx_schedule_build_from(host_wav, n, sampleRate, grain_starts, grain_lens, ngrains, markers, nmarkers, 0., -1,
                                steps, nsteps, nsamples, nullptr);
}

static int schedule_common(const float *host_wav, const float *firsts, int64_t n, int sampleRate, const int32_t *grain_starts,
                           const int32_t *grain_lens, int64_t ngrains, const mx_marker *markers, int nmarkers,
                           double cursor0, int64_t need, mx_step **steps, int64_t *nsteps, int64_t *nsamples,
                           double *cursor_end);

int mx_schedule_build_from(const float *host_wav, int64_t n, int sampleRate, const int32_t *grain_starts,
                           const int32_t *grain_lens, int64_t ngrains, const mx_marker *markers, int nmarkers,
                           double cursor0, int64_t need, mx_step **steps, int64_t *nsteps, int64_t *nsamples,
                           double *cursor_end) {
  if (n > 0 && !host_wav) return fail(MX_ERR_INVALID, "bad argument");
  return schedule_common(host_wav, nullptr, n, sampleRate, grain_starts, grain_lens, ngrains, markers, nmarkers, cursor0,
                         need, steps, nsteps, nsamples, cursor_end);
}

int mx_schedule_build_table(int64_t n, int sampleRate, const int32_t *grain_starts, const int32_t *grain_lens,
                            const float *grain_firsts, int64_t ngrains, const mx_marker *markers, int nmarkers,
                            double cursor0, int64_t need, mx_step **steps, int64_t *nsteps, int64_t *nsamples,
                            double *cursor_end) {
  if (ngrains > 0 && !grain_firsts) return fail(MX_ERR_INVALID, "bad argument");
  static const float kNoGrain = 0.f;  // (an empty table: the loop never reads a first sample)
  return schedule_common(nullptr, grain_firsts ? grain_firsts : &kNoGrain, n, sampleRate, grain_starts, grain_lens, ngrains,
                         markers, nmarkers, cursor0, need, steps, nsteps, nsamples, cursor_end);
}

static int schedule_common(const float *host_wav, const float *firsts, int64_t n, int sampleRate, const int32_t *grain_starts,
                           const int32_t *grain_lens, int64_t ngrains, const mx_marker *markers, int nmarkers,
                           double cursor0, int64_t need, mx_step **steps, int64_t *nsteps, int64_t *nsamples,
                           double *cursor_end) {
  if (!steps || !nsteps || !nsamples || n < 0 || ngrains < 0 || nmarkers < 0 ||
      (ngrains > 0 && (!grain_starts || !grain_lens)) || (nmarkers > 0 && !markers))
    return fail(MX_ERR_INVALID, "bad argument");
  const bool tr = getenv("MELONIX_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  for (int64_t g = 0; g < ngrains; ++g)
    if (grain_starts[g] < 0 || grain_lens[g] <= 0 || (int64_t)grain_starts[g] + grain_lens[g] > n)
      return fail(MX_ERR_INVALID, "grain %lld lies outside the audio", (long long)g);
  try {
    std::vector<mx_step> v;
    std::string err;
    int64_t total = 0;
    const auto t1 = std::chrono::steady_clock::now();
    const int rc = build_schedule(host_wav, n, sampleRate, grain_starts, grain_lens, ngrains, markers, nmarkers, v,
                                  total, err, cursor0, need, cursor_end, firsts);
    const auto t2 = std::chrono::steady_clock::now();
    if (rc) return fail(rc, "%s", err.c_str());
    mx_step *p = (mx_step *)malloc(sizeof(mx_step) * std::max<size_t>(v.size(), 1));
    if (!p) return fail(MX_ERR_NOMEM, "out of host memory");
    if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(mx_step));
    if (tr)
      fprintf(stderr, "mx_schedule_build: validate %.2f ms, recurrence %.2f ms (%zu steps), hand-over %.2f ms\n",
              std::chrono::duration<double, std::milli>(t1 - t0).count(),
              std::chrono::duration<double, std::milli>(t2 - t1).count(), v.size(),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
    *steps = p;
    *nsteps = (int64_t)v.size();
    *nsamples = total;
    return MX_OK;
  } catch (const std::bad_alloc &) {
    return fail(MX_ERR_NOMEM, "out of host memory");
  }
}

void mx_free(void *p) { free(p); }

// ---- resynthesis -------------------------------------------------------------------
int mx_resynth_dev(mx_ctx *ctx, const mx_audio *a, const mx_step *d_steps, int64_t nsteps, int64_t nsamples,
                   float *d_pcm_f32, int16_t *d_pcm_i16) {
  if (!ctx || !a || nsteps < 0 || nsamples < 0 || (nsteps > 0 && !d_steps))
    return fail(MX_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  // Samples past the last step's run are zeros (the terminating process() calls append 1500 zeros each: one for an
  // export, ceil(missing/1500) for a playback refill — mx_schedule_build_from).  The device-resident entry point does
  // not see the schedule, so the kernel itself clears [covered, nsamples): its last workgroup knows where the steps end.
  if (nsteps == 0 && nsamples > 0) {
    if (d_pcm_f32) HIP_TRY(hipMemsetAsync(d_pcm_f32, 0, (size_t)nsamples * sizeof(float), ctx->stream));
    if (d_pcm_i16) HIP_TRY(hipMemsetAsync(d_pcm_i16, 0, (size_t)nsamples * sizeof(int16_t), ctx->stream));
  }
  ResynthArgs r{};
  r.audio = a->d_padded;
  r.steps = d_steps;
  r.nsteps = nsteps;
  r.nsamples = nsamples;
  r.pcm_f32 = d_pcm_f32;
  r.pcm_i16 = d_pcm_i16;
  HIP_TRY(launch_resynth(r, ctx->stream));
  return MX_OK;
}

}  // extern "C"

namespace {
// The device half of mx_resynth: checks the schedule's invariants, uploads it, allocates the requested PCM buffers and
// runs the kernel (asynchronously on the context's stream).  The caller downloads and frees.
struct ResynthBuffers {
  mx_step *d_steps = nullptr;
  float *d_f = nullptr;
  int16_t *d_i = nullptr;
  ~ResynthBuffers() { hipFree(d_steps); hipFree(d_f); hipFree(d_i); }
};
int resynth_to_device(mx_ctx *ctx, const mx_audio *a, const mx_step *steps, int64_t nsteps, int64_t nsamples, bool want_f,
                      bool want_i, ResynthBuffers &b) {
  if (!ctx || !a || nsteps < 0 || nsamples < 0 || (nsteps > 0 && !steps)) return fail(MX_ERR_INVALID, "bad argument");
  int64_t covered = 0;
  for (int64_t i = 0; i < nsteps; ++i) {
    const mx_step &s = steps[i];
    if (s.out_offset != covered || s.sz < 0 || s.grain_start < 0 || s.grain_len <= 0 ||
        (int64_t)s.grain_start + s.grain_len > a->n)
      return fail(MX_ERR_INVALID, "step %lld is inconsistent with the schedule invariants", (long long)i);
    covered += s.sz;
  }
  if (covered > nsamples) return fail(MX_ERR_INVALID, "steps emit %lld samples, nsamples is %lld", (long long)covered,
                                      (long long)nsamples);
  HIP_TRY(hipSetDevice(ctx->device));
  hipError_t e = hipSuccess;
  if (nsteps) e = hipMalloc(&b.d_steps, (size_t)nsteps * sizeof(mx_step));
  if (e == hipSuccess && want_f && nsamples) e = hipMalloc(&b.d_f, (size_t)nsamples * sizeof(float));
  if (e == hipSuccess && want_i && nsamples) e = hipMalloc(&b.d_i, (size_t)nsamples * sizeof(int16_t));
  if (e != hipSuccess) return fail(MX_ERR_NOMEM, "device buffers: %s", hipGetErrorString(e));
  if (nsteps)
    if ((e = hipMemcpyAsync(b.d_steps, steps, (size_t)nsteps * sizeof(mx_step), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
      return fail(MX_ERR_DEVICE, "schedule upload: %s", hipGetErrorString(e));
  // anything between the covered run and the tail is zero by definition
  if (b.d_f) hipMemsetAsync(b.d_f + covered, 0, (size_t)(nsamples - covered) * sizeof(float), ctx->stream);
  if (b.d_i) hipMemsetAsync(b.d_i + covered, 0, (size_t)(nsamples - covered) * sizeof(int16_t), ctx->stream);
  return mx_resynth_dev(ctx, a, b.d_steps, nsteps, nsamples, b.d_f, b.d_i);
}
}  // namespace

extern "C" {

int mx_resynth(mx_ctx *ctx, const mx_audio *a, const mx_step *steps, int64_t nsteps, int64_t nsamples,
               float *pcm_f32_out, int16_t *pcm_i16_out) {
  ResynthBuffers b;
  int rc = resynth_to_device(ctx, a, steps, nsteps, nsamples, pcm_f32_out != nullptr, pcm_i16_out != nullptr, b);
  if (rc == MX_OK) {
    hipError_t e = hipSuccess;
    if (b.d_f) e = hipMemcpyAsync(pcm_f32_out, b.d_f, (size_t)nsamples * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && b.d_i)
      e = hipMemcpyAsync(pcm_i16_out, b.d_i, (size_t)nsamples * sizeof(int16_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(MX_ERR_DEVICE, "PCM download: %s", hipGetErrorString(e));
  } else if (ctx) {
    hipStreamSynchronize(ctx->stream);  // nothing of ours may still be in flight when the buffers go
  }
  return rc;
}

int mx_resynth_to_wav(mx_ctx *ctx, const mx_audio *a, const mx_step *steps, int64_t nsteps, int64_t nsamples,
                      const char *path, int sampleRate, int strict_reference_header) {
  if (!ctx || !a || !path) return fail(MX_ERR_INVALID, "bad argument");
  // The PCM never exists as one host buffer: it leaves the device in 16 MiB pieces through two pinned landing
  // buffers, and each piece goes into the file while the next one is in flight.
  ResynthBuffers b;
  int rc = resynth_to_device(ctx, a, steps, nsteps, nsamples, false, true, b);
  WavStream ws;
  if (rc == MX_OK && wav_begin(ws, path, nsamples, sampleRate, strict_reference_header != 0) != MX_OK)
    rc = fail(MX_ERR_IO, "cannot write %s", path);
  if (rc != MX_OK) {
    hipStreamSynchronize(ctx->stream);  // nothing of ours may still be in flight when the buffers go
    return rc;
  }
  constexpr int64_t kPiece = 8 << 20;  // samples
  int16_t *land[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipError_t e = hipSuccess;
  const int64_t pieces = (nsamples + kPiece - 1) / kPiece;
  for (int i = 0; i < 2 && e == hipSuccess && i < pieces; ++i) {
    e = hipHostMalloc((void **)&land[i], (size_t)std::min<int64_t>(kPiece, nsamples) * sizeof(int16_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
  }
  auto piece_len = [&](int64_t k) { return std::min<int64_t>(kPiece, nsamples - k * kPiece); };
  for (int64_t k = 0; k <= pieces && e == hipSuccess; ++k) {
    if (k < pieces) {
      e = hipMemcpyAsync(land[k & 1], b.d_i + k * kPiece, (size_t)piece_len(k) * sizeof(int16_t), hipMemcpyDeviceToHost,
                         ctx->stream);
      if (e == hipSuccess) e = hipEventRecord(ev[k & 1], ctx->stream);
    }
    if (k > 0 && e == hipSuccess) {
      e = hipEventSynchronize(ev[(k - 1) & 1]);
      if (e == hipSuccess) wav_append(ws, land[(k - 1) & 1], piece_len(k - 1));
    }
  }
  const hipError_t es = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = es;
  if (e != hipSuccess) rc = fail(MX_ERR_DEVICE, "PCM download: %s", hipGetErrorString(e));
  for (int i = 0; i < 2; ++i) {
    if (ev[i]) hipEventDestroy(ev[i]);
    if (land[i]) hipHostFree(land[i]);
  }
  if (wav_end(ws) != MX_OK && rc == MX_OK) rc = fail(MX_ERR_IO, "cannot write %s", path);
  return rc;
}

int mx_export_wav(mx_ctx *ctx, const float *host_wav, int64_t n, int sampleRate, const mx_marker *markers,
                  int nmarkers, const char *path, int strict_reference_header) {
  if (!ctx || !path || n < 0 || (n > 0 && !host_wav)) return fail(MX_ERR_INVALID, "bad argument");
  const bool tr = getenv("MELONIX_TIMING") != nullptr;
  using clk = std::chrono::steady_clock;
  auto ms = [](clk::time_point x, clk::time_point y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
  const auto t0 = clk::now();
  mx_audio *a = nullptr;
  int rc = mx_audio_upload(ctx, host_wav, n, &a);
  if (rc) return rc;
  const auto t1 = clk::now();
  int32_t *gs = nullptr, *gl = nullptr;
  int64_t ng = 0, nsteps = 0, nsamples = 0;
  mx_step *steps = nullptr;
  rc = mx_grains_dev(ctx, a, &gs, &gl, &ng);
  const auto t2 = clk::now();
  if (rc == MX_OK) rc = mx_schedule_build(host_wav, n, sampleRate, gs, gl, ng, markers, nmarkers, &steps, &nsteps, &nsamples);
  const auto t3 = clk::now();
  if (rc == MX_OK) rc = mx_resynth_to_wav(ctx, a, steps, nsteps, nsamples, path, sampleRate, strict_reference_header);
  const auto t4 = clk::now();
  const auto t5 = t4;
  mx_free(steps); mx_free(gs); mx_free(gl);
  mx_audio_free(ctx, a);
  if (tr)
    fprintf(stderr, "mx_export_wav: upload %.2f ms, grains %.2f, schedule %.2f, resynth + D2H + file %.2f, free %.2f\n",
            ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t5, clk::now()));
  return rc;
}

// ---- waveform pyramid ---------------------------------------------------------------------
int mx_minmax_pyramid_dev(mx_ctx *ctx, const mx_audio *a, float *d_picks, int64_t *counts_out, int *nlevels) {
  if (!ctx || !a || !counts_out || !nlevels || (a->n > 2 && !d_picks)) return fail(MX_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(launch_picks(a->d_padded, a->n, d_picks, counts_out, nlevels, ctx->stream));
  return MX_OK;
}

int mx_minmax_pyramid(mx_ctx *ctx, const mx_audio *a, float *picks_out, int64_t *counts_out, int *nlevels) {
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
}

void mx_minmax_range(const float *host_wav, int64_t n, const float *picks, const int64_t *counts, int nlevels, int start,
                     int end, float *mn, float *mx) {
  float a = 0.f, b = 0.f;
  if (host_wav && picks && counts) minmax_from_range(host_wav, n, picks, counts, nlevels, start, end, a, b);
  if (mn) *mn = a;
  if (mx) *mx = b;
}

int mx_save_wav(const char *path, const int16_t *pcm, int64_t m, int sampleRate, int strict_reference_header) {
  const int rc = write_wav(path, pcm, m, sampleRate, strict_reference_header != 0);
  if (rc == MX_ERR_INVALID) return fail(rc, "bad argument");
  if (rc == MX_ERR_IO) return fail(rc, "cannot write %s", path);
  return rc;
}

}  // extern "C"
