// capi_ctx.cpp — context, streams, scratch, pinned memory, audio handles, the error slot, free().
// One unit of the C-ABI implementation behind include/melonix_amd.h (see capi_internal.h).  There is no CPU compute path:
// every transform entry point needs a live gfx950 device and fails with MX_ERR_DEVICE otherwise.
#include "capi_internal.h"
#include "stft_tables.h"

using namespace mx;

namespace mx {
namespace {
thread_local char g_err[512] = "";  // (no std::string: recording an error must not allocate — mx_guard's handlers call fail)
}
int fail(int code, const char *fmt, ...) noexcept {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

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
                mx_pitch *d_pitch, uint8_t *d_rgb, float cmap_k, int run_length) {
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

}  // namespace mx

extern "C" {

const char *mx_last_error(void) { return g_err; }
// (MX_SRC_SHA: melonix_amd/build.py source_sha() — 12 hex digits of the sha1 over the library's sources and compile flags)
#ifndef MX_SRC_SHA
#define MX_SRC_SHA "unknown"
#endif
const char *mx_version(void) { return "melonix_amd 0.1.0 gfx950 src:" MX_SRC_SHA; }

int mx_ctx_create(int device, mx_ctx **out) {
  return mx_guard([&]() -> int {
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
  });
}

void mx_ctx_destroy(mx_ctx *ctx) {
  mx_guard_void([&] {
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
    pv_release(ctx);
    hipStreamDestroy(ctx->own_stream);
    delete ctx;
  });
}

int mx_ctx_set_stream(mx_ctx *ctx, void *hip_stream) {
  return mx_guard([&]() -> int {
    if (!ctx) return fail(MX_ERR_INVALID, "null context");
    ctx->stream = (hipStream_t)hip_stream;  // NULL is the HIP null stream (torch's default stream)
    return MX_OK;
  });
}

int mx_ctx_use_own_stream(mx_ctx *ctx) {
  return mx_guard([&]() -> int {
    if (!ctx) return fail(MX_ERR_INVALID, "null context");
    ctx->stream = ctx->own_stream;
    return MX_OK;
  });
}

int mx_ctx_synchronize(mx_ctx *ctx) {
  return mx_guard([&]() -> int {
    if (!ctx) return fail(MX_ERR_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MX_OK;
  });
}

int mx_ctx_release_scratch(mx_ctx *ctx) {
  return mx_guard([&]() -> int {
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
      pv_release(ctx);  // (a staged multi-GPU job lives in that arena: it ends here)
      ctx->pv_budget_auto = 0;  // (the automatic budget is taken again from what is free at the next first use)
    ctx->pv_rec_full = false;
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
  });
}

int mx_pinned_alloc(mx_ctx *ctx, size_t bytes, void **out) {
  return mx_guard([&]() -> int {
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
  });
}

void mx_pinned_free(mx_ctx *ctx, void *p) {
  mx_guard_void([&] {
    if (!p) return;
    if (ctx) hipSetDevice(ctx->device);
    hipHostFree(p);
  });
}

int mx_ctx_set_frames_per_block(mx_ctx *ctx, int g) {  // tuning knob (bench sweeps)
  return mx_guard([&]() -> int {
    if (!ctx || g < 0) return fail(MX_ERR_INVALID, "bad argument");
    ctx->frames_per_block = g;
    return MX_OK;
  });
}

// ---- audio ------------------------------------------------------------------
int mx_audio_upload(mx_ctx *ctx, const float *host_wav, int64_t n, mx_audio **out) {
  return mx_guard([&]() -> int {
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
  });
}

int mx_audio_wrap_device(mx_ctx *ctx, const float *d_padded, int64_t n, mx_audio **out) {
  return mx_guard([&]() -> int {
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
  });
}

int64_t mx_audio_length(const mx_audio *a) {
  return mx_guard([&]() -> int64_t {
    return a ? a->n : -1;
  });
}

int mx_audio_free(mx_ctx *ctx, mx_audio *a) {
  return mx_guard([&]() -> int {
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
  });
}
void mx_free(void *p) {
  mx_guard_void([&] {
    free(p);
  });
}

}  // extern "C"
