// capi_resynth.cpp — time maps, grains, the export schedule, granular resynthesis, WAV export.
// One unit of the C-ABI implementation behind include/melonix_amd.h (see capi_internal.h).  There is no CPU compute path:
// every transform entry point needs a live gfx950 device and fails with MX_ERR_DEVICE otherwise.
#include "capi_internal.h"

using namespace mx;

extern "C" {

// ---- time maps -----------------------------------------------------------------
double mx_sample2time(const mx_marker *m, int nm, int sr, int val) {
  return mx_guard_or<double>(std::nan(""), [&]() -> double {
    return TimeMap(m, nm, sr, 0).sample2time(val);
  });
}
int mx_time2sample(const mx_marker *m, int nm, int sr, double val) {
  return mx_guard([&]() -> int {
    return TimeMap(m, nm, sr, 0).time2sample(val);
  });
}
double mx_duration(const mx_marker *m, int nm, int sr, int64_t n) {
  return mx_guard_or<double>(std::nan(""), [&]() -> double {
    return TimeMap(m, nm, sr, n).duration();
  });
}
float mx_time2pitchbend(const mx_marker *m, int nm, int sr, int64_t n, double val) {
  return mx_guard_or<float>(std::nanf(""), [&]() -> float {
    return TimeMap(m, nm, sr, n).time2pitchbend(val);
  });
}
void mx_column_range(const mx_marker *m, int nm, int sr, double time, int width, double rangeTime, int *key,
                     int *start, int *end) {
  mx_guard_void([&] {
    const TimeMap tm(m, nm, sr, 0);
    const int k = static_cast<int>(time * width / rangeTime);  // spec-cache.cpp:12
    const double st = k * rangeTime / width;                   // spec-cache.cpp:63
    const double pixelSize = rangeTime / width;                // spec-cache.cpp:64
    if (key) *key = k;
    if (start) *start = tm.time2sample(st);                    // spec-cache.cpp:65
    if (end) *end = tm.time2sample(st + pixelSize);
  });
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
  return mx_guard([&]() -> int {
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
  });
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
  return mx_guard([&]() -> int {
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
  });
}

int mx_grains_dev(mx_ctx *ctx, const mx_audio *a, int32_t **starts, int32_t **lens, int64_t *count) {
  return mx_guard([&]() -> int {
    return mx_grain_table_dev(ctx, a, starts, lens, nullptr, count);
  });
}

int mx_schedule_build(const float *host_wav, int64_t n, int sampleRate, const int32_t *grain_starts,
                      const int32_t *grain_lens, int64_t ngrains, const mx_marker *markers, int nmarkers,
                      mx_step **steps, int64_t *nsteps, int64_t *nsamples) {
  return mx_guard([&]() -> int {
    return mx_schedule_build_from(host_wav, n, sampleRate, grain_starts, grain_lens, ngrains, markers, nmarkers, 0., -1,
                                  steps, nsteps, nsamples, nullptr);
  });
}

static int schedule_common(const float *host_wav, const float *firsts, int64_t n, int sampleRate, const int32_t *grain_starts,
                           const int32_t *grain_lens, int64_t ngrains, const mx_marker *markers, int nmarkers,
                           double cursor0, int64_t need, mx_step **steps, int64_t *nsteps, int64_t *nsamples,
                           double *cursor_end);

int mx_schedule_build_from(const float *host_wav, int64_t n, int sampleRate, const int32_t *grain_starts,
                           const int32_t *grain_lens, int64_t ngrains, const mx_marker *markers, int nmarkers,
                           double cursor0, int64_t need, mx_step **steps, int64_t *nsteps, int64_t *nsamples,
                           double *cursor_end) {
  return mx_guard([&]() -> int {
    if (n > 0 && !host_wav) return fail(MX_ERR_INVALID, "bad argument");
    return schedule_common(host_wav, nullptr, n, sampleRate, grain_starts, grain_lens, ngrains, markers, nmarkers, cursor0,
                           need, steps, nsteps, nsamples, cursor_end);
  });
}

int mx_schedule_build_table(int64_t n, int sampleRate, const int32_t *grain_starts, const int32_t *grain_lens,
                            const float *grain_firsts, int64_t ngrains, const mx_marker *markers, int nmarkers,
                            double cursor0, int64_t need, mx_step **steps, int64_t *nsteps, int64_t *nsamples,
                            double *cursor_end) {
  return mx_guard([&]() -> int {
    if (ngrains > 0 && !grain_firsts) return fail(MX_ERR_INVALID, "bad argument");
    static const float kNoGrain = 0.f;  // (an empty table: the loop never reads a first sample)
    return schedule_common(nullptr, grain_firsts ? grain_firsts : &kNoGrain, n, sampleRate, grain_starts, grain_lens, ngrains,
                           markers, nmarkers, cursor0, need, steps, nsteps, nsamples, cursor_end);
  });
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


// ---- resynthesis -------------------------------------------------------------------
int mx_resynth_dev(mx_ctx *ctx, const mx_audio *a, const mx_step *d_steps, int64_t nsteps, int64_t nsamples,
                   float *d_pcm_f32, int16_t *d_pcm_i16) {
  return mx_guard([&]() -> int {
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
  });
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
  return mx_guard([&]() -> int {
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
  });
}

int mx_resynth_to_wav(mx_ctx *ctx, const mx_audio *a, const mx_step *steps, int64_t nsteps, int64_t nsamples,
                      const char *path, int sampleRate, int strict_reference_header) {
  return mx_guard([&]() -> int {
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
  });
}

int mx_export_wav(mx_ctx *ctx, const float *host_wav, int64_t n, int sampleRate, const mx_marker *markers,
                  int nmarkers, const char *path, int strict_reference_header) {
  return mx_guard([&]() -> int {
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
  });
}

int mx_save_wav(const char *path, const int16_t *pcm, int64_t m, int sampleRate, int strict_reference_header) {
  return mx_guard([&]() -> int {
    const int rc = write_wav(path, pcm, m, sampleRate, strict_reference_header != 0);
    if (rc == MX_ERR_INVALID) return fail(rc, "bad argument");
    if (rc == MX_ERR_IO) return fail(rc, "cannot write %s", path);
    return rc;
  });
}

}  // extern "C"
