// capi_pv.cpp — the build-defined phase-vocoder pitch shift (no reference counterpart).
// One unit of the C-ABI implementation behind include/melonix_amd.h (see capi_internal.h).  There is no CPU compute path:
// every transform entry point needs a live gfx950 device and fails with MX_ERR_DEVICE otherwise.
#include "capi_internal.h"
#include "stft_tables.h"

using namespace mx;

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

extern "C" {

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

}  // extern "C"
