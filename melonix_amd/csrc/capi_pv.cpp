// capi_pv.cpp — the build-defined phase-vocoder pitch shift (no reference counterpart).
// One unit of the C-ABI implementation behind include/melonix_amd.h (see capi_internal.h).  There is no CPU compute path:
// every transform entry point needs a live gfx950 device and fails with MX_ERR_DEVICE otherwise.
//
// The vocoder works inside a work arena with a memory BUDGET (round 6: mx_pv_set_arena_budget / MELONIX_PV_ARENA_MB; the
// default is a quarter of what the device has free at the context's first phase-vocoder call).  A call whose frames fit the
// budget is ONE chunk: its spectra stay resident between analysis and synthesis (one slot, 22 KiB per frame with the compact
// record regions below: 17.9 GB for an hour at +3 st) and a rank of a multi-GPU run analyses its frames once.  What does not fit is walked CHUNK BY CHUNK, two
// slots alternating, with the longest chunk the budget holds (round 5: a fixed 32768 frames whatever was free; rounds 1-4 laid
// the whole signal out at 41 KiB per frame and an 8-hour signal did not fit the GPU).  What one chunk hands the next is what one rank of a
// multi-GPU run hands its neighbour (pv_kernels.hip, mx_pv_shard_*): the frame before the chunk is analysed again as its
// row 0, the dense offset row behind the chunk's last frame is the next chunk's carry_in, and the N - Hs samples across
// the boundary are the left chunk's raw tail plus the right chunk's raw head.  Chunks start on multiples of 32 frames
// (= the synthesis workgroups), so every float sum groups exactly as in one launch over the whole signal: outputs are
// bit-identical whatever the chunk length (tests/test_pv.py::test_gpu_chunked_equals_whole).
//
//   the context's stream   the two big kernels, one at a time:  A(0) A(1) S(0) A(2) S(1) A(3) S(2) ...
//   a side stream          everything small, beside an analysis: while A(k + 1) runs, the first-frame records, chunk maps
//                          and offsets of chunk k (pv_heads, pv_lock_walk, pv_lock_chunks), then pv_fixup + pv_resample of
//                          chunk k - 2
// Two slots of spectra + records alternate (chunk k + 1's analysis follows chunk k - 1's synthesis on the stream); the
// stretched signal and the workgroup halos of a chunk live in a ring of four.  Why nothing runs beside a synthesis, and
// why the two big kernels do not overlap each other: profiles/timeline_r05_pv_pipeline.log.
#include "capi_internal.h"
#include "stft_tables.h"

using namespace mx;

namespace mx {

namespace {
constexpr int kPvN = 4096, kPvM = kPvN / 2, kPvHs = 256, kPvSeam = kPvN - kPvHs;
constexpr int64_t kPvMaxChunk = 1 << 22;
constexpr int kPvSlots = 2;  // (three or four buy nothing: profiles/timeline_r05_pv_pipeline.log)
constexpr int kPvPlanRing = 4;  // chunk k + 3's plan rows are written while chunk k - 1's are long read
constexpr int kPvOutRing = 4;  // chunk k's synthesis writes while chunk k - 2's fix-up reads k - 2, k - 1 (head) and k - 3 (boundary)
// Peak records: every analysis workgroup packs its frames' records into a region of its own of kPvRecPerFrame x (its frames)
// entries — a quarter of the 2048 a frame can have (an impulse): sweeps and music have tens to a few hundred peaks per frame,
// white noise ~410.  A run whose signal does not fit raises the overflow flag and is repeated, once, with full regions
// (kPvM per frame: cannot overflow); the context then stays with those until its scratch is released.
constexpr int kPvRecPerFrame = 512;
constexpr int kPvMinScan = 64;              // frames per scan chunk of the phase recurrence, at least
constexpr int64_t kPvMaxScanChunks = 1536;  // one round of row-walking workgroups, six per CU

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
}  // namespace

// What an arena is made for: chunks of C frames; `slots` sets of spectra + records (one: the call is a single chunk and
// its rows stay resident; two: chunks alternate), and the rings of the pipeline (one entry each where there is one chunk).
struct PvShape {
  int64_t C = 0;
  int slots = kPvSlots, outs = kPvOutRing, plans = kPvPlanRing;
  int rpf = kPvRecPerFrame;  // record capacity per frame of an analysis workgroup's region (kPvM: full)
  bool operator==(const PvShape &o) const { return C == o.C && slots == o.slots && outs == o.outs && plans == o.plans && rpf == o.rpf; }
};
constexpr PvShape pv_chunked(int64_t C, int rpf) { return PvShape{C, kPvSlots, kPvOutRing, kPvPlanRing, rpf}; }
constexpr PvShape pv_resident(int64_t C, int rpf) { return PvShape{C, 1, 1, 1, rpf}; }

struct PvPipe {
  PvShape shape;
  int64_t C = 0;  // frames per chunk (a multiple of 32); a slot has room for C + 32 frames and the row before them
  char *base = nullptr;
  size_t bytes = 0;
  // constants: the two windows, the split twiddles of the inverse transform
  float *hann = nullptr, *hann_scaled = nullptr;
  float2 *wsplit = nullptr;
  struct Plan {  // a chunk's analysis plan (positions, hops, stretch factors): written three chunks ahead, a ring of its own
    int64_t *apos;
    uint32_t *hop;
    double *hratio;
  } plan[kPvPlanRing] = {};
  struct Slot {  // what the analysis of a chunk leaves and its synthesis reads
    float2 *xrows;
    uint2 *recs;
    uint32_t *pkmap, *pkcount;
    float *fthr;
    uint32_t *chunk_sums, *group_sums, *tot_sums;
    uint16_t *chunk_org, *group_org, *tot_org;
  } slot[kPvSlots] = {};
  int NS = kPvSlots, NOUT = kPvOutRing, NPLAN = kPvPlanRing;  // the entries of `shape` (slot / out / plan of chunk k: k % N..)
  struct Out {  // what the synthesis of a chunk leaves and the fix-up / resampler read (+ the resampler's plan rows)
    float *halo, *s;
    double *tf, *rf;
    int64_t *i0;
  } out[kPvOutRing] = {};
  // [1] raised by an analysis whose record regions are too small for the signal: page-locked host memory the kernels write
  // through (once, on the rare overflow) and the host reads behind the call's synchronisation without another API call
  uint32_t *rec_overflow = nullptr;
  uint32_t *carry[2];  // the dense offset row behind chunk k's last frame: carry[k & 1]
  // one rank of a multi-GPU run: what it gets from its neighbours and owes them
  uint32_t *carry_in = nullptr;
  float *prev_tail = nullptr, *next_head = nullptr, *head_raw = nullptr, *tail_raw = nullptr, *edge_head = nullptr, *edge_tail = nullptr;
  hipStream_t ss = nullptr, sf = nullptr;  // the side streams: the recurrence; fix-up + resampling
  hipEvent_t ev_begin = nullptr, ev_fin = nullptr, ev_an[kPvSlots] = {}, ev_lock[kPvSlots] = {}, ev_syn[kPvSlots] = {};
  int64_t last_chunks = 0;  // chunks of the last run (mx_pv_last_chunks)
  // the staged job between mx_pv_shard_analyze and _finish
  struct Shard {
    bool active = false, first = false, last = false, single = false;
    int rank = 0, world = 1;
    const mx_audio *a = nullptr;
    double semitones = 0., r = 1.;
    int64_t F_lo = 0, F_hi = 0, out_lo = 0, out_hi = 0;
    bool synthesized = false;
    int64_t head_hi = 0, tail_lo = 0;  // the outputs [out_lo, head_hi) and [tail_lo, out_hi) wait for the neighbours' seams
    // the rank's outputs between stage 2 and stage 3 (device): the library's own buffers (host-pointer entry points, freed
    // by pv_shard_drop) or the caller's (_dev entry points)
    bool own_pcm = false;
    float *d_f = nullptr;
    int16_t *d_i = nullptr;
  } job;
};

namespace {

size_t pv_layout(PvPipe &p, const PvShape &sh, char *base) {
  const int64_t C = sh.C;
  const int64_t rows = C + 32 + 1;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return base ? base + o : nullptr;
  };
  p.hann = reinterpret_cast<float *>(take(kPvN * 4));
  p.hann_scaled = reinterpret_cast<float *>(take(kPvN * 4));
  p.wsplit = reinterpret_cast<float2 *>(take(kPvM * 8));
  // (a short chunk cuts its frame axis into fewer scan chunks than the cap: pv_run's scan_chunk is at least kPvMinScan frames)
  const size_t nmaps = (size_t)std::min<int64_t>(kPvMaxScanChunks, (rows + kPvMinScan - 1) / kPvMinScan) + 1, ngroups = (nmaps + 31) / 32;
  for (int i = 0; i < sh.plans; ++i) {
    PvPipe::Plan &pl = p.plan[i];
    pl.apos = reinterpret_cast<int64_t *>(take((size_t)rows * 8));
    pl.hop = reinterpret_cast<uint32_t *>(take((size_t)rows * 4));
    pl.hratio = reinterpret_cast<double *>(take((size_t)rows * 8));
  }
  for (int si = 0; si < sh.slots; ++si) {
    PvPipe::Slot &sl = p.slot[si];
    sl.xrows = reinterpret_cast<float2 *>(take((size_t)rows * kPvM * 8));
    // the record pool: one region per analysis workgroup (8 or 16 frames: rows rounded up to 16 covers either cut), rpf entries
    // per frame; + one row of slack (a walk's lanes past a row's count read entries nobody wrote — behind the last region too)
    sl.recs = reinterpret_cast<uint2 *>(take(((size_t)(rows + 16) * (size_t)sh.rpf + kPvM) * 8));
    sl.pkmap = reinterpret_cast<uint32_t *>(take((size_t)rows * (kPvM / 32) * 4));
    sl.pkcount = reinterpret_cast<uint32_t *>(take((size_t)rows * 4));
    sl.fthr = reinterpret_cast<float *>(take((size_t)rows * 4));
    sl.chunk_sums = reinterpret_cast<uint32_t *>(take(nmaps * kPvM * 4));
    sl.chunk_org = reinterpret_cast<uint16_t *>(take(nmaps * kPvM * 2));
    sl.group_sums = reinterpret_cast<uint32_t *>(take(ngroups * kPvM * 4));
    sl.group_org = reinterpret_cast<uint16_t *>(take(ngroups * kPvM * 2));
    sl.tot_sums = reinterpret_cast<uint32_t *>(take(kPvM * 4));
    sl.tot_org = reinterpret_cast<uint16_t *>(take(kPvM * 2));
  }
  for (int i = 0; i < sh.outs; ++i) {
    PvPipe::Out &o = p.out[i];
    o.halo = reinterpret_cast<float *>(take((size_t)pv_halo_floats(C + 32) * 4));
    o.s = reinterpret_cast<float *>(take(((size_t)(C + 32) * kPvHs + kPvN + 8) * 4));
    o.tf = reinterpret_cast<double *>(take((size_t)rows * 8));
    o.rf = reinterpret_cast<double *>(take((size_t)rows * 8));
    o.i0 = reinterpret_cast<int64_t *>(take((size_t)(rows + 1) * 8));
  }
  for (auto &c : p.carry) c = reinterpret_cast<uint32_t *>(take(kPvM * 4));
  p.carry_in = reinterpret_cast<uint32_t *>(take(kPvM * 4));
  p.prev_tail = reinterpret_cast<float *>(take(kPvSeam * 4));
  p.next_head = reinterpret_cast<float *>(take(kPvSeam * 4));
  p.head_raw = reinterpret_cast<float *>(take(kPvSeam * 4));
  p.tail_raw = reinterpret_cast<float *>(take(kPvSeam * 4));
  p.edge_head = reinterpret_cast<float *>(take((kPvSeam + 8) * 4));
  p.edge_tail = reinterpret_cast<float *>(take((kPvSeam + 8) * 4));
  return off;
}
size_t pv_shape_bytes(const PvShape &sh) {
  PvPipe tmp;
  return pv_layout(tmp, sh, nullptr);
}

void pv_shard_drop(PvPipe &p) {
  if (p.job.own_pcm) {
    hipFree(p.job.d_f);
    hipFree(p.job.d_i);
  }
  p.job = PvPipe::Shard{};
}

// ---- the arena's policy -------------------------------------------------------------------------------------------------
// The budget: mx_pv_set_arena_budget, else MELONIX_PV_ARENA_MB, else a quarter of what the device had free when the context
// first needed an arena (taken once and kept until mx_ctx_release_scratch: a budget that followed the free memory call by call
// would rebuild the arena call by call).
int pv_budget(mx_ctx *ctx, size_t *out) {
  if (ctx->pv_budget_bytes > 0) {
    *out = (size_t)ctx->pv_budget_bytes;
    return MX_OK;
  }
  if (const char *e = getenv("MELONIX_PV_ARENA_MB")) {
    const long long mb = atoll(e);
    if (mb > 0) {
      *out = (size_t)mb << 20;
      return MX_OK;
    }
  }
  if (ctx->pv_budget_auto <= 0) {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (ctx->pv) free_b += ctx->pv->bytes;  // (what the context holds already counts as available to it)
    ctx->pv_budget_auto = (int64_t)std::max<size_t>(free_b / 4, (size_t)64 << 20);
  }
  *out = (size_t)ctx->pv_budget_auto;
  return MX_OK;
}

// The shape of the arena a call over `frames` frames wants.  An explicit chunk length (mx_pv_set_chunk_frames /
// MELONIX_PV_CHUNK_FRAMES: tests that want many chunk boundaries in a short signal) is taken as it is, two slots; otherwise one
// resident chunk if the budget holds the call's frames, else the longest chunks (multiples of 32 frames) two slots of which
// fit the budget.
int pv_shape_for(mx_ctx *ctx, int64_t frames, PvShape *out) {
  int64_t C = ctx->pv_chunk_frames;
  if (C <= 0)
    if (const char *e = getenv("MELONIX_PV_CHUNK_FRAMES")) C = atoll(e);
  // (MELONIX_PV_FULL_RECORDS=1: full-size regions from the start — the A/B of the compact layout, tests/test_pv.py)
  const char *full_env = getenv("MELONIX_PV_FULL_RECORDS");
  const int rpf = (ctx->pv_rec_full || (full_env && full_env[0] == '1')) ? kPvM : kPvRecPerFrame;
  if (C > 0) {
    *out = pv_chunked(std::min<int64_t>(kPvMaxChunk, (C + 31) / 32 * 32), rpf);
    return MX_OK;
  }
  size_t budget = 0;
  const int rc = pv_budget(ctx, &budget);
  if (rc) return rc;
  const int64_t Fr = std::max<int64_t>(32, (frames + 31) / 32 * 32);
  if (pv_shape_bytes(pv_resident(Fr, rpf)) <= budget) {
    *out = pv_resident(Fr, rpf);
    return MX_OK;
  }
  // bytes are affine in C up to the 256-byte roundings: solve, then step down onto the budget
  const size_t b0 = pv_shape_bytes(pv_chunked(32, rpf)), b1 = pv_shape_bytes(pv_chunked(32 + 32 * 1024, rpf));
  if (b0 > budget)
    return fail(MX_ERR_NOMEM, "phase-vocoder arena budget of %zu MiB is below the %zu MiB the smallest chunks need", budget >> 20, (b0 >> 20) + 1);
  const double per32 = (double)(b1 - b0) / 1024.0;
  C = 32 + 32 * (int64_t)((double)(budget - b0) / per32);
  C = std::min<int64_t>(kPvMaxChunk, std::max<int64_t>(32, C));
  while (C > 32 && pv_shape_bytes(pv_chunked(C, rpf)) > budget) C -= 32;
  *out = pv_chunked(C, rpf);
  return MX_OK;
}

// The pipe of the context for a call over `frames` frames, built (or rebuilt in another shape) on demand.  An arena that
// holds the call in one chunk is kept whatever it was made for; so is a chunked one of the wanted chunk length.  Caller holds
// ctx->pv_mu.
int pv_pipe(mx_ctx *ctx, int64_t frames, PvPipe **out) {
  HIP_TRY(hipSetDevice(ctx->device));  // HIP's current device is per thread
  PvShape want;
  int rc = pv_shape_for(ctx, frames, &want);
  if (rc) return rc;
  const bool pinned = want.slots == kPvSlots && (ctx->pv_chunk_frames > 0 || getenv("MELONIX_PV_CHUNK_FRAMES"));
  if (ctx->pv) {
    const PvShape &have = ctx->pv->shape;
    const bool keep = pinned ? have == want : ((have.C >= (frames + 31) / 32 * 32 && have.rpf == want.rpf) || have == want);
    if (keep) {
      *out = ctx->pv;
      return MX_OK;
    }
  }
  const int64_t C = want.C;
  pv_release(ctx);
  std::unique_ptr<PvPipe> p(new (std::nothrow) PvPipe());
  if (!p) return fail(MX_ERR_NOMEM, "out of host memory");
  p->shape = want;
  p->C = C;
  p->NS = want.slots;
  p->NOUT = want.outs;
  p->NPLAN = want.plans;
  p->bytes = pv_layout(*p, want, nullptr);
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < p->bytes)
    return fail(MX_ERR_NOMEM, "phase-vocoder work arena: %zu MiB needed for chunks of %lld frames, %zu MiB free", p->bytes >> 20,
                (long long)C, free_b >> 20);
  void *mem = nullptr;
  const hipError_t em = hipMalloc(&mem, p->bytes);
  if (em != hipSuccess) return fail(MX_ERR_NOMEM, "phase-vocoder work arena (%zu MiB): %s", p->bytes >> 20, hipGetErrorString(em));
  p->base = static_cast<char *>(mem);
  pv_layout(*p, want, p->base);
  ctx->pv = p.release();
  PvPipe &q = *ctx->pv;
  hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&q.rec_overflow), 64, hipHostMallocDefault);
  if (e == hipSuccess) *q.rec_overflow = 0u;
  if (e == hipSuccess) {
    // (what gets the side stream's small kernels through beside a transform is their WAVE priority — s_setprio in the
    // kernels: 0.5 ms per hour; the queue's priority measured nothing either way and is left at the default)
    e = hipStreamCreateWithFlags(&q.ss, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&q.sf, hipStreamNonBlocking);
  }
  hipEvent_t *const evs[] = {&q.ev_begin,  &q.ev_fin,     &q.ev_an[0],  &q.ev_an[1],
                             &q.ev_lock[0], &q.ev_lock[1], &q.ev_syn[0], &q.ev_syn[1]};
  static_assert(kPvSlots == 2, "the event list above names both slots");
  for (hipEvent_t *ev : evs)
    if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
  // the constants (host tables on the way: a failed allocation must not leave a half-built pipe behind — the next call would
  // find an arena of the wanted shape and use it)
  try {
    std::vector<float> hann((size_t)kPvN), hann_sc((size_t)kPvN);
    for (int j = 0; j < kPvN; ++j) {
      hann[(size_t)j] = (float)(0.5 - 0.5 * std::cos(2.0 * 3.14159265358979323846 * j / kPvN));
      hann_sc[(size_t)j] = hann[(size_t)j] * fold_scale(kPvN);  // exact: a power of two
    }
    std::vector<float2> wsplit((size_t)kPvM);
    for (int c = 0; c < kPvM; ++c) {
      const double ang = 2.0 * 3.14159265358979323846 * c / kPvN;
      wsplit[(size_t)c] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    if (e == hipSuccess) e = hipMemcpy(q.hann, hann.data(), kPvN * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(q.hann_scaled, hann_sc.data(), kPvN * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(q.wsplit, wsplit.data(), (size_t)kPvM * 8, hipMemcpyHostToDevice);
  } catch (...) {
    pv_release(ctx);
    throw;
  }
  if (e != hipSuccess) {
    pv_release(ctx);
    return fail(MX_ERR_DEVICE, "phase vocoder setup: %s", hipGetErrorString(e));
  }
  *out = ctx->pv;
  return MX_OK;
}

// One run of the pipeline: the frames [F_lo, F_hi) of a signal of F frames.
struct PvRun {
  const mx_audio *a = nullptr;
  double r = 1.;                 // constant ratio (ignored with a plan)
  const PvPlan *plan = nullptr;  // marker-driven variant (whole signal only)
  const std::vector<uint32_t> *plan_hop = nullptr;
  const std::vector<double> *plan_hratio = nullptr;
  int sample_rate = 0;
  int64_t F_lo = 0, F_hi = 0;
  const uint32_t *carry_in = nullptr;  // device; null where the run starts at frame 0
  bool totals_only = false;            // stage 1 of a rank: analysis and maps, nothing synthesised
  uint32_t *totmaps_sums = nullptr;    // [chunks][M] per-chunk total maps (totals_only)
  uint16_t *totmaps_org = nullptr;
  bool reuse_analysis = false;         // the run is one chunk and slot 0 still holds its analysis (stage 2 behind stage 1)
  bool defer_head = false, defer_tail = false;  // a rank's edges wait for its neighbours' seams
  bool seams = false;                           // keep the raw sums behind the last hop (tail_raw)
  int64_t head_hi = 0, tail_lo = 0;             // out: the outputs [out_lo, head_hi) and [tail_lo, out_hi) were deferred
  float *pcm_f32 = nullptr;
  int16_t *pcm_i16 = nullptr;
  int64_t pcm_base = 0, out_lo = 0, out_hi = 0;
};

struct PvChunk {
  int64_t lo, hi;
};
std::vector<PvChunk> pv_chunks_of(int64_t F_lo, int64_t F_hi, int64_t C) {
  std::vector<PvChunk> v;
  for (int64_t lo = F_lo; lo < F_hi;) {
    int64_t hi = std::min(F_hi, lo + C);
    if (F_hi - hi < 32) hi = F_hi;  // a remainder shorter than one synthesis workgroup rides with the last chunk
    v.push_back({lo, hi});
    lo = hi;
  }
  return v;
}

#define PV_TRY(expr)                 \
  do {                               \
    if (e == hipSuccess) e = (expr); \
  } while (0)

int pv_run(mx_ctx *ctx, PvPipe &p, PvRun &run) {
  NTables t;
  int rc = get_tables(ctx, kPvN, t);
  if (rc) return rc;
  const hipStream_t sm = ctx->stream, ss = p.ss, sf = p.sf;
  const std::vector<PvChunk> chunks = pv_chunks_of(run.F_lo, run.F_hi, p.C);
  const int64_t K = (int64_t)chunks.size();
  if (K > 1 && (p.NS < kPvSlots || p.NOUT < kPvOutRing || p.NPLAN < kPvPlanRing))
    return fail(MX_ERR_INVALID, "phase vocoder: %lld chunks through an arena made for one", (long long)K);
  p.last_chunks = K;
  hipError_t e = hipSuccess;
  std::vector<PvArgs> args((size_t)K);
  for (int64_t k = 0; k < K; ++k) {
    const PvChunk c = chunks[(size_t)k];
    PvPipe::Slot &sl = p.slot[k % p.NS];
    PvPipe::Out &o = p.out[k % p.NOUT];
    const int64_t first = c.lo > 0 ? 1 : 0, Fl = c.hi - c.lo + first;
    PvArgs &g = args[(size_t)k];
    g = PvArgs{};
    g.audio = run.a->d_padded;
    g.n = run.a->n;
    g.ratio = run.r;
    g.frames = Fl;
    g.first = first;
    g.global_first = c.lo == 0;
    g.frame_base = c.lo;
    g.tw2 = t.tw2;
    g.tw3 = t.tw3;
    g.ubase = t.ubase;
    g.hann = p.hann;
    g.hann_scaled = p.hann_scaled;
    g.wsplit = p.wsplit;
    g.apos = p.plan[k % p.NPLAN].apos;
    g.hop = p.plan[k % p.NPLAN].hop;
    g.hratio = p.plan[k % p.NPLAN].hratio;
    g.xrows = sl.xrows;
    g.recs = sl.recs;
    g.rec_overflow = p.rec_overflow;
    g.pkmap = sl.pkmap;
    g.pkcount = sl.pkcount;
    g.fthr = sl.fthr;
    g.chunk_sums = sl.chunk_sums;
    g.chunk_org = sl.chunk_org;
    g.group_sums = sl.group_sums;
    g.group_org = sl.group_org;
    g.scan_chunk = (int)std::max<int64_t>(kPvMinScan, (Fl - first + kPvMaxScanChunks - 1) / kPvMaxScanChunks);
    g.halo = o.halo;
    g.s = o.s;
    g.s_len = (Fl - first) * kPvHs + kPvN;
    g.s_origin = c.lo * kPvHs;
    g.sample_rate = run.sample_rate;
    // frames per analysis workgroup: 16 in one launch over the whole signal (flat from 8 to 24 there); a chunk is four
    // rounds of workgroups at most, and what its launch loses is its ragged end — 8 (16: +0.8 ms per hour, 4: +0.2)
    g.frames_per_block = K > 1 ? 8 : 16;
    g.rec_wg_cap = (uint32_t)(g.frames_per_block * p.shape.rpf);
    g.rec_fpb_shift = g.frames_per_block == 8 ? 3 : 4;
    if (run.plan) {
      g.tf = o.tf;
      g.rf = o.rf;
      g.i0 = o.i0;
    }
    if (run.totals_only) {
      g.tot_sums = run.totmaps_sums + (size_t)k * kPvM;
      g.tot_org = run.totmaps_org + (size_t)k * kPvM;
    }
    g.carry_in = k > 0 ? p.carry[(k - 1) & 1] : run.carry_in;
    g.carry_out = p.carry[k & 1];
    // outputs of the chunk: those whose interpolation starts inside its hops (marker-driven: every frame owns its samples)
    if (!run.plan) {
      g.out_lo = k == 0 ? run.out_lo : pv_first_output_at(c.lo * kPvHs, run.r, run.a->n);
      g.out_hi = k == K - 1 ? run.out_hi : pv_first_output_at(c.hi * kPvHs, run.r, run.a->n);
    }
    g.pcm_f32 = run.pcm_f32;
    g.pcm_i16 = run.pcm_i16;
    g.pcm_base = run.pcm_base;
    if (k == 0 && run.defer_head) {
      // the rank's first N - Hs stretched samples need the previous rank's tail: their outputs wait for stage 3
      g.skip_head = 1;
      g.out_lo = std::min(g.out_hi, pv_first_output_at(c.lo * kPvHs + kPvSeam, run.r, run.a->n));
      run.head_hi = g.out_lo;
    }
    if (k == K - 1 && run.defer_tail) {
      // ... and the outputs that interpolate into the first sample behind the rank's hops need the next rank's head
      g.skip_tail = 1;
      g.out_hi = std::max(g.out_lo, std::min(g.out_hi, pv_first_output_at(c.hi * kPvHs - 1, run.r, run.a->n)));
      run.tail_lo = g.out_hi;
    }
    // the boundary between two chunks is finished in the left chunk's s: the right chunk copies it
    if (k > 0) g.prev_final = args[(size_t)k - 1].s + (args[(size_t)k - 1].frames - args[(size_t)k - 1].first) * kPvHs;
  }
  for (int64_t k = 0; k + 1 < K; ++k) args[(size_t)k].next_head = args[(size_t)k + 1].halo;  // the right chunk's head: its workgroup 0's halo

  // ---- the stages of one chunk ----
  // the constant-ratio plan rows of chunk k, on stream st (binary64 division and floor on the device: launch_pv_plan_const)
  auto plan_rows = [&](int64_t k, hipStream_t st) {
    const PvArgs &g = args[(size_t)k];
    const PvPipe::Plan &pl = p.plan[k % p.NPLAN];
    PV_TRY(launch_pv_plan_const(pl.apos, pl.hop, pl.hratio, g.frames, chunks[(size_t)k].lo - g.first, run.r, st));
  };
  // (the pipeline writes them three chunks ahead on the fix-up stream; only the rows of a run's first chunks, a marker plan's
  // and those of a rank's maps-only pass are made in front of their analysis)
  const bool plan_ahead = !run.plan && !run.totals_only;
  auto analysis = [&](int64_t k, hipStream_t sm) {  // (sm: the stream the transforms go on)
    const PvChunk c = chunks[(size_t)k];
    const PvArgs &g = args[(size_t)k];
    const PvPipe::Plan &pl = p.plan[k % p.NPLAN];
    if (run.plan) {
      const int64_t g0 = c.lo - g.first;  // global frame of local row 0
      PV_TRY(hipMemcpyAsync(pl.apos, run.plan->apos.data() + g0, (size_t)g.frames * 8, hipMemcpyHostToDevice, sm));
      PV_TRY(hipMemcpyAsync(pl.hop, run.plan_hop->data() + g0, (size_t)g.frames * 4, hipMemcpyHostToDevice, sm));
      PV_TRY(hipMemcpyAsync(pl.hratio, run.plan_hratio->data() + g0, (size_t)g.frames * 8, hipMemcpyHostToDevice, sm));
      // (the resampler's rows live with the chunk's stretched signal: the slot has a new tenant by the time it runs.  Its
      // last reader was the resampling of chunk k - 4 on the fix-up stream, queued at step k - 1 with an ev_fin behind it:
      // the event's latest record is that one when this is called)
      if (k >= p.NOUT) PV_TRY(hipStreamWaitEvent(sm, p.ev_fin, 0));
      PV_TRY(hipMemcpyAsync(const_cast<double *>(g.tf), run.plan->tf.data() + c.lo, (size_t)(c.hi - c.lo) * 8, hipMemcpyHostToDevice, sm));
      PV_TRY(hipMemcpyAsync(const_cast<double *>(g.rf), run.plan->rf.data() + c.lo, (size_t)(c.hi - c.lo) * 8, hipMemcpyHostToDevice, sm));
      PV_TRY(hipMemcpyAsync(const_cast<int64_t *>(g.i0), run.plan->i0.data() + c.lo, (size_t)(c.hi - c.lo + 1) * 8, hipMemcpyHostToDevice, sm));
    } else if (!plan_ahead) {
      plan_rows(k, sm);
    }
    // (plan_ahead: the rows were written on the fix-up stream at chunk k - 3's step, in front of its ev_fin; the side stream
    // waited for that event before chunk k - 2's recurrence, whose ev_lock S(k - 2) waited for — and S(k - 2) is in front of
    // this launch on this stream: no wait of its own between the two big kernels)
    PV_TRY(launch_pv_analysis(g, sm));
    // (the side stream waits for S(k - 1), behind this launch on the stream, where there is one: one marker fewer between the
    // two big kernels)
    if (k == 0 || run.totals_only) PV_TRY(hipEventRecord(p.ev_an[k % p.NS], sm));
  };
  auto synthesis = [&](int64_t k) {  // main stream
    const PvArgs &g = args[(size_t)k];
    PV_TRY(hipStreamWaitEvent(sm, p.ev_lock[k % p.NS], 0));
    PV_TRY(launch_pv_synthesis(g, sm));
    if (k == 0 && run.defer_head) {
      PV_TRY(hipMemcpyAsync(p.head_raw, g.halo, kPvSeam * 4, hipMemcpyDeviceToDevice, sm));
      // (the sample behind the seam left the synthesis finished: hop 15 of the chunk's first workgroup)
      PV_TRY(hipMemcpyAsync(p.edge_head + kPvSeam, g.s + kPvSeam, 4, hipMemcpyDeviceToDevice, sm));
    }
    if (k == K - 1 && run.seams)  // the raw sums behind the rank's last hop (the fix-up normalises them in place)
      PV_TRY(hipMemcpyAsync(p.tail_raw, g.s + (g.frames - g.first) * kPvHs, kPvSeam * 4, hipMemcpyDeviceToDevice, sm));
    if (k == K - 1 && run.defer_tail)  // (the last sample of the rank's hops is finished too: the last workgroup's last hop)
      PV_TRY(hipMemcpyAsync(p.edge_tail, g.s + (g.frames - g.first) * kPvHs - 1, 4, hipMemcpyDeviceToDevice, sm));
    PV_TRY(hipEventRecord(p.ev_syn[k % p.NS], sm));
  };

  if (!run.reuse_analysis) *p.rec_overflow = 0u;  // (host memory; nothing of an earlier call is in flight: every entry point joins its work)
  PV_TRY(hipEventRecord(p.ev_begin, sm));  // the input, and whatever used the arena before, are stream-ordered before this
  PV_TRY(hipStreamWaitEvent(ss, p.ev_begin, 0));
  PV_TRY(hipStreamWaitEvent(sf, p.ev_begin, 0));
  if (run.totals_only) {
    // stage 1 of a rank: transforms on the main stream, the maps (and the chunk's total map) beside the next chunk's
    for (int64_t k = 0; k < K && e == hipSuccess; ++k) {
      if (k >= p.NS) PV_TRY(hipStreamWaitEvent(sm, p.ev_lock[k % p.NS], 0));  // (the slot's maps are made)
      analysis(k, sm);
      PV_TRY(hipStreamWaitEvent(ss, p.ev_an[k % p.NS], 0));
      PV_TRY(launch_pv_maps(args[(size_t)k], ss));
      PV_TRY(hipEventRecord(p.ev_lock[k % p.NS], ss));
    }
  } else {
    // The main stream carries the two big kernels, one at a time:  A(0) A(1) S(0) A(2) S(1) A(3) S(2) ...  The side stream
    // carries everything small, beside an ANALYSIS: while A(k + 1) runs — S(k - 1) is through —, the records, maps and offsets
    // of chunk k, then the fix-up and resampling of chunk k - 2 (whose right neighbour's head S(k - 1) has just left).
    // Nothing runs beside a synthesis: its workgroups take a whole CU's LDS and registers, four to a CU, exactly one round
    // of them per chunk — a small kernel beside it displaces workgroups into a second round.
    if (plan_ahead && !run.reuse_analysis)
      for (int64_t k = 0; k < std::min<int64_t>(K, 3); ++k) plan_rows(k, sm);
    for (int64_t step = 0; step <= K && e == hipSuccess; ++step) {
      if (step < K && !run.reuse_analysis) analysis(step, sm);
      const int64_t j = step - 1;
      if (j < 0) continue;
      // side stream: chunk j's recurrence ...
      if (j >= 1) PV_TRY(hipStreamWaitEvent(ss, p.ev_syn[(j - 1) % p.NS], 0));  // (behind it on the main stream: A(j) is done too)
      else if (!run.reuse_analysis) PV_TRY(hipStreamWaitEvent(ss, p.ev_an[j % p.NS], 0));
      PvArgs gl = args[(size_t)j];
      gl.tot_sums = nullptr;
      gl.tot_org = nullptr;
      if (!run.reuse_analysis) PV_TRY(launch_pv_maps(gl, ss));
      PV_TRY(launch_pv_offsets(gl, ss));
      // the last hop of s is beyond every frame, and s[s_len] backs the interpolation's m + 1
      PV_TRY(hipMemsetAsync(gl.s + (gl.s_len - kPvHs), 0, (size_t)(kPvHs + 1) * 4, ss));
      PV_TRY(hipEventRecord(p.ev_lock[j % p.NS], ss));
      // ... and, on a stream of its own (it must not hold the recurrence up, nor sit beside the synthesis the recurrence
      // releases): the plan rows of chunk j + 3 (their ring slot's last reader was chunk j - 1's maps, in front of S(j - 1)) and
      // chunk j - 2's fix-up and resampling, S(j - 1) being through
      if (j >= 1) PV_TRY(hipStreamWaitEvent(sf, p.ev_syn[(j - 1) % p.NS], 0));
      if (plan_ahead && j + 3 < K) plan_rows(j + 3, sf);
      if (j >= 2) PV_TRY(launch_pv_finish(args[(size_t)j - 2], sf));
      PV_TRY(hipEventRecord(p.ev_fin, sf));
      // (S(j + 1) reuses a buffer the fix-up reads, A(j + 3) reads the plan rows: the next chunk's recurrence waits for both)
      PV_TRY(hipStreamWaitEvent(ss, p.ev_fin, 0));
      synthesis(j);
    }
    // the last two chunks' fix-up and resampling
    PV_TRY(hipStreamWaitEvent(sf, p.ev_syn[(K - 1) % p.NS], 0));
    if (K >= 2) PV_TRY(launch_pv_finish(args[(size_t)K - 2], sf));
    PV_TRY(launch_pv_finish(args[(size_t)K - 1], sf));
    PV_TRY(hipEventRecord(p.ev_fin, sf));
    PV_TRY(hipStreamWaitEvent(sm, p.ev_fin, 0));  // everything joins the context's stream
  }
  if (e != hipSuccess) {
    hipStreamSynchronize(ss);
    hipStreamSynchronize(sf);
    hipStreamSynchronize(sm);
    return fail(MX_ERR_DEVICE, "phase vocoder: %s", hipGetErrorString(e));
  }
  return MX_OK;
}

// Behind a run whose work is complete (the caller has synchronised): did an analysis overflow its compact record regions?
// Then the run's results are void: the context switches to full-size regions for good (until its scratch is released), the
// arena goes back, and the caller repeats its run on the one pv_pipe builds next.
bool pv_take_overflow(mx_ctx *ctx, PvPipe &p) {
  if (!p.rec_overflow || *p.rec_overflow == 0u) return false;
  *p.rec_overflow = 0u;
  ctx->pv_rec_full = true;
  pv_release(ctx);
  return true;
}

}  // namespace

void pv_release(mx_ctx *ctx) {
  PvPipe *p = ctx->pv;
  if (!p) return;
  hipSetDevice(ctx->device);
  if (p->ss) hipStreamSynchronize(p->ss);
  if (p->sf) hipStreamSynchronize(p->sf);
  hipStreamSynchronize(ctx->stream);
  pv_shard_drop(*p);
  for (hipEvent_t ev : {p->ev_begin, p->ev_fin})
    if (ev) hipEventDestroy(ev);
  for (int i = 0; i < kPvSlots; ++i)
    for (hipEvent_t ev : {p->ev_an[i], p->ev_lock[i], p->ev_syn[i]})
      if (ev) hipEventDestroy(ev);
  if (p->ss) hipStreamDestroy(p->ss);
  if (p->sf) hipStreamDestroy(p->sf);
  hipFree(p->base);
  if (p->rec_overflow) hipHostFree(p->rec_overflow);
  delete p;
  ctx->pv = nullptr;
}

}  // namespace mx

extern "C" {

int mx_pv_set_chunk_frames(mx_ctx *ctx, int64_t frames) {
  return mx_guard([&]() -> int {
    if (!ctx || frames < 0) return fail(MX_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> plk(ctx->pv_mu);
    ctx->pv_chunk_frames = frames;  // (the arena is rebuilt by the next call that needs another size)
    return MX_OK;
  });
}

int64_t mx_pv_arena_bytes(mx_ctx *ctx) {
  return mx_guard([&]() -> int64_t {
    if (!ctx) return fail(MX_ERR_INVALID, "null context");
    std::lock_guard<std::mutex> plk(ctx->pv_mu);
    return ctx->pv ? (int64_t)ctx->pv->bytes : 0;
  });
}

int mx_pv_set_arena_budget(mx_ctx *ctx, int64_t bytes) {
  return mx_guard([&]() -> int {
    if (!ctx || bytes < 0) return fail(MX_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> plk(ctx->pv_mu);
    ctx->pv_budget_bytes = bytes;
    // (an arena above the new budget goes back now; one inside it is kept for as long as it serves)
    if (bytes > 0 && ctx->pv && ctx->pv->bytes > (size_t)bytes) pv_release(ctx);
    return MX_OK;
  });
}

int64_t mx_pv_arena_budget(mx_ctx *ctx) {
  return mx_guard([&]() -> int64_t {
    if (!ctx) return fail(MX_ERR_INVALID, "null context");
    std::lock_guard<std::mutex> plk(ctx->pv_mu);
    HIP_TRY(hipSetDevice(ctx->device));
    size_t b = 0;
    const int rc = pv_budget(ctx, &b);
    return rc ? (int64_t)rc : (int64_t)b;
  });
}

int64_t mx_pv_last_chunks(mx_ctx *ctx) {
  return mx_guard([&]() -> int64_t {
    if (!ctx) return fail(MX_ERR_INVALID, "null context");
    std::lock_guard<std::mutex> plk(ctx->pv_mu);
    return ctx->pv ? ctx->pv->last_chunks : 0;
  });
}

}  // extern "C"

namespace {
int pv_pitch_shift_dev_impl(mx_ctx *ctx, const mx_audio *a, double semitones, float *d_pcm_f32, int16_t *d_pcm_i16) {
  if (!ctx || !a) return fail(MX_ERR_INVALID, "null context or audio handle");
  if (!(semitones >= -48.0 && semitones <= 48.0)) return fail(MX_ERR_INVALID, "semitones out of range [-48, 48]");
  if (a->n == 0 || (!d_pcm_f32 && !d_pcm_i16)) return MX_OK;
  std::lock_guard<std::mutex> plk(ctx->pv_mu);
  PvRun run;
  run.a = a;
  run.r = std::pow(2.0, semitones / 12.0);
  run.F_lo = 0;
  run.F_hi = pv_frame_count(a->n, run.r);
  run.out_lo = 0;
  run.out_hi = a->n;
  run.pcm_f32 = d_pcm_f32;
  run.pcm_i16 = d_pcm_i16;
  for (int attempt = 0;; ++attempt) {
    PvPipe *p = nullptr;
    int rc = pv_pipe(ctx, run.F_hi, &p);
    if (rc) return rc;
    pv_shard_drop(*p);
    rc = pv_run(ctx, *p, run);
    const hipError_t es = hipStreamSynchronize(ctx->stream);
    if (rc) return rc;
    if (es != hipSuccess) return fail(MX_ERR_DEVICE, "phase vocoder: %s", hipGetErrorString(es));
    if (!pv_take_overflow(ctx, *p)) return MX_OK;
    // (more peaks than the compact record regions hold — an impulse train, noise —: once more, with full-size regions)
    if (attempt) return fail(MX_ERR_DEVICE, "phase vocoder: record regions overflowed at full size");
  }
}
}  // namespace

extern "C" {

int mx_pv_pitch_shift_dev(mx_ctx *ctx, const mx_audio *a, double semitones, float *d_pcm_f32, int16_t *d_pcm_i16) {
  return mx_guard([&]() -> int {
    return pv_pitch_shift_dev_impl(ctx, a, semitones, d_pcm_f32, d_pcm_i16);
  });
}

// Marker-driven variant: the vocoder steered by the editor's markers as App::exportWav is (warped time, pitch bend).
int64_t mx_pv_render_length(int64_t n, int sampleRate, const mx_marker *markers, int nmarkers) {
  return mx_guard([&]() -> int64_t {
    if (n < 0 || nmarkers < 0 || (nmarkers > 0 && !markers)) return fail(MX_ERR_INVALID, "bad argument");
    PvPlan plan;
    std::string err;
    const int rc = build_pv_plan(markers, nmarkers, sampleRate, n, plan, err);
    if (rc) return fail(rc, "%s", err.c_str());
    return plan.n_out;
  });
}

int mx_pv_plan(int64_t n, int sampleRate, const mx_marker *markers, int nmarkers, int64_t **apos, double **tf,
               double **rf, int64_t **i0, int64_t *frames, int64_t *nsamples) {
  return mx_guard([&]() -> int {
    if (n < 0 || nmarkers < 0 || (nmarkers > 0 && !markers) || !apos || !tf || !rf || !i0 || !frames || !nsamples)
      return fail(MX_ERR_INVALID, "bad argument");
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
  });
}

int mx_pv_render_dev(mx_ctx *ctx, const mx_audio *a, int sampleRate, const mx_marker *markers, int nmarkers,
                     float *d_pcm_f32, int16_t *d_pcm_i16) {
  return mx_guard([&]() -> int {
    if (!ctx || !a || nmarkers < 0 || (nmarkers > 0 && !markers)) return fail(MX_ERR_INVALID, "bad argument");
    if (a->n == 0 || (!d_pcm_f32 && !d_pcm_i16)) return MX_OK;
    PvPlan plan;
    std::string err;
    int rc = build_pv_plan(markers, nmarkers, sampleRate, a->n, plan, err);
    if (rc) return fail(rc, "%s", err.c_str());
    if (plan.n_out == 0) return MX_OK;
    for (int64_t c : plan.apos)
      if (c < -(int64_t)MX_AUDIO_PAD / 2 || c > a->n + (int64_t)MX_AUDIO_PAD / 2)
        return fail(MX_ERR_INVALID, "a marker maps warped time outside the audio");
    // hops and stretch factors of the plan (the constant-ratio plan computes them on the device)
    const size_t F = plan.apos.size();
    std::vector<uint32_t> hop(F, 0u);
    std::vector<double> hratio(F, 0.0);
    for (size_t j = 1; j < F; ++j) {
      const int64_t h = plan.apos[j] - plan.apos[j - 1];
      if (h >= 1 && h <= 0x7fffffffLL) {
        hop[j] = (uint32_t)h;
        hratio[j] = (double)kPvHs / (double)h;
      }
    }
    std::lock_guard<std::mutex> plk(ctx->pv_mu);
    PvRun run;
    run.a = a;
    run.plan = &plan;
    run.plan_hop = &hop;
    run.plan_hratio = &hratio;
    run.sample_rate = sampleRate;
    run.F_lo = 0;
    run.F_hi = (int64_t)F;
    run.pcm_f32 = d_pcm_f32;
    run.pcm_i16 = d_pcm_i16;
    for (int attempt = 0;; ++attempt) {
      PvPipe *p = nullptr;
      rc = pv_pipe(ctx, (int64_t)F, &p);
      if (rc) return rc;
      pv_shard_drop(*p);
      rc = pv_run(ctx, *p, run);
      const hipError_t es = hipStreamSynchronize(ctx->stream);  // (the plan's host arrays die with this frame)
      if (rc) return rc;
      if (es != hipSuccess) return fail(MX_ERR_DEVICE, "phase vocoder: %s", hipGetErrorString(es));
      if (!pv_take_overflow(ctx, *p)) return MX_OK;
      if (attempt) return fail(MX_ERR_DEVICE, "phase vocoder: record regions overflowed at full size");
    }
  });
}

int mx_pv_render(mx_ctx *ctx, const mx_audio *a, int sampleRate, const mx_marker *markers, int nmarkers,
                 float *pcm_f32_out, int16_t *pcm_i16_out) {
  return mx_guard([&]() -> int {
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
  });
}


// ---- one rank of a multi-GPU phase-vocoder run (the overlap-add seams) ------------------------
// Every rank holds the whole input and takes a contiguous range of the frame axis (boundaries on multiples of 32
// frames = the synthesis workgroups, so the float sums group exactly as in a single-GPU run).  Two small exchanges
// happen outside this library (RCCL / gloo all-gathers in the caller): after stage 1 the per-rank phase totals
// (2048 x {restart, phase}), after stage 2 the seams (2 x 3840 raw partial sums).  A rank walks its range through the
// same bounded arena as a single GPU walks the whole signal; what that costs it: stage 1 cannot keep the spectra of
// more than one chunk, so a rank whose range is longer than a chunk analyses its frames twice (stage 1 for the maps
// alone, stage 2 again with the carry) — a range of one chunk is analysed once.
int mx_pv_shard_frames(int64_t n, double semitones, int rank, int world, int64_t *frame_lo, int64_t *frame_hi,
                       int64_t *out_lo, int64_t *out_hi) {
  return mx_guard([&]() -> int {
    if (n <= 0 || world < 1 || rank < 0 || rank >= world || !(semitones >= -48.0 && semitones <= 48.0))
      return fail(MX_ERR_INVALID, "bad argument");
    const double r = std::pow(2.0, semitones / 12.0);
    const int64_t F = pv_frame_count(n, r);
    int64_t per = (F + world - 1) / world;
    per = (per + 31) / 32 * 32;
    // (every rank gets at least one synthesis workgroup of its own: the seams either side of a rank must not overlap)
    if (world > 1 && F - per * (world - 1) < 32)
      return fail(MX_ERR_INVALID, "signal too short for %d ranks (%lld frames)", world, (long long)F);
    const int64_t lo = (int64_t)rank * per, hi = rank == world - 1 ? F : lo + per;
    if (frame_lo) *frame_lo = lo;
    if (frame_hi) *frame_hi = hi;
    if (out_lo) *out_lo = rank == 0 ? 0 : pv_first_output_at(lo * kPvHs, r, n);
    if (out_hi) *out_hi = rank == world - 1 ? n : pv_first_output_at(hi * kPvHs, r, n);
    return MX_OK;
  });
}

// The three stages, on device memory throughout; the host-pointer entry points wrap them with copies, the _dev entry points
// hand the caller's buffers straight through.
//   stage 1 -> d_map_out: 2048 uint32 sums, then 2048 uint16 source bins (12 KiB: one rank's entry of the first all-gather)
static int pv_shard_analyze_core(mx_ctx *ctx, const mx_audio *a, double semitones, int rank, int world, void *map_out, bool map_on_device) {
  if (!ctx || !a || !map_out) return fail(MX_ERR_INVALID, "bad argument");
  int64_t lo, hi, olo, ohi;
  int rc = mx_pv_shard_frames(a->n, semitones, rank, world, &lo, &hi, &olo, &ohi);
  if (rc) return rc;
  std::lock_guard<std::mutex> plk(ctx->pv_mu);
  PvPipe *p = nullptr;
  PvRun run;
  int64_t K = 1;
  for (int attempt = 0;; ++attempt) {
    rc = pv_pipe(ctx, hi - lo + 1, &p);  // (+ the row before the range)
    if (rc) return rc;
    pv_shard_drop(*p);
    K = (int64_t)pv_chunks_of(lo, hi, p->C).size();
    // every chunk's total map (12 KiB each), folded into the rank's behind the last analysis; a resident range has one: it is
    // written where it stays (no allocation on the way of a rank that fits its budget)
    uint32_t *d_sums = p->slot[0].tot_sums, *own_sums = nullptr;
    uint16_t *d_org = p->slot[0].tot_org, *own_org = nullptr;
    hipError_t e = hipSuccess;
    if (K > 1) {
      e = hipMalloc(&own_sums, (size_t)K * kPvM * 4);
      if (e == hipSuccess) e = hipMalloc(&own_org, (size_t)K * kPvM * 2);
      if (e != hipSuccess) {
        hipFree(own_sums);
        return fail(MX_ERR_NOMEM, "phase-vocoder chunk maps: %s", hipGetErrorString(e));
      }
      d_sums = own_sums;
      d_org = own_org;
    }
    run = PvRun{};
    run.a = a;
    run.r = std::pow(2.0, semitones / 12.0);
    run.F_lo = lo;
    run.F_hi = hi;
    run.totals_only = true;
    run.totmaps_sums = d_sums;
    run.totmaps_org = d_org;
    try {
      rc = pv_run(ctx, *p, run);
    } catch (...) {
      hipStreamSynchronize(p->ss);
      hipStreamSynchronize(ctx->stream);
      hipFree(own_sums);
      hipFree(own_org);
      throw;
    }
    if (rc == MX_OK) {
      const uint32_t *rs = d_sums;
      const uint16_t *ro = d_org;
      if (K > 1) {
        e = launch_pv_compose_maps(d_sums, d_org, K, p->slot[0].tot_sums, p->slot[0].tot_org, p->ss);
        rs = p->slot[0].tot_sums;
        ro = p->slot[0].tot_org;
      }
      const hipMemcpyKind kind = map_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
      if (e == hipSuccess) e = hipMemcpyAsync(map_out, rs, kPvM * 4, kind, p->ss);
      if (e == hipSuccess) e = hipMemcpyAsync(static_cast<char *>(map_out) + kPvM * 4, ro, kPvM * 2, kind, p->ss);
    }
    const hipError_t es = hipStreamSynchronize(p->ss);
    hipFree(own_sums);
    hipFree(own_org);
    if (rc) return rc;
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) return fail(MX_ERR_DEVICE, "phase vocoder (analysis): %s", hipGetErrorString(e));
    if (!pv_take_overflow(ctx, *p)) break;
    // (more peaks than the compact record regions hold: once more, with full-size regions)
    if (attempt) return fail(MX_ERR_DEVICE, "phase vocoder: record regions overflowed at full size");
  }
  PvPipe::Shard &j = p->job;
  j.active = true;
  j.rank = rank;
  j.world = world;
  j.first = rank == 0;
  j.last = rank == world - 1;
  j.single = K == 1;  // the rows are resident: stage 2 goes straight to the offsets and the synthesis
  j.a = a;
  j.semitones = semitones;
  j.r = run.r;
  j.F_lo = lo;
  j.F_hi = hi;
  j.out_lo = olo;
  j.out_hi = ohi;
  return MX_OK;
}

//   stage 2: the carry into the rank — a device row (carry_dev), a host row, or folded here from the gathered maps of the
//   ranks below (maps_all: [world] x 12 KiB as stage 1 wrote them) -> the rank's outputs but its edges, and its two seams
//   (head then tail, 2 x 3840 floats: one rank's entry of the second all-gather)
static int pv_shard_synthesize_core(mx_ctx *ctx, const uint32_t *carry_host, const void *d_maps_all, float *d_pcm_f32, int16_t *d_pcm_i16,
                                    bool own_pcm, float *head_out, float *tail_out, bool seams_on_device) {
  if (!ctx || !head_out || !tail_out) return fail(MX_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> plk(ctx->pv_mu);
  PvPipe *p = ctx->pv;
  if (!p || !p->job.active || p->job.synthesized) return fail(MX_ERR_INVALID, "mx_pv_shard_analyze has not run on this context");
  PvPipe::Shard &j = p->job;
  if (!j.first && !carry_host && !d_maps_all) return fail(MX_ERR_INVALID, "carry_in is required on every rank but the first");
  if (!own_pcm && !d_pcm_f32 && !d_pcm_i16) return fail(MX_ERR_INVALID, "no output buffer");
  HIP_TRY(hipSetDevice(ctx->device));
  const hipStream_t sm = ctx->stream;
  const int64_t cnt = j.out_hi - j.out_lo;
  hipError_t e = hipSuccess;
  // the carry first: nothing is allocated yet if it cannot be had
  if (!j.first) {
    if (d_maps_all) {
      // the maps of ranks 0 .. rank - 1 composed in order and applied to a zero row: the composed map's sums (a bin whose
      // source is a bin of the zero row ends at its sum; so does one that restarted) — read where the all-gather left them,
      // one 8 KiB sums row and one 4 KiB source-bin row per rank
      constexpr int kEntry = kPvM * 6;
      const uint32_t *ms = static_cast<const uint32_t *>(d_maps_all);
      const uint16_t *mo = reinterpret_cast<const uint16_t *>(static_cast<const char *>(d_maps_all) + (size_t)kPvM * 4);
      PV_TRY(launch_pv_compose_maps(ms, mo, j.rank, p->carry_in, p->slot[0].tot_org, sm, kEntry / 4, kEntry / 2));
    } else {
      PV_TRY(hipMemcpyAsync(p->carry_in, carry_host, kPvM * 4, hipMemcpyHostToDevice, sm));
    }
    if (e != hipSuccess) {
      hipStreamSynchronize(sm);
      pv_shard_drop(*p);
      return fail(MX_ERR_DEVICE, "phase vocoder (carry): %s", hipGetErrorString(e));
    }
  }
  // the rank's outputs wait on the device for stage 3: in the caller's buffers, or (host-pointer entry points: the caller
  // chooses the formats in stage 3) in buffers of the library's own, both formats
  j.own_pcm = own_pcm;
  if (own_pcm) {
    if (cnt > 0) {
      e = hipMalloc(&j.d_f, (size_t)cnt * 4);
      if (e == hipSuccess) e = hipMalloc(&j.d_i, (size_t)cnt * 2);
      if (e != hipSuccess) {
        hipStreamSynchronize(sm);
        pv_shard_drop(*p);
        return fail(MX_ERR_NOMEM, "device PCM buffers: %s", hipGetErrorString(e));
      }
    }
  } else {
    j.d_f = d_pcm_f32;
    j.d_i = d_pcm_i16;
  }
  PvRun run;
  run.a = j.a;
  run.r = j.r;
  run.F_lo = j.F_lo;
  run.F_hi = j.F_hi;
  run.carry_in = j.first ? nullptr : p->carry_in;
  run.reuse_analysis = j.single;
  run.defer_head = !j.first;
  run.defer_tail = !j.last;
  run.seams = true;
  run.pcm_f32 = j.d_f;
  run.pcm_i16 = j.d_i;
  run.pcm_base = j.out_lo;
  run.out_lo = run.head_hi = j.out_lo;
  run.out_hi = run.tail_lo = j.out_hi;
  int rc;
  try {
    rc = pv_run(ctx, *p, run);
  } catch (...) {  // (host containers inside: a failed allocation must not leave the job holding the PCM buffers it has just taken)
    hipStreamSynchronize(sm);
    pv_shard_drop(*p);
    throw;
  }
  // the seams, raw: this rank's sums into the N - Hs samples before its first complete hop (all zero on the first rank,
  // whose first hops are complete) and after its last hop
  if (rc == MX_OK) {
    const hipMemcpyKind kind = seams_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (!j.first) e = hipMemcpyAsync(head_out, p->head_raw, kPvSeam * 4, kind, sm);
    else if (seams_on_device) e = hipMemsetAsync(head_out, 0, kPvSeam * 4, sm);
    else memset(head_out, 0, kPvSeam * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(tail_out, p->tail_raw, kPvSeam * 4, kind, sm);
  }
  const hipError_t es = hipStreamSynchronize(sm);
  if (rc == MX_OK && e == hipSuccess) e = es;
  if (rc || e != hipSuccess) {
    pv_shard_drop(*p);
    return rc ? rc : fail(MX_ERR_DEVICE, "phase vocoder (synthesis): %s", hipGetErrorString(e));
  }
  if (*p->rec_overflow != 0u) {  // (a chunked range analyses again here, cut as in stage 1, which passed: cannot happen)
    pv_shard_drop(*p);
    return fail(MX_ERR_DEVICE, "phase vocoder (synthesis): record regions overflowed behind a stage 1 that fitted");
  }
  j.synthesized = true;
  j.head_hi = run.head_hi;
  j.tail_lo = run.tail_lo;
  return MX_OK;
}

//   stage 3: the neighbours' seams (host rows, or device rows: entries of the gathered seams) -> the rank's edge outputs
static int pv_shard_finish_core(mx_ctx *ctx, const float *prev_tail, const float *next_head, bool seams_on_device, const void *d_seams_all,
                                float *pcm_f32_out, int16_t *pcm_i16_out) {
  if (!ctx) return fail(MX_ERR_INVALID, "null context");
  std::lock_guard<std::mutex> plk(ctx->pv_mu);
  PvPipe *p = ctx->pv;
  if (!p || !p->job.active || !p->job.synthesized) return fail(MX_ERR_INVALID, "mx_pv_shard_synthesize has not run on this context");
  PvPipe::Shard &j = p->job;
  if (d_seams_all) {  // [world] x {head, tail} x 3840 floats, as stage 2 wrote them
    const float *all = static_cast<const float *>(d_seams_all);
    prev_tail = j.first ? nullptr : all + ((size_t)(j.rank - 1) * 2 + 1) * kPvSeam;
    next_head = j.last ? nullptr : all + (size_t)(j.rank + 1) * 2 * kPvSeam;
    seams_on_device = true;
  }
  if ((!j.first && !prev_tail) || (!j.last && !next_head)) return fail(MX_ERR_INVALID, "a neighbour's seam is missing");
  HIP_TRY(hipSetDevice(ctx->device));
  const hipStream_t sm = ctx->stream;
  const int64_t cnt = j.out_hi - j.out_lo;
  const hipMemcpyKind kind = seams_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  hipError_t e = hipSuccess;
  PvArgs g{};
  g.ratio = j.r;
  g.pcm_f32 = j.d_f;
  g.pcm_i16 = j.d_i;
  g.pcm_base = j.out_lo;
  if (!j.first) {
    // the rank's first N - Hs stretched samples: its head + the previous rank's tail, then their outputs
    PV_TRY(hipMemcpyAsync(p->prev_tail, prev_tail, kPvSeam * 4, kind, sm));
    PV_TRY(launch_pv_edge_sum(p->edge_head, p->head_raw, p->prev_tail, kPvSeam, sm));
    g.s = p->edge_head;
    g.s_origin = j.F_lo * kPvHs;
    g.out_lo = j.out_lo;
    g.out_hi = j.head_hi;
    PV_TRY(launch_pv_resample(g, sm));
  }
  if (!j.last) {
    // the outputs that interpolate between the rank's last stretched sample and the first one behind it
    PV_TRY(hipMemcpyAsync(p->next_head, next_head, kPvSeam * 4, kind, sm));
    PV_TRY(launch_pv_edge_sum(p->edge_tail + 1, p->tail_raw, p->next_head, 4, sm));
    g.s = p->edge_tail;
    g.s_origin = j.F_hi * kPvHs - 1;
    g.out_lo = j.tail_lo;
    g.out_hi = j.out_hi;
    PV_TRY(launch_pv_resample(g, sm));
  }
  if (j.own_pcm) {
    if (pcm_f32_out && cnt) PV_TRY(hipMemcpyAsync(pcm_f32_out, j.d_f, (size_t)cnt * 4, hipMemcpyDeviceToHost, sm));
    if (pcm_i16_out && cnt) PV_TRY(hipMemcpyAsync(pcm_i16_out, j.d_i, (size_t)cnt * 2, hipMemcpyDeviceToHost, sm));
  }
  const hipError_t es = hipStreamSynchronize(sm);
  if (e == hipSuccess) e = es;
  pv_shard_drop(*p);
  if (e != hipSuccess) return fail(MX_ERR_DEVICE, "phase vocoder (finish): %s", hipGetErrorString(e));
  return MX_OK;
}

int mx_pv_shard_analyze(mx_ctx *ctx, const mx_audio *a, double semitones, int rank, int world, uint32_t *tot_sums_out,
                        uint16_t *tot_org_out) {
  return mx_guard([&] {
    if (!tot_sums_out || !tot_org_out) return fail(MX_ERR_INVALID, "bad argument");
    // (the two host arrays need not be adjacent: through one 12 KiB landing buffer)
    std::vector<uint32_t> map((size_t)kPvM * 6 / 4);
    const int rc = pv_shard_analyze_core(ctx, a, semitones, rank, world, map.data(), false);
    if (rc) return rc;
    memcpy(tot_sums_out, map.data(), (size_t)kPvM * 4);
    memcpy(tot_org_out, map.data() + kPvM, (size_t)kPvM * 2);
    return MX_OK;
  });
}
int mx_pv_shard_synthesize(mx_ctx *ctx, const uint32_t *carry_in, float *head_out, float *tail_out) {
  return mx_guard([&] { return pv_shard_synthesize_core(ctx, carry_in, nullptr, nullptr, nullptr, true, head_out, tail_out, false); });
}
int mx_pv_shard_finish(mx_ctx *ctx, const float *prev_tail, const float *next_head, float *pcm_f32_out,
                       int16_t *pcm_i16_out) {
  return mx_guard([&] { return pv_shard_finish_core(ctx, prev_tail, next_head, false, nullptr, pcm_f32_out, pcm_i16_out); });
}

// The same three stages with everything a rank exchanges left on the device, laid out as the two all-gathers move it: stage 1
// writes the rank's 12 KiB entry of the maps, stage 2 reads the gathered maps ([world] entries; it folds those of the ranks below
// into its carry on the device) and writes the rank's 30 KiB entry of the seams, stage 3 reads the gathered seams.  The rank's
// PCM goes straight into the caller's device buffers (out_hi - out_lo samples, either may be NULL) from stage 2 on.
int mx_pv_shard_analyze_dev(mx_ctx *ctx, const mx_audio *a, double semitones, int rank, int world, void *d_map_out) {
  return mx_guard([&] { return pv_shard_analyze_core(ctx, a, semitones, rank, world, d_map_out, true); });
}
int mx_pv_shard_synthesize_dev(mx_ctx *ctx, const void *d_maps_all, float *d_pcm_f32, int16_t *d_pcm_i16, void *d_seams_out) {
  return mx_guard([&] {
    if (!d_seams_out || !d_maps_all) return fail(MX_ERR_INVALID, "bad argument");
    float *seams = static_cast<float *>(d_seams_out);
    return pv_shard_synthesize_core(ctx, nullptr, d_maps_all, d_pcm_f32, d_pcm_i16, false, seams, seams + kPvSeam, true);
  });
}
int mx_pv_shard_finish_dev(mx_ctx *ctx, const void *d_seams_all) {
  return mx_guard([&] {
    if (!d_seams_all) return fail(MX_ERR_INVALID, "bad argument");
    return pv_shard_finish_core(ctx, nullptr, nullptr, true, d_seams_all, nullptr, nullptr);
  });
}

int mx_pv_pitch_shift(mx_ctx *ctx, const mx_audio *a, double semitones, float *pcm_f32_out, int16_t *pcm_i16_out) {
  return mx_guard([&]() -> int {
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
  });
}

}  // extern "C"
