/*
 * melonix_amd.h — C-ABI boundary of the MI355X-native melonix hot path.
 *
 * This is the drop-in boundary: plain C types, opaque handles, caller-allocated
 * outputs, int status codes (0 = ok, negative = error; text via mx_last_error()).
 * No C++ types, no exceptions and no torch types cross it.  The C++ facade in
 * melonix_amd/cpp/ (Spec, SpecCache, saveWav — the reference's own class
 * surface) and every parity test / bench call through these entry points.
 *
 * Each entry point names the reference interface it replaces; file:line are
 * relative to the reference tree (mika314/melonix @ 2025-05-23).
 *
 * Device layout of an mx_audio (HBM): [MX_AUDIO_PAD zeros][n samples f32][MX_AUDIO_PAD zeros]
 * so that a frame [end-N, end) that straddles either end of the file reads
 * zeros exactly like spec.cpp:50-54 without per-sample bounds checks.
 */
#ifndef MELONIX_AMD_H
#define MELONIX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MX_OK 0
#define MX_ERR_INVALID (-1)  /* bad argument */
#define MX_ERR_DEVICE (-2)   /* HIP runtime error / no MI355X visible */
#define MX_ERR_NOMEM (-3)
#define MX_ERR_IO (-4)

#define MX_AUDIO_PAD 32768 /* samples of zero padding either side (>= largest N) */

typedef struct mx_ctx mx_ctx;     /* one per GPU / per process rank */
typedef struct mx_audio mx_audio; /* device-resident mono f32 audio */

/* Per-frame pitch record (build-defined op, SURVEY §8 a-6: the reference has no
 * detector; note law from app.cpp:499-516). */
typedef struct mx_pitch {
  int32_t bin; /* argmax_k mag[k], k in [kmin,kmax], ties -> lowest k */
  float mag;   /* mag[bin] */
} mx_pitch;

/* marker.hpp:4-9 */
typedef struct mx_marker {
  int32_t sample;
  double note;
  double dTime;
  double pitchBend;
} mx_marker;

/* One App::process() call of the export loop (app.cpp:294-345), precomputed. */
typedef struct mx_step {
  double cursor;       /* warped time at entry (app.cpp:1201-1206) */
  int32_t grain_start; /* key of the chosen grain = first source sample (app.cpp:298-301) */
  int32_t grain_len;   /* grain.size() */
  float rate;          /* powf(2, pitchBend/12) (app.cpp:297) */
  float next_first;    /* nextGrainFirstSample (app.cpp:312-329) */
  int32_t sz;          /* samples this step emits (app.cpp:332-343) */
  int32_t _pad;
  int64_t out_offset;  /* exclusive prefix sum of sz = position in the PCM stream */
} mx_step;

/* ---- context ------------------------------------------------------------ */

/* device: HIP ordinal.  Fails with MX_ERR_DEVICE when no gfx950 device is usable. */
int mx_ctx_create(int device, mx_ctx **out);
void mx_ctx_destroy(mx_ctx *ctx);
/* A new context launches on its own non-blocking stream.  mx_ctx_set_stream makes it launch on a
 * caller-owned hipStream_t instead (e.g. torch's current stream; NULL = the HIP null stream);
 * mx_ctx_use_own_stream switches back. */
int mx_ctx_set_stream(mx_ctx *ctx, void *hip_stream);
int mx_ctx_use_own_stream(mx_ctx *ctx);
int mx_ctx_synchronize(mx_ctx *ctx);
/* The context keeps its work buffers between calls (device staging of the host-pointer entry points, the phase
 * vocoder's budgeted arena and its automatic budget, the host landing zone of mx_grains_dev); this releases them.
 * mx_ctx_destroy does so too. */
int mx_ctx_release_scratch(mx_ctx *ctx);
/* Run length: consecutive frames one workgroup of a bulk (uniform-hop) launch walks.  The sliding-window kernels
 * restart their window from the exact weights at the head of every run, so a row's last bits depend on where the runs
 * start.  The default is a function of the launch's frame count (a power of two, at most 32; short launches use short
 * runs so that every CU gets work): mx_stft_run_length returns it (> 0; < 0 = error code).  A job that computes one
 * signal in several launches — the frame shards of a multi-GPU run — gets the rows of the single launch bit for bit by
 * starting every launch on a multiple of 32 frames and pinning its run length to the whole signal's:
 * mx_ctx_set_frames_per_block(ctx, mx_stft_run_length(N, hop, total_frames)) (0 = back to the default).  The pin is
 * per CONTEXT, not per (N, hop): it applies to every bulk launch of that context at any size until it is set back to 0
 * (a job that mixes sizes re-pins between them); ranges-mode launches (mx_stft_ranges*) ignore it. */
int mx_stft_run_length(int N, int hop, int64_t count);
int mx_ctx_set_frames_per_block(mx_ctx *ctx, int frames);
/* Page-locked host memory for the buffers the host-pointer entry points fill (magnitude / texel rows, PCM): a
 * device->host copy into it is a direct DMA at PCIe rate, a copy into fresh pageable memory is several times slower
 * (staging + page faults).  The reference has no counterpart (its rows never leave the CPU, spec.cpp:61-65); the
 * facade's Spec worker lands every batch in such a buffer and recycles it.  Free with mx_pinned_free. */
int mx_pinned_alloc(mx_ctx *ctx, size_t bytes, void **out);
void mx_pinned_free(mx_ctx *ctx, void *p);
/* ERRORS.  Every entry point reports failure through its return value — a negative status (int / int64_t entry points), NaN (the
 * double / float time maps), MX_ERR_* as the int of mx_time2sample — and a thread-local description of the last error returned on
 * this thread; nothing is ever thrown across this boundary (every entry point's body runs inside a catch-all: std::bad_alloc ->
 * MX_ERR_NOMEM, any other exception -> MX_ERR_INVALID), and a failed call leaves its output pointers and handles untouched.
 * The reference signals no errors at all (spec.cpp / app.cpp:628-666 degrade to empty results); the facade maps a failed call to
 * an empty vector / a black column the same way. */
const char *mx_last_error(void);
/* "melonix_amd <version> gfx950" */
const char *mx_version(void);

/* ---- audio ---------------------------------------------------------------
 * Replaces Spec::Spec(std::span<float> wav) borrowing App::wavData
 * (spec.cpp:10-16, app.cpp:251): the samples are copied to HBM once. */
int mx_audio_upload(mx_ctx *ctx, const float *host_wav, int64_t n, mx_audio **out);
/* Zero-copy: wrap a device buffer already laid out [PAD zeros][n][PAD zeros]
 * (d_padded points at the first pad sample, 16-byte aligned).  The caller keeps ownership. */
int mx_audio_wrap_device(mx_ctx *ctx, const float *d_padded, int64_t n, mx_audio **out);
int64_t mx_audio_length(const mx_audio *a);
int mx_audio_free(mx_ctx *ctx, mx_audio *a);

/* ---- STFT magnitude spectrogram + pitch pick -------------------------------
 * Replaces Spec::internalGetSpec (spec.cpp:44-66) with SpectrSize (spec.cpp:8)
 * generalised to N in {4096, 16384, 32768}: frame = samples [end-N, end),
 * one-sided exponential window expf(-2.5e-4f*(start-i)) left of `start`,
 * zeros outside the file, |DFT|/N for bins 0..N/2-1.
 *
 * kmin/kmax: pitch-pick band (inclusive, clamped to [0, N/2-1]); pass -1,-1
 * for the default band (notes 24..84 = 55..1760 Hz) at 48 kHz; for any other
 * sample rate pass the bins mx_pitch_band() returns.
 */

/* Default pitch band for (N, sampleRate): kmin=ceil(55*N/sr), kmax=floor(1760*N/sr). */
void mx_pitch_band(int N, int sampleRate, int *kmin, int *kmax);

/* The editor's note <-> frequency law (app.cpp:498-499: f(note) = 55 * 2^((note-24)/12) Hz; bin k of an
 * N-point frame sits at k*sampleRate/N Hz).  mx_bin_note turns a pitch record's bin into the value a
 * Marker::note (marker.hpp:6) holds: 24 + 12*log2(k*sampleRate/N/55); -HUGE_VAL for k <= 0.
 * mx_note_bin is its inverse (fractional bin).  Host, double. */
double mx_bin_note(int bin, int N, int sampleRate);
double mx_note_bin(double note, int N, int sampleRate);

/* Arbitrary (start,end) pairs — the drop-in mode: one call drains a whole
 * batch of Spec::getSpec jobs (spec.cpp:18-42, 68-97).  ranges = count x {start,end}
 * (host).  mags_out = count x N/2 f32 (host, may be NULL); pitch_out = count
 * records (host, may be NULL).  Blocks until the results are in host memory. */
int mx_stft_ranges(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *ranges, int64_t count,
                   int kmin, int kmax, float *mags_out, mx_pitch *pitch_out);

/* Uniform hop — the bulk mode: frame h in [first_frame, first_frame+count) is
 * (start,end) = (h*hop, (h+1)*hop), i.e. the UI's column indexing
 * (spec-cache.cpp:12,63-65) with an identity time map and `hop` samples per pixel.
 * Host outputs as above; blocks. */
int mx_stft_hop(mx_ctx *ctx, const mx_audio *a, int N, int hop, int64_t first_frame, int64_t count,
                int kmin, int kmax, float *mags_out, mx_pitch *pitch_out);

/* Same, outputs stay in HBM (d_mags: count x N/2 f32, d_pitch: count records;
 * either may be NULL).  Asynchronous on the ctx stream. */
int mx_stft_hop_dev(mx_ctx *ctx, const mx_audio *a, int N, int hop, int64_t first_frame,
                    int64_t count, int kmin, int kmax, float *d_mags, mx_pitch *d_pitch);
int mx_stft_ranges_dev(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *d_ranges,
                       int64_t count, int kmin, int kmax, float *d_mags, mx_pitch *d_pitch);

/* Fused colormap (SpecCache::populateTex, spec-cache.cpp:77-96): the STFT
 * kernel's epilogue applies  magnitude * k -> clamp -> 3-segment RGB8  to the
 * row it has just produced, so texture rows leave the device as 3 bytes per bin
 * from ONE launch.  rgb = count x N/2 x 3 bytes.  The *_mags variant returns the
 * magnitude rows of the same launch as well (mags_out / d_mags may be NULL):
 * that is what a Spec worker feeding both getSpec and a SpecCache wants. */
int mx_stft_ranges_rgb(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *ranges, int64_t count,
                       float k, uint8_t *rgb_out);
int mx_stft_ranges_rgb_mags(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *ranges,
                            int64_t count, float k, float *mags_out, uint8_t *rgb_out);
int mx_stft_ranges_rgb_dev(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *d_ranges,
                           int64_t count, float k, float *d_mags, uint8_t *d_rgb);
/* The colormap alone on device-resident magnitude rows (nbins_total a multiple
 * of 4), e.g. to re-colour cached rows after the user changed k (app.cpp:75). */
int mx_colormap_dev(mx_ctx *ctx, const float *d_mags, int64_t nbins_total, float k, uint8_t *d_rgb);

/* Device-resident magnitude rows — the device-side row cache of a Spec worker (the reference keeps its
 * rows in host memory, spec.cpp:18-42).  mx_stft_ranges_keep is mx_stft_ranges_rgb_mags whose magnitude rows STAY in HBM
 * (*rows_out, count rows; release with mx_rows_free) whatever comes back to the host: rgb_out (k != 0) and mags_out
 * may each be NULL.  mx_rows_fetch copies rows [first, first+count) of a kept batch to the host; mx_rows_colormap returns
 * their texel rows for a scale k by running the colormap alone (no transform) — what a changed brightness (app.cpp:75)
 * or a late getSpec of a texel-only column costs instead of a second STFT. */
typedef struct mx_rows mx_rows;
int mx_stft_ranges_keep(mx_ctx *ctx, const mx_audio *a, int N, const int32_t *ranges, int64_t count, float k,
                        float *mags_out, uint8_t *rgb_out, mx_rows **rows_out);
int64_t mx_rows_count(const mx_rows *rows);
void mx_rows_free(mx_ctx *ctx, mx_rows *rows);
int mx_rows_fetch(mx_ctx *ctx, const mx_rows *rows, int64_t first, int64_t count, float *mags_out);
int mx_rows_colormap(mx_ctx *ctx, const mx_rows *rows, int64_t first, int64_t count, float k, uint8_t *rgb_out);

/* Number of frames of the bulk indexing: ceil(n / hop). */
int64_t mx_frame_count(int64_t n, int hop);

/* ---- time maps (host, cold-cache pure functions of the marker list) -------
 * Replace App::sample2Time / time2Sample / time2PitchBend / duration
 * (app.cpp:1020-1122).  markers must be sorted by sample. */
double mx_sample2time(const mx_marker *markers, int nmarkers, int sampleRate, int val);
int mx_time2sample(const mx_marker *markers, int nmarkers, int sampleRate, double val);
double mx_duration(const mx_marker *markers, int nmarkers, int sampleRate, int64_t nsamples);
float mx_time2pitchbend(const mx_marker *markers, int nmarkers, int sampleRate, int64_t nsamples,
                        double val);
/* SpecCache::getTex key + populateTex range (spec-cache.cpp:12, 63-65). */
void mx_column_range(const mx_marker *markers, int nmarkers, int sampleRate, double time,
                     int screenWidth, double rangeTime, int *key, int *start, int *end);

/* ---- grains + resynthesis schedule ------------------------------------------
 * mx_grains replaces the grain scan of App::preproc (app.cpp:153-235).
 * starts/lens are library-allocated (free with mx_free). */
int mx_grains(const float *host_wav, int64_t n, int32_t **starts, int32_t **lens, int64_t *count);
/* Device version: zero-crossing predicates evaluated on the GPU from the
 * resident audio; same outputs. */
int mx_grains_dev(mx_ctx *ctx, const mx_audio *a, int32_t **starts, int32_t **lens, int64_t *count);
/* The same chain as a grain TABLE: additionally firsts[g] = wav[starts[g]], the only samples the export loop reads
 * (App::process's nextGrainFirstSample, app.cpp:323-328) — with it mx_schedule_build_table needs no host copy of
 * the audio.  The chain itself is built on the device (ranks of the crossings, successor of every crossing, binary
 * lifting, expansion from start 0): only the table comes back.  firsts may be NULL. */
int mx_grain_table_dev(mx_ctx *ctx, const mx_audio *a, int32_t **starts, int32_t **lens, float **firsts, int64_t *count);

/* Replays the cursor recurrence of App::exportWav / App::process
 * (app.cpp:1200-1207, 294-331) on the host: one mx_step per process() call that
 * finds a grain.  *nsamples = sum(sz) + 1500 (the terminating call appends
 * preferredGrainSize zeros, app.cpp:303-309).  steps is library-allocated. */
int mx_schedule_build(const float *host_wav, int64_t n, int sampleRate, const int32_t *grain_starts,
                      const int32_t *grain_lens, int64_t ngrains, const mx_marker *markers,
                      int nmarkers, mx_step **steps, int64_t *nsteps, int64_t *nsamples);
/* The same chain of process() calls from any warped time: App::playback's refill loop
 * (app.cpp:272-274, `while (restWav.size() < need) tmpCursor += process(tmpCursor, restWav)` from an
 * empty restWav).  need >= 0: stop once *nsamples >= need; a call that finds no grain left adds 1500
 * zeros and leaves the cursor where it is, as often as the loop asks (the zeros are the tail of the
 * PCM, not steps).  need < 0: run to the end as mx_schedule_build does, from cursor0.
 * *cursor_end (may be NULL): the loop's cursor on exit = where the next refill continues. */
int mx_schedule_build_from(const float *host_wav, int64_t n, int sampleRate, const int32_t *grain_starts,
                           const int32_t *grain_lens, int64_t ngrains, const mx_marker *markers,
                           int nmarkers, double cursor0, int64_t need, mx_step **steps,
                           int64_t *nsteps, int64_t *nsamples, double *cursor_end);
/* mx_schedule_build_from on a grain table (mx_grain_table_dev) instead of the audio itself. */
int mx_schedule_build_table(int64_t n, int sampleRate, const int32_t *grain_starts, const int32_t *grain_lens,
                            const float *grain_firsts, int64_t ngrains, const mx_marker *markers, int nmarkers,
                            double cursor0, int64_t need, mx_step **steps, int64_t *nsteps, int64_t *nsamples,
                            double *cursor_end);
void mx_free(void *p);

/* Gather-lerp resampler + float->int16 (app.cpp:332-343, 1209-1212) over a
 * precomputed schedule.  pcm_f32_out / pcm_i16_out: nsamples each (host, either
 * may be NULL).  Bit-exact vs the reference arithmetic (no FMA contraction). */
int mx_resynth(mx_ctx *ctx, const mx_audio *a, const mx_step *steps, int64_t nsteps,
               int64_t nsamples, float *pcm_f32_out, int16_t *pcm_i16_out);
/* Device-resident variant: d_steps / outputs in HBM, asynchronous.
 * PRECONDITION (not checked on the device — mx_resynth, the host-pointer entry point, does check it): the steps are
 * in output order and contiguous, out_offset[0] == 0 and out_offset[i+1] == out_offset[i] + sz[i], as every
 * mx_schedule_build* call produces them (a rank's slice of a schedule rebased to its own buffer, shard_schedule in
 * melonix_amd/shard.py, qualifies).  Then every one of the nsamples outputs is written: step i fills its sz[i]
 * samples, and the samples past the LAST step record's run (the zeros of the terminating process() calls,
 * app.cpp:303-309) are cleared on the device.  With gaps or reordered records the gaps keep their previous contents. */
int mx_resynth_dev(mx_ctx *ctx, const mx_audio *a, const mx_step *d_steps, int64_t nsteps,
                   int64_t nsamples, float *d_pcm_f32, int16_t *d_pcm_i16);

/* The tail of App::exportWav (app.cpp:1209-1214) for a schedule that is already built and audio that is already
 * on the device: resynthesis to int16 and saveWav, with the PCM streamed device -> pinned pieces -> file (it never
 * exists as one host buffer).  File bytes = mx_resynth's int16 output through mx_save_wav. */
int mx_resynth_to_wav(mx_ctx *ctx, const mx_audio *a, const mx_step *steps, int64_t nsteps,
                      int64_t nsamples, const char *path, int sampleRate, int strict_reference_header);

/* Whole App::exportWav (app.cpp:1194-1215): grains -> schedule -> GPU resynth
 * -> int16 -> saveWav.  strict_reference_header!=0 reproduces save-wav.cpp:43. */
int mx_export_wav(mx_ctx *ctx, const float *host_wav, int64_t n, int sampleRate,
                  const mx_marker *markers, int nmarkers, const char *path,
                  int strict_reference_header);

/* ---- phase-vocoder pitch shift (BUILD-DEFINED) ------------------------------------
 * The reference has no phase vocoder (its pitch shift is the granular resampler above); BASELINE.json's
 * north_star names one, so the build defines it: N = 4096, synthesis hop 256, periodic Hann analysis and
 * synthesis windows, time-stretch by r = 2^(semitones/12) with integer phase propagation and identity phase
 * locking (spectral peaks carry the phase, every other bin rides on its nearest peak), overlap-add,
 * linear resampling by r back to the input length (definition: oracle/pv_oracle.py).  Constant shift over
 * the whole file; output has mx_audio_length(a) samples.  int16 = (int16)(clamp(v,-1,1) * 32767.).
 * Parity is unpinned by construction: the only oracle is the build's own CPU restatement. */
int mx_pv_pitch_shift(mx_ctx *ctx, const mx_audio *a, double semitones, float *pcm_f32_out,
                      int16_t *pcm_i16_out);
/* Same, outputs stay in HBM (either may be NULL); blocks until done.
 * WORK ARENA: one device allocation with a BUDGET, made at the first call, kept by the context for the next one (regrown
 * only when a call needs another shape) and released by mx_ctx_release_scratch / mx_ctx_destroy.
 *   budget   mx_pv_set_arena_budget(ctx, bytes); 0 (the default) = the environment variable MELONIX_PV_ARENA_MB if set,
 *            else a quarter of what hipMemGetInfo reports free when the context first needs an arena (taken once, kept
 *            until mx_ctx_release_scratch).  mx_pv_arena_budget: the budget in force.  An arena above a newly set budget
 *            is given back at once.
 *   resident a call whose frames fit the budget (22 KiB per frame: spectra 16 KiB, peak records 4 KiB, the stretched signal,
 *            maps and plan rows; 17.9 GB for an hour at +3 semitones) is ONE chunk: analysis, phase recurrence and synthesis
 *            each run once over the whole call, and a rank of a multi-GPU run (below) analyses its frames once.
 *   chunked  what does not fit (8 h on one GPU) is walked in chunks — the longest multiple of 32 frames of which two
 *            slots of spectra + records and a ring of four stretched-signal buffers (47 KiB per frame of a chunk) fit
 *            the budget.  Chunks meet on multiples of 32 frames — the synthesis workgroups — and hand each other the
 *            phase row and the overlap-add seam the way the ranks of a multi-GPU run do: the output is bit-identical
 *            whatever the chunk length, resident included.  While a chunked call runs, two internal streams carry the
 *            phase recurrence and the fix-up / resampling beside the transforms; both are joined before the call returns.
 *   records  the peak records are packed (room for 512 peaks per frame on average over an analysis workgroup's 8 or 16
 *            frames; a frame can have 2048).  A signal with more (an impulse train) makes the call repeat itself once,
 *            transparently, on an arena with full-size record regions (34 / 71 KiB per frame), which the context then keeps
 *            until mx_ctx_release_scratch; the samples are the same either way.  MELONIX_PV_FULL_RECORDS=1 starts there.
 *   MX_ERR_NOMEM (checked against hipMemGetInfo before allocating) if the device cannot give the arena, or if the
 *            budget is below what the smallest chunks need (two slots of 32 frames: 3.5 MiB).
 * mx_pv_set_chunk_frames(frames > 0) overrides the policy with two-slot chunks of exactly that length (rounded up to a
 * multiple of 32; 0 = back to the budget; the environment variable MELONIX_PV_CHUNK_FRAMES likewise): a test that wants
 * many chunk boundaries in a short signal.  mx_pv_arena_bytes: what the context holds right now (0 before the first
 * call).  mx_pv_last_chunks: the number of chunks the last run over this arena took (1 = resident). */
int mx_pv_set_chunk_frames(mx_ctx *ctx, int64_t frames);
int mx_pv_set_arena_budget(mx_ctx *ctx, int64_t bytes);
int64_t mx_pv_arena_budget(mx_ctx *ctx);
int64_t mx_pv_arena_bytes(mx_ctx *ctx);
int64_t mx_pv_last_chunks(mx_ctx *ctx);
int mx_pv_pitch_shift_dev(mx_ctx *ctx, const mx_audio *a, double semitones, float *d_pcm_f32,
                          int16_t *d_pcm_i16);

/* The same vocoder steered by the editor's markers the way App::exportWav is (app.cpp:1194-1207, 296-301): the
 * output runs over warped time t in [0, duration()); around warped time t the source is read at time2Sample(t)
 * and shifted by 2^(time2PitchBend(t)/12), the bend taken constant over a frame's hop (definition:
 * oracle/pv_oracle.py marker_plan / render).  Still build-defined, parity unpinned.
 * mx_pv_render_length: the number of output samples (those with i/sampleRate < duration()), < 0 on error. */
int64_t mx_pv_render_length(int64_t n, int sampleRate, const mx_marker *markers, int nmarkers);
/* The frame plan itself (library-allocated, free each with mx_free): per frame the analysis centre, warped time and
 * ratio, and i0[f] = first output sample of frame f (frames + 1 entries, the last = *nsamples). */
int mx_pv_plan(int64_t n, int sampleRate, const mx_marker *markers, int nmarkers, int64_t **apos, double **tf,
               double **rf, int64_t **i0, int64_t *frames, int64_t *nsamples);
int mx_pv_render(mx_ctx *ctx, const mx_audio *a, int sampleRate, const mx_marker *markers, int nmarkers,
                 float *pcm_f32_out, int16_t *pcm_i16_out);
int mx_pv_render_dev(mx_ctx *ctx, const mx_audio *a, int sampleRate, const mx_marker *markers, int nmarkers,
                     float *d_pcm_f32, int16_t *d_pcm_i16);

/* One rank of a multi-GPU phase-vocoder run (the overlap-add seams).  Every rank holds the whole
 * input and takes a contiguous range of frames (boundaries on multiples of 32 frames, so the sums group exactly
 * as in a single-GPU run and the concatenated outputs are bit-identical to mx_pv_pitch_shift's).  The caller does
 * the two small exchanges between the stages with its own collective (melonix_amd/shard.py: RCCL / gloo
 * all-gathers): the per-rank phase totals after stage 1, the seams after stage 2.  A rank works inside the same budgeted
 * arena as a single GPU (above): a range that fits the budget stays RESIDENT between the stages and is analysed once
 * (stage 1 = analysis + maps, stage 2 = offsets + synthesis from the rows stage 1 left); only a range beyond the budget is
 * walked in chunks, and then analysed twice (stage 1 keeps the maps only, stage 2 analyses again with the carry).  Every
 * rank needs at least 32 frames of its own.
 * Two forms.  Host pointers (mx_pv_shard_analyze / _synthesize / _finish): the exchanged rows travel through host memory
 * and the rank's outputs wait in library-owned device buffers (6 bytes per output sample) until stage 3 downloads them.
 * Device pointers (the _dev forms): everything a rank exchanges stays in HBM, laid out as the two all-gathers move it — each
 * stage writes the rank's entry of the next all-gather's send buffer and reads the previous all-gather's receive buffer
 * as it is — and the PCM goes straight into the caller's device buffers.
 *   mx_pv_shard_frames      the rank's frame range and the output samples [out_lo, out_hi) it will deliver
 *   mx_pv_shard_analyze     stage 1; tot_sums_out[2048] / tot_org_out[2048]: this rank's frames as one map of the
 *                           phase row: bin k ends at value[org[k]] + sums[k] (mod 2^32), or at sums[k] where
 *                           org[k] = 0xFFFF (the bin restarted inside the rank)
 *   mx_pv_shard_synthesize  stage 2; carry_in[2048]: the phase row the rank starts from = the maps of all lower
 *                           ranks applied in order to a zero row (NULL on rank 0);
 *                           head_out / tail_out[3840]: raw partial sums either side of the rank's frames
 *   mx_pv_shard_finish      stage 3; prev_tail = rank-1's tail_out (NULL on rank 0), next_head = rank+1's head_out
 *                           (NULL on the last rank); out_hi-out_lo samples each (host, either may be NULL)
 *   mx_pv_shard_analyze_dev     stage 1; d_map_out: 12 KiB on the device = 2048 uint32 sums then 2048 uint16 source bins
 *   mx_pv_shard_synthesize_dev  stage 2; d_maps_all: [world] x 12 KiB, the all-gathered stage-1 entries (the maps of the
 *                               ranks below are folded into the rank's carry on the device); d_pcm_f32 / d_pcm_i16:
 *                               out_hi - out_lo samples each on the device (either may be NULL), complete but for the
 *                               rank's edges; d_seams_out: 30 KiB = head then tail, 3840 floats each
 *   mx_pv_shard_finish_dev      stage 3; d_seams_all: [world] x 30 KiB, the all-gathered stage-2 entries; fills the edges
 *                               of the PCM buffers given to stage 2
 * Every stage blocks until its device work is done. */
int mx_pv_shard_frames(int64_t n, double semitones, int rank, int world, int64_t *frame_lo, int64_t *frame_hi,
                       int64_t *out_lo, int64_t *out_hi);
int mx_pv_shard_analyze(mx_ctx *ctx, const mx_audio *a, double semitones, int rank, int world,
                        uint32_t *tot_sums_out, uint16_t *tot_org_out);
int mx_pv_shard_synthesize(mx_ctx *ctx, const uint32_t *carry_in, float *head_out, float *tail_out);
int mx_pv_shard_finish(mx_ctx *ctx, const float *prev_tail, const float *next_head, float *pcm_f32_out,
                       int16_t *pcm_i16_out);
int mx_pv_shard_analyze_dev(mx_ctx *ctx, const mx_audio *a, double semitones, int rank, int world, void *d_map_out);
int mx_pv_shard_synthesize_dev(mx_ctx *ctx, const void *d_maps_all, float *d_pcm_f32, int16_t *d_pcm_i16,
                               void *d_seams_out);
int mx_pv_shard_finish_dev(mx_ctx *ctx, const void *d_seams_all);

/* ---- waveform min/max pyramid ---------------------------------------------------
 * Replaces App::calcPicks (app.cpp:347-378): level l = floor(n / 2^(l+1)) {min,max} pairs over blocks
 * of 2^(l+1) samples, for every l with n > 2^(l+1).  picks_out (host, caller-allocated, 2*n floats
 * always suffice) receives the levels one after the other as interleaved {min,max}; counts_out
 * (>= 64 entries) the pairs per level; *nlevels the number of levels. */
int mx_minmax_pyramid(mx_ctx *ctx, const mx_audio *a, float *picks_out, int64_t *counts_out, int *nlevels);
/* Same, the pairs stay in HBM (d_picks: 2*n floats of capacity); counts_out / nlevels are host. */
int mx_minmax_pyramid_dev(mx_ctx *ctx, const mx_audio *a, float *d_picks, int64_t *counts_out, int *nlevels);
/* Replaces App::getMinMaxFromRange (app.cpp:380-426) over such a pyramid (host), quirks included. */
void mx_minmax_range(const float *host_wav, int64_t n, const float *picks, const int64_t *counts, int nlevels,
                     int start, int end, float *mn, float *mx);

/* ---- WAV writer -------------------------------------------------------------
 * Replaces saveWav (save-wav.cpp:17-48).  strict_reference_header != 0
 * reproduces the size-field quirk of save-wav.cpp:43 byte for byte (data size
 * = 2m+16, PCM samples 0 and 1 zeroed); 0 writes a correct RIFF header. */
int mx_save_wav(const char *path, const int16_t *pcm, int64_t m, int sampleRate,
                int strict_reference_header);

#ifdef __cplusplus
}
#endif
#endif /* MELONIX_AMD_H */
