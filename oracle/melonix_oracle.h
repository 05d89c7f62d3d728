/*
 * melonix_oracle.h — CPU restatement of the melonix reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under melonix_amd/ (the product) may
 * include, link, dlopen or import this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call it, and only as the checker /
 * reported CPU baseline — never as the thing measured as the GPU path.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it restates.  The restatement is plain C99 + libm, built
 * with -O2 -ffp-contract=off (the reference's float expressions must not be
 * fused into FMAs — see SURVEY.md §7 "Bit-exact schedule").
 *
 * PINNING STATUS (also in DESIGN.md §3):
 *   - saveWav        : pinned against the reference itself (oracle/_ref builds
 *                      /root/reference/save-wav.cpp unmodified; byte-compared).
 *   - grains/process/export/time maps : pinned against the known-answer facts
 *                      the survey recorded from the compiled reference
 *                      (319 grains; 320/379/254/491 steps;
 *                      480407/478903/479781/479189 samples; WAV size+quirk).
 *   - STFT magnitudes: the DFT arithmetic lives in FFTW3 (third-party, not in
 *                      /root/reference, version unpinned by the reference:
 *                      README.md:9 "libfftw3-dev").  The reference has no
 *                      tests / golden vectors, and spec.cpp cannot be built
 *                      here without stand-in headers (fftw3.h, log/log.hpp),
 *                      which the rules forbid.  The restatement implements the
 *                      published definition FFTW documents for
 *                      fftw_plan_dft_1d(FFTW_FORWARD): the unnormalised
 *                      forward DFT in double; it is cross-checked against an
 *                      independent f64 FFT (numpy pocketfft) and analytic
 *                      DFT pairs.  By the reference's own tests: PARITY UNPINNED.
 *   - pitch pick     : build-defined (the reference has no detector); oracle
 *                      is argmax over oracle magnitudes.  PARITY UNPINNED.
 */
#ifndef MELONIX_ORACLE_H
#define MELONIX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* marker.hpp:4-9 */
typedef struct mxo_marker {
  int sample;
  double note;
  double dTime;
  double pitchBend;
} mxo_marker;

/* ---- STFT (spec.cpp) -------------------------------------------------- */

/* Unnormalised forward c2c DFT, double, size N (power of two), interleaved
 * re/im.  Stands where fftw_execute stands (spec.cpp:15,60). */
int mxo_fft_c2c_f64(int N, const double *in, double *out);

/* spec.cpp:44-66 with SpectrSize generalised to N.  out has N/2 floats. */
int mxo_spec_frame(const float *wav, int n, int N, int start, int end, float *out);

/* Bulk uniform-hop STFT: frame h -> (start,end) = (h*hop,(h+1)*hop)
 * (SURVEY.md §8a "bulk frame indexing").  mags may be NULL (pitch only).
 * pitch_bin/pitch_mag may be NULL.  nthreads>=1 (pthreads, static partition). */
int mxo_stft_hop(const float *wav, int n, int N, int hop, long first_frame, long count,
                 int kmin, int kmax, float *mags, int32_t *pitch_bin, float *pitch_mag,
                 int nthreads);

/* The same two functions with the DFT executed by a library that implements the FFTW3 API, dlopen'ed at run time
 * (libfftw3.so.3 if present, else Intel MKL's FFTW3 interface; MXO_FFTW_LIB overrides): spec.cpp's own call
 * sequence fftw_plan_dft_1d(N, in, out, FFTW_FORWARD, FFTW_MEASURE) / fftw_execute on a production implementation.
 * Return -2 when no such library can be loaded.  mxo_fftw_api_name(): "fftw3", "mkl-fftw3-interface" or "none". */
const char *mxo_fftw_api_name(void);
int mxo_spec_frame_fftw_api(const float *wav, int n, int N, int start, int end, float *out);
int mxo_stft_hop_p(const float *wav, int n, int N, int hop, long first_frame, long count,
                   int kmin, int kmax, float *mags, int32_t *pitch_bin, float *pitch_mag,
                   int nthreads, int use_fftw_api);
/* Throughput probe for bench.py's cpu_baseline: the same frames, nothing stored; every thread makes its plan and
 * buffers first (FFTW-API planning is serialised), then all start together; *loop_seconds = first thread into its
 * frames .. last thread out (setup excluded, as the reference plans once per Spec, spec.cpp:11-15). */
int mxo_stft_hop_timed(const float *wav, int n, int N, int hop, long first_frame, long count,
                       int kmin, int kmax, int nthreads, int use_fftw_api, double *loop_seconds);


/* Build-defined pitch pick: argmax over k in [kmin,kmax],
 * ties -> lowest k. */
int mxo_pitch_pick(const float *mags, int nbins, int kmin, int kmax, int32_t *bin, float *mag);

/* Default pitch band: notes 24..84 => 55..1760 Hz (app.hpp:45-46, app.cpp:499-516). */
void mxo_pitch_band(int N, int sampleRate, int *kmin, int *kmax);

/* spec-cache.cpp:77-96 colormap: mags[nbins] * k -> rgb[3*nbins]. */
void mxo_colormap(const float *mags, int nbins, float k, unsigned char *rgb);

/* ---- time maps (app.cpp:1020-1122) ------------------------------------ */

typedef struct mxo_timemap mxo_timemap;
/* memo!=0 reproduces the reference's memo tables (keyed by int(val*sr)). */
mxo_timemap *mxo_timemap_new(const mxo_marker *markers, int nmarkers, int sampleRate,
                             long nsamples, int memo);
void mxo_timemap_free(mxo_timemap *);
double mxo_sample2time(mxo_timemap *, int val);    /* app.cpp:1020-1050 */
int mxo_time2sample(mxo_timemap *, double val);    /* app.cpp:1052-1082 */
double mxo_duration(mxo_timemap *);                /* app.cpp:1084-1087 */
float mxo_time2pitchbend(mxo_timemap *, double v); /* app.cpp:1089-1122 */

/* spec-cache.cpp:12,63-65: pixel key and (start,end) sample range of the column at `time`. */
void mxo_column_range(mxo_timemap *, double time, int width, double rangeTime, int *key,
                      int *start, int *end);

/* ---- grains (app.cpp:153-235) ----------------------------------------- */

/* Returns number of grains; *starts / *lens are malloc'd (caller frees with mxo_free). */
long mxo_grains(const float *wav, long n, int **starts, int **lens);
void mxo_free(void *);

/* ---- resynthesis (app.cpp:294-345, 1194-1215) -------------------------- */

typedef struct mxo_step {
  double cursor;     /* warped time at which process() was entered */
  int grain_start;   /* key of the chosen grain (source sample index) */
  int grain_len;     /* L */
  float rate;        /* powf(2, pb/12) */
  float next_first;  /* nextGrainFirstSample */
  int sz;            /* samples emitted */
  long out_offset;   /* exclusive prefix sum of sz */
} mxo_step;

typedef struct mxo_export {
  long nsteps;
  mxo_step *steps;
  long nsamples;     /* incl. the 1500 trailing zeros of the terminating process() call */
  float *pcm;
  double cursor_end; /* the loop's cursor after the last call */
} mxo_export;

/* Replays App::exportWav's loop (app.cpp:1200-1207) incl. the final
 * "no grain left" call that appends 1500 zeros (app.cpp:303-309). */
int mxo_export_run(const float *wav, long n, int sampleRate, const mxo_marker *markers,
                   int nmarkers, int memo, mxo_export *out);
/* Replays App::playback's refill loop (app.cpp:272-274) from an empty restWav: process() calls chained
 * from warped time cursor0 until at least `need` samples exist.  A call that finds no grain left appends
 * 1500 zeros and leaves the cursor where it is (app.cpp:303-309), as often as the loop asks. */
int mxo_playback_fill(const float *wav, long n, int sampleRate, const mxo_marker *markers,
                      int nmarkers, int memo, double cursor0, long need, mxo_export *out);
void mxo_export_free(mxo_export *);

/* app.cpp:1209-1212 */
void mxo_pcm_to_i16(const float *pcm, long m, int16_t *out);

/* save-wav.cpp:17-48 incl. the :43 size-field quirk (8-byte write at offset 40). */
int mxo_save_wav(const char *path, const int16_t *pcm, long m, int sampleRate);
/* Same bytes into memory; buf must hold 44+2*m (and at least 48) bytes; returns length. */
long mxo_wav_bytes(const int16_t *pcm, long m, int sampleRate, unsigned char *buf);

/* ---- waveform min/max pyramid (app.cpp:347-426) ------------------------- */
/* calcPicks: level l holds floor(n / 2^(l+1)) (min,max) pairs over blocks of 2^(l+1) samples, while
 * n > 2^(l+1).  Returns the number of levels; pairs are written level after level, interleaved
 * {min,max}, into out (capacity: n pairs is always enough); counts[l] = pairs in level l. */
int mxo_calc_picks(const float *wav, long n, float *out, long *counts, int max_levels);
/* getMinMaxFromRange(start, end) over a pyramid built by mxo_calc_picks (incl. its quirks). */
void mxo_minmax_range(const float *wav, long n, const float *picks, const long *counts, int nlevels,
                      int start, int end, float *mn, float *mx);

/* ---- synthetic input ----------------------------------- */
void mxo_sweep(float *out, long n, int sampleRate, double f0, double f1, double amp);

#ifdef __cplusplus
}
#endif
#endif
