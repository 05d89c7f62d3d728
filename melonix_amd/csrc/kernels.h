// kernels.h — internal launch interface between the C-ABI layer (capi_*.cpp) and
// the gfx950 kernels (*.hip).  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/melonix_amd.h"

namespace mx {

enum StftMode {
  kBulkAligned = 0,  // uniform hop, even hop: 64-bit sample/weight loads
  kBulkAny = 1,      // uniform hop, odd hop
  kRanges = 2,       // per-frame (start,end) list
};

struct StftArgs {
  const float *audio;  // padded image: [MX_AUDIO_PAD zeros][n][MX_AUDIO_PAD zeros]
  int64_t n;
  const float *wtab;      // bulk: N forward weights, pre-scaled by 1/(2N)
  const float *wext;      // ranges: d-indexed weights, pre-scaled by 1/(2N) (stft_tables.h)
  float decay;            // exp(-2.5e-4*hop): per-hop decay of the sliding window
  const float2 *tw2;      // pass-2 twiddles
  const float2 *tw3;      // pass-3 twiddles
  const float2 *ubase;    // post-split bases
  const int32_t *ranges;  // ranges mode: count x {start,end} (device)
  int hop;
  int64_t first_frame;
  int64_t count;
  int kmin, kmax;
  float *mags;       // count x N/2, may be null
  mx_pitch *pitch;   // count, may be null
  uint8_t *rgb;      // count x N/2 x 3 (fused colormap), may be null
  float cmap_k;
  int frames_per_block;
};

hipError_t launch_stft(int N, int mode, const StftArgs &a, hipStream_t s);
// Points per thread of the plan launch_stft uses for N (selects the twiddle tables to upload).
constexpr int kPlan4096E = 16;
int stft_points_per_thread(int N);
// Longest run of consecutive frames worth giving one workgroup for (N, mode, hop): the kernels that carry a register
// image from frame to frame amortise their first frame's direct load over the run.
int stft_frames_per_block_cap(int N, int mode, int hop);

struct ResynthArgs {
  const float *audio;  // padded image
  const mx_step *steps;
  int64_t nsteps;
  int64_t nsamples;    // total incl. trailing zeros
  float *pcm_f32;      // may be null
  int16_t *pcm_i16;    // may be null
};
hipError_t launch_resynth(const ResynthArgs &a, hipStream_t s);

// Build-defined phase-vocoder pitch shifter (pv_kernels.hip; no reference counterpart).
struct PvArgs {
  const float *audio;  // padded image (zeros in the pads)
  int64_t n;
  double ratio;            // r = 2^(semitones/12)
  const int64_t *apos;     // analysis centres a_f = floor(f*Hs/r), frames entries
  const uint32_t *hop;     // a_f - a_{f-1}; 0 for local row 0 of the whole signal and where a plan does not advance
  const double *hratio;    // Hs / hop (binary64 quotient), 0 where hop is 0
  int64_t frames;
  const float *hann_scaled;  // periodic Hann * 1/(2N): the analysis window with the transform's folded scale
  const float *hann;         // periodic Hann (synthesis window)
  const float2 *wsplit;      // e^{+2 pi i c/N}, c = 0..N/2-1 (pre-split of the inverse transform)
  const float2 *tw2, *tw3, *ubase;  // Plan<4096,16> tables
  // Per frame (row) the analysis leaves: the one-sided spectrum X/N, complex (what synthesis rotates), the frame's spectral
  // peaks as a 2048-bit map, and one 8-byte record per peak, in bin order — everything the phase recurrence needs of the
  // frame (the recurrence runs over peaks only; every other bin rides on its owner peak's phasor):
  //   rec.x = p | q << 11 | qvalid << 22 | cont << 23   p: the peak's bin; q: the peak of the PREVIOUS frame that owns bin p
  //                                                     (nearest within reach); cont: p carried signal in both frames and h >= 1
  //   rec.y = delta = P_{f-1}[p] + inc_f[p] - P_f[p]    (uint32 turns; inc: pv_inc)
  // so that the peak's synthesis offset is  C_f[p] = E_{f-1}[p] + delta  (cont) with E_{f-1}[p] = C_{f-1}[q] where q
  // continued itself, else 0; a peak that does not continue restarts (offset 0: its bins keep their analysis phases).
  float2 *xrows;       // [frames][N/2]
  uint32_t *pkmap;     // [frames][N/64] bit k: bin k is a spectral peak of the frame
  uint32_t *pkcount;   // [frames] bits 0..11: number of peaks (= records) of the frame; bits 12..: where they start inside the
                       // frame's analysis workgroup's region of the record pool (below)
  float *fthr;         // [frames] the frame's activity threshold (squared magnitude): what the next frame's records compare with
  uint2 *recs;         // the frames' records, in the pool; the second sweep REPLACES rec.y by C_f of
                       // the peak (0 = restarted): delta has no reader after it.  Every analysis workgroup (frames_per_block
                       // consecutive frames) packs its frames' records one behind the other into a region of rec_wg_cap entries
                       // of its own (no atomics: the layout is a function of the signal).  A region that does not hold its
                       // frames (rec_wg_cap < frames_per_block * N/2 is a bet on the signal: an impulse makes every bin a peak)
                       // raises *rec_overflow: the caller discards the run and repeats it with full-sized regions
  uint32_t rec_wg_cap; // records per analysis workgroup's region
  int rec_fpb_shift;   // log2(frames_per_block of the analysis launch): local frame f lies in region f >> rec_fpb_shift
  uint32_t *rec_overflow;  // [1] set to 1 by an analysis workgroup whose region is too small (the run's results are void)
  uint32_t *chunk_sums;  // [ceil(frames/scan_chunk)][N/2] a chunk's composed map: delta (or restart value) ...
  uint16_t *chunk_org;   // ... and source bin at the chunk's start (0xFFFF: restart); then the chunk-start offsets
  uint32_t *group_sums;  // [ceil(chunks/32)][N/2] the same for groups of 32 chunks (the composition is two-level)
  uint16_t *group_org;
  int scan_chunk;
  float *halo;        // pv_halo_floats(frames): partial sums right of each synthesis-workgroup boundary
  float *s;           // stretched signal, s_len = frames*Hs + N, index 0 = stretched time -N/2
  int64_t s_len;
  float *pcm_f32;     // output sample i of the whole signal is pcm[i - pcm_base]; may be null
  int16_t *pcm_i16;   // likewise
  int64_t pcm_base;
  int frames_per_block;
  // multi-GPU (one rank's part of the frame axis; all zero / null for a whole-signal run)
  int64_t first;             // local frames [first, frames) are this rank's; first = 1: row 0 is the frame before them
  int global_first;          // this rank holds the signal's frame 0
  const uint32_t *carry_in;  // [N/2] synthesis phase at the end of the previous rank's / chunk's last frame (null: zero)
  uint32_t *carry_out;       // [N/2] out: the same row at the end of THIS range's last frame (null: not wanted)
  uint32_t *tot_sums;        // [N/2] out: this rank's total map over its frames (null: not wanted): delta / value ...
  uint16_t *tot_org;         // ... and source bin at the rank's start (0xFFFF: restart)
  const float *prev_tail;    // [N-Hs] the previous rank's tail seam (raw sums), null on the first rank
  const float *next_head;    // [N-Hs] the next rank's head seam, null on the last rank
  const float *prev_final;   // [N-Hs] the previous chunk's FINISHED samples across boundary 0 (one GPU walking the signal chunk by
                             // chunk: the previous chunk's fix-up already added the two sides and normalised); overrides prev_tail
  int skip_head, skip_tail;  // pv_fixup leaves boundary 0 / the last boundary alone (a rank's edges wait for its neighbours' seams)
  int64_t out_lo, out_hi;    // output samples [out_lo, out_hi) of the whole signal are resampled here
  int64_t s_origin;          // stretched sample index (incl. the N/2 offset) of s[0]
  // marker-driven plan (null for a constant ratio): per frame warped time, ratio and first output sample
  // (indexed like the frames this range synthesises: entry 0 is local frame `first`)
  const double *tf, *rf;
  const int64_t *i0;         // one entry more than frames
  int64_t frame_base;        // index in the whole signal of local frame `first`
  int sample_rate;
};
hipError_t launch_pv(const PvArgs &a, hipStream_t s);
hipError_t launch_pv_analyze(const PvArgs &a, hipStream_t s);
hipError_t launch_pv_synthesize(const PvArgs &a, hipStream_t s);
// the same stages in the pieces the chunked pipeline puts on streams of their own (pv_kernels.hip: launch_pv_analyze =
// analysis + maps, launch_pv_synthesize = offsets + synthesis)
hipError_t launch_pv_analysis(const PvArgs &a, hipStream_t s);
hipError_t launch_pv_maps(const PvArgs &a, hipStream_t s);
hipError_t launch_pv_offsets(const PvArgs &a, hipStream_t s);
hipError_t launch_pv_synthesis(const PvArgs &a, hipStream_t s);
hipError_t launch_pv_finish(const PvArgs &a, hipStream_t s);
int64_t pv_halo_floats(int64_t frames);
// the composition, in order, of n maps -> one; sums_stride / org_stride: elements between consecutive maps' rows (0: dense, N/2)
hipError_t launch_pv_compose_maps(const uint32_t *sums, const uint16_t *org, int64_t n, uint32_t *out_sums, uint16_t *out_org, hipStream_t s,
                                  int sums_stride = 0, int org_stride = 0);
hipError_t launch_pv_resample(const PvArgs &a, hipStream_t s);
hipError_t launch_pv_edge_sum(float *dst, const float *x, const float *y, int n, hipStream_t s);
// constant-ratio analysis plan written on the device: rows of apos / hop / hratio for global frames fbase, fbase+1, ...
hipError_t launch_pv_plan_const(int64_t *apos, uint32_t *hop, double *hratio, int64_t rows, int64_t fbase, double r,
                                hipStream_t s);

// spec-cache.cpp:77-96 colormap: nbins_total magnitudes -> 3*nbins_total bytes (both device).
hipError_t launch_colormap(const float *mags, uint8_t *rgb, int64_t nbins_total, float k, hipStream_t s);

// App::calcPicks (app.cpp:347-378): all levels, level after level, into d_out ({min,max} pairs; n pairs
// of capacity always suffice); counts[l] = pairs in level l (host array of >= 64), *nlevels = levels.
hipError_t launch_picks(const float *audio_padded, int64_t n, float *d_out, int64_t *counts, int *nlevels, hipStream_t s);

// Zero-crossing predicate bitmaps for grain segmentation (app.cpp:167-181, 202-216).
hipError_t launch_zc_bitmaps(const float *audio_padded, int64_t n, uint64_t *zc7, uint64_t *zc3,
                             hipStream_t s);

// The grain chain of App::preproc on the device (grain_chain.hip): phase 1 ranks the look-around-3 bits (read header word 2
// = node count back), phase 2 builds the chain; header word 1 = grain count, table = starts / lens / first samples.
size_t grain_rank_scratch_bytes(int64_t n);
hipError_t launch_grain_rank(const float *audio_padded, int64_t n, const uint64_t *zc7, const uint64_t *zc3, void *rank_scratch,
                             hipStream_t s);
void grain_chain_sizes(int64_t n, uint32_t nodes, int *levels_out, uint32_t *out_cap_out, size_t *bytes_out);
hipError_t launch_grain_chain(const float *audio_padded, int64_t n, const uint64_t *zc7, const uint64_t *zc3, void *rank_scratch,
                              uint32_t nodes, void *chain_scratch, int32_t **d_starts, int32_t **d_lens, float **d_firsts,
                              hipStream_t s);

}  // namespace mx
