/*
 * melonix_oracle.c — CPU restatement of the melonix reference hot path.
 * TEST INFRASTRUCTURE ONLY (see melonix_oracle.h for the rules and the
 * pinning status of each function).  Build: see oracle/Makefile
 * (gcc -O2 -ffp-contract=off -std=gnu99, no -march flags: the reference's
 * float expressions must evaluate as separate IEEE mul/add).
 *
 * All file:line citations are relative to /root/reference.
 */
#define _GNU_SOURCE
#include "melonix_oracle.h"

#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ======================================================================
 * Double-precision forward DFT — stands where FFTW stands
 * (spec.cpp:15 fftw_plan_dft_1d(N, in, out, FFTW_FORWARD, ...), spec.cpp:60).
 * FFTW's documented definition: Y[k] = sum_j X[j] * exp(-2*pi*i*j*k/N),
 * unnormalised.  Stockham autosort, radix 4 (+ one radix-2 pass when
 * log2 N is odd); twiddles from long-double cosl/sinl rounded to double.
 * ====================================================================== */

typedef struct fft_plan {
  int N;
  double *tw; /* tw[2k], tw[2k+1] = cos, -sin of 2*pi*k/N, k in [0,N) */
  struct fft_plan *next;
} fft_plan;

static fft_plan *g_plans = NULL;
static pthread_mutex_t g_plan_mu = PTHREAD_MUTEX_INITIALIZER;

static const fft_plan *plan_for(int N) {
  pthread_mutex_lock(&g_plan_mu);
  fft_plan *p = g_plans;
  while (p && p->N != N) p = p->next;
  if (!p) {
    p = (fft_plan *)malloc(sizeof(*p));
    p->N = N;
    p->tw = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < N; ++k) {
      long double a = two_pi * (long double)k / (long double)N;
      p->tw[2 * k] = (double)cosl(a);
      p->tw[2 * k + 1] = (double)(-sinl(a));
    }
    p->next = g_plans;
    g_plans = p;
  }
  pthread_mutex_unlock(&g_plan_mu);
  return p;
}

static inline void cmul(double ar, double ai, double br, double bi, double *cr, double *ci) {
  *cr = ar * br - ai * bi;
  *ci = ar * bi + ai * br;
}

/* One Stockham pass of radix R in {2,4}: Ns = size of finished sub-transforms. */
static void pass_r4(const fft_plan *pl, int Ns, const double *x, double *y) {
  const int N = pl->N, Q = N / 4;
  const int tstep = N / (Ns * 4);
  for (int j = 0; j < Q; ++j) {
    const int k = j % Ns;
    double v[4][2];
    for (int r = 0; r < 4; ++r) {
      const double xr = x[2 * (j + r * Q)], xi = x[2 * (j + r * Q) + 1];
      const int t = r * k * tstep; /* exp(-2*pi*i*r*k/(4*Ns)) */
      cmul(xr, xi, pl->tw[2 * t], pl->tw[2 * t + 1], &v[r][0], &v[r][1]);
    }
    const double a0r = v[0][0] + v[2][0], a0i = v[0][1] + v[2][1];
    const double a1r = v[0][0] - v[2][0], a1i = v[0][1] - v[2][1];
    const double a2r = v[1][0] + v[3][0], a2i = v[1][1] + v[3][1];
    const double a3r = v[1][0] - v[3][0], a3i = v[1][1] - v[3][1];
    const int j0 = (j / Ns) * Ns * 4 + k;
    y[2 * (j0)] = a0r + a2r;
    y[2 * (j0) + 1] = a0i + a2i;
    /* -i * a3 = (a3i, -a3r) */
    y[2 * (j0 + Ns)] = a1r + a3i;
    y[2 * (j0 + Ns) + 1] = a1i - a3r;
    y[2 * (j0 + 2 * Ns)] = a0r - a2r;
    y[2 * (j0 + 2 * Ns) + 1] = a0i - a2i;
    y[2 * (j0 + 3 * Ns)] = a1r - a3i;
    y[2 * (j0 + 3 * Ns) + 1] = a1i + a3r;
  }
}

static void pass_r2(const fft_plan *pl, int Ns, const double *x, double *y) {
  const int N = pl->N, H = N / 2;
  const int tstep = N / (Ns * 2);
  for (int j = 0; j < H; ++j) {
    const int k = j % Ns;
    const double ar = x[2 * j], ai = x[2 * j + 1];
    double br, bi;
    cmul(x[2 * (j + H)], x[2 * (j + H) + 1], pl->tw[2 * (k * tstep)], pl->tw[2 * (k * tstep) + 1],
         &br, &bi);
    const int j0 = (j / Ns) * Ns * 2 + k;
    y[2 * j0] = ar + br;
    y[2 * j0 + 1] = ai + bi;
    y[2 * (j0 + Ns)] = ar - br;
    y[2 * (j0 + Ns) + 1] = ai - bi;
  }
}

static int is_pow2(int N) { return N >= 2 && (N & (N - 1)) == 0; }

/* scratch must hold 2*N doubles; result lands in out. `in` is not modified. */
static int fft_exec(const fft_plan *pl, const double *in, double *out, double *scratch) {
  const int N = pl->N;
  int npass = 0;
  for (int m = N; m > 1;) {
    if (m % 4 == 0) m /= 4; else m /= 2;
    ++npass;
  }
  /* ping-pong so that the last pass writes `out` */
  const double *src = in;
  double *bufs[2] = {out, scratch};
  int which = (npass % 2 == 1) ? 0 : 1;
  int Ns = 1, rem = N;
  while (rem > 1) {
    double *dst = bufs[which];
    if (rem % 4 == 0) { pass_r4(pl, Ns, src, dst); Ns *= 4; rem /= 4; }
    else              { pass_r2(pl, Ns, src, dst); Ns *= 2; rem /= 2; }
    src = dst;
    which ^= 1;
  }
  return 0;
}

int mxo_fft_c2c_f64(int N, const double *in, double *out) {
  if (!is_pow2(N)) return -1;
  const fft_plan *pl = plan_for(N);
  double *scratch = (double *)malloc(sizeof(double) * 2 * (size_t)N);
  double *tmp_in = NULL;
  if (in == out) { /* out-of-place internally */
    tmp_in = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    memcpy(tmp_in, in, sizeof(double) * 2 * (size_t)N);
    in = tmp_in;
  }
  fft_exec(pl, in, out, scratch);
  free(scratch);
  free(tmp_in);
  return 0;
}

/* ======================================================================
 * Spec::internalGetSpec — spec.cpp:44-66, SpectrSize (spec.cpp:8) -> N
 * ====================================================================== */

/* ---- optional provider: a library that implements the FFTW3 API, loaded at run time --------------------
 * The reference calls fftw_plan_dft_1d(N, in, out, FFTW_FORWARD, FFTW_MEASURE) + fftw_execute (spec.cpp:11-15,
 * 60).  No FFTW header or library ships with the reference or the image, so nothing is linked: if the machine
 * has libfftw3.so.3, or Intel MKL's FFTW3 interface (libmkl_rt.so exports the same entry points), it is
 * dlopen'ed and the reference's very call sequence runs on it.  Used (a) to cross-check the built-in DFT
 * against a production implementation of that API and (b) as the CPU baseline's FFT, which is then of the class
 * the reference's FFTW path would have.  Plans are created under a lock (FFTW's planner is not thread-safe),
 * executed concurrently on per-thread buffers. */
typedef void *(*fftw_plan_fn)(int, void *, void *, int, unsigned);
typedef void (*fftw_exec_fn)(void *);
typedef void (*fftw_destroy_fn)(void *);
static struct {
  int tried, ok;
  const char *name;
  fftw_plan_fn plan;
  fftw_exec_fn exec;
  fftw_destroy_fn destroy;
  pthread_mutex_t mu;
} g_fftw = {0, 0, "none", NULL, NULL, NULL, PTHREAD_MUTEX_INITIALIZER};

static int fftw_api_load(void) {
  pthread_mutex_lock(&g_fftw.mu);
  if (!g_fftw.tried) {
    g_fftw.tried = 1;
    /* one FFT per host thread of ours: MKL must not bring its own thread pool (no effect on a real FFTW) */
    setenv("MKL_THREADING_LAYER", "SEQUENTIAL", 0);
    setenv("MKL_NUM_THREADS", "1", 0);
    const char *env = getenv("MXO_FFTW_LIB");
    const char *cands[] = {env, "libfftw3.so.3", "libfftw3.so", "libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so", NULL};
    for (int i = 0; i < 5 && !g_fftw.ok; ++i) {
      if (!cands[i] || !*cands[i]) continue;
      void *h = dlopen(cands[i], RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      g_fftw.plan = (fftw_plan_fn)dlsym(h, "fftw_plan_dft_1d");
      g_fftw.exec = (fftw_exec_fn)dlsym(h, "fftw_execute");
      g_fftw.destroy = (fftw_destroy_fn)dlsym(h, "fftw_destroy_plan");
      if (g_fftw.plan && g_fftw.exec && g_fftw.destroy) {
        g_fftw.ok = 1;
        g_fftw.name = strstr(cands[i], "mkl") ? "mkl-fftw3-interface" : "fftw3";
        /* one FFT per host thread: keep the library from spawning its own */
        void (*setn)(int) = (void (*)(int))dlsym(h, "MKL_Set_Num_Threads");
        if (setn) setn(1);
        /* first use from this one thread: a library that initialises itself lazily then does so before the
         * worker threads arrive (MKL initialised from eight threads at once ran five times slower afterwards) */
        double *tmp = (double *)calloc(4 * 64, sizeof(double));
        if (tmp) {
          void *pl = g_fftw.plan(64, tmp, tmp + 2 * 64, -1, 0u);
          if (pl) { g_fftw.exec(pl); g_fftw.destroy(pl); }
          free(tmp);
        }
      }
    }
  }
  const int ok = g_fftw.ok;
  pthread_mutex_unlock(&g_fftw.mu);
  return ok;
}
const char *mxo_fftw_api_name(void) { return fftw_api_load() ? g_fftw.name : "none"; }

typedef struct frame_ws {
  int N;
  const fft_plan *pl;
  double *in, *out, *scratch;
  void *fftw_plan; /* non-NULL: execute through the FFTW-API library */
} frame_ws;

static int ws_init_p(frame_ws *ws, int N, int use_fftw_api) {
  memset(ws, 0, sizeof *ws);
  if (!is_pow2(N)) return -1;
  ws->N = N;
  ws->pl = plan_for(N);
  ws->fftw_plan = NULL;
  ws->in = (double *)calloc(2 * (size_t)N, sizeof(double));
  ws->out = (double *)calloc(2 * (size_t)N, sizeof(double));
  ws->scratch = (double *)calloc(2 * (size_t)N, sizeof(double));
  if (!(ws->in && ws->out && ws->scratch)) return -1;
  if (use_fftw_api) {
    if (!fftw_api_load()) return -2;
    pthread_mutex_lock(&g_fftw.mu);
    ws->fftw_plan = g_fftw.plan(N, ws->in, ws->out, -1 /* FFTW_FORWARD */, 0u /* FFTW_MEASURE, spec.cpp:15 */);
    pthread_mutex_unlock(&g_fftw.mu);
    if (!ws->fftw_plan) return -2;
    memset(ws->in, 0, sizeof(double) * 2 * (size_t)N); /* FFTW_MEASURE may scribble on the buffers */
  }
  return 0;
}
static int ws_init(frame_ws *ws, int N) { return ws_init_p(ws, N, 0); }
static void ws_free(frame_ws *ws) {
  if (ws->fftw_plan) {
    pthread_mutex_lock(&g_fftw.mu);
    g_fftw.destroy(ws->fftw_plan);
    pthread_mutex_unlock(&g_fftw.mu);
  }
  free(ws->in); free(ws->out); free(ws->scratch);
}

static void spec_frame_ws(frame_ws *ws, const float *wav, int n, int start, int end, float *ret) {
  const int N = ws->N;
  int p = 0;
  /* spec.cpp:47-59 */
  for (int i = end - N; i < end; ++i, ++p) {
    ws->in[2 * p + 1] = 0;
    if (i >= n || i < 0) {
      ws->in[2 * p] = 0;
      continue;
    }
    if (i >= start)
      ws->in[2 * p] = wav[i];
    else
      ws->in[2 * p] = expf(-2.5e-4f * (start - i)) * wav[i]; /* float expr, spec.cpp:58 */
  }
  if (ws->fftw_plan) g_fftw.exec(ws->fftw_plan); /* spec.cpp:60 on the FFTW-API library */
  else fft_exec(ws->pl, ws->in, ws->out, ws->scratch); /* spec.cpp:60 */
  /* spec.cpp:61-65: bins 0..N/2-1, double sqrt, / N, narrowed to float */
  for (int k = 0; k < N / 2; ++k) {
    const double re = ws->out[2 * k], im = ws->out[2 * k + 1];
    ret[k] = (float)(sqrt(re * re + im * im) / N);
  }
}

int mxo_spec_frame(const float *wav, int n, int N, int start, int end, float *out) {
  frame_ws ws;
  if (ws_init(&ws, N)) return -1;
  spec_frame_ws(&ws, wav, n, start, end, out);
  ws_free(&ws);
  return 0;
}

/* Build-defined pitch pick, SURVEY.md §8 a-6. */
int mxo_pitch_pick(const float *mags, int nbins, int kmin, int kmax, int32_t *bin, float *mag) {
  if (kmin < 0) kmin = 0;
  if (kmax > nbins - 1) kmax = nbins - 1;
  int best = -1;
  float bm = 0.f;
  for (int k = kmin; k <= kmax; ++k) {
    if (best < 0 || mags[k] > bm) { best = k; bm = mags[k]; }
  }
  *bin = best;
  *mag = bm;
  return best < 0 ? -1 : 0;
}

/* Default band = the default view's notes 24..84 (app.hpp:45-46);
 * f(note) = 55*2^((note-24)/12) Hz, bin = f*N/sr (app.cpp:499-516). */
void mxo_pitch_band(int N, int sampleRate, int *kmin, int *kmax) {
  const double flo = 55.0, fhi = 55.0 * 32.0;
  *kmin = (int)ceil(flo * N / sampleRate);
  *kmax = (int)floor(fhi * N / sampleRate);
}

typedef struct hop_job {
  const float *wav; int n, N, hop; long first, count; int kmin, kmax;
  float *mags; int32_t *pbin; float *pmag; int rc; int use_fftw_api;
  pthread_barrier_t *start; /* timed runs: every thread has its plan and buffers before any starts on its frames */
  int pick;                 /* timed runs: pick the pitch although nothing is stored */
  long sink;                /* ... and keep the pick observable */
  double t_begin, t_end;    /* CLOCK_MONOTONIC seconds around this thread's frame loop */
} hop_job;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *hop_worker(void *arg) {
  hop_job *jb = (hop_job *)arg;
  frame_ws ws;
  const int bad = ws_init_p(&ws, jb->N, jb->use_fftw_api);
  float *row = (float *)malloc(sizeof(float) * (size_t)(jb->N / 2));
  if (jb->start) pthread_barrier_wait(jb->start);
  if (bad) { free(row); jb->rc = -1; return NULL; }
  jb->t_begin = now_s();
  for (long f = 0; f < jb->count; ++f) {
    const long h = jb->first + f;
    const int start = (int)(h * jb->hop), end = (int)((h + 1) * jb->hop);
    float *dst = jb->mags ? jb->mags + (size_t)f * (size_t)(jb->N / 2) : row;
    spec_frame_ws(&ws, jb->wav, jb->n, start, end, dst);
    if (jb->pbin || jb->pmag || jb->pick) {
      int32_t b; float m;
      mxo_pitch_pick(dst, jb->N / 2, jb->kmin, jb->kmax, &b, &m);
      if (jb->pbin) jb->pbin[f] = b;
      if (jb->pmag) jb->pmag[f] = m;
      jb->sink += b;
    }
  }
  jb->t_end = now_s();
  free(row);
  ws_free(&ws);
  jb->rc = 0;
  return NULL;
}

int mxo_stft_hop(const float *wav, int n, int N, int hop, long first_frame, long count,
                 int kmin, int kmax, float *mags, int32_t *pitch_bin, float *pitch_mag,
                 int nthreads) {
  return mxo_stft_hop_p(wav, n, N, hop, first_frame, count, kmin, kmax, mags, pitch_bin, pitch_mag, nthreads, 0);
}

int mxo_spec_frame_fftw_api(const float *wav, int n, int N, int start, int end, float *out) {
  frame_ws ws;
  const int rc = ws_init_p(&ws, N, 1);
  if (rc) { ws_free(&ws); return rc; }
  spec_frame_ws(&ws, wav, n, start, end, out);
  ws_free(&ws);
  return 0;
}

static int stft_hop_impl(const float *wav, int n, int N, int hop, long first_frame, long count,
                         int kmin, int kmax, float *mags, int32_t *pitch_bin, float *pitch_mag,
                         int nthreads, int use_fftw_api, double *loop_seconds) {
  if (!is_pow2(N) || hop <= 0 || count < 0) return -1;
  if (use_fftw_api) {
    if (!fftw_api_load()) return -2;
    /* the first plan of this size is made (and run once) here, before the worker threads make theirs */
    frame_ws warm;
    if (ws_init_p(&warm, N, 1) == 0) g_fftw.exec(warm.fftw_plan);
    ws_free(&warm);
  }
  if (nthreads < 1) nthreads = 1;
  if (nthreads > count) nthreads = count > 0 ? (int)count : 1;
  plan_for(N); /* build the twiddle table before threads race for it */
  hop_job *jobs = (hop_job *)calloc((size_t)nthreads, sizeof(hop_job));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  pthread_barrier_t start;
  const int timed = loop_seconds != NULL && nthreads > 1;
  if (timed) pthread_barrier_init(&start, NULL, (unsigned)nthreads);
  long done = 0;
  for (int t = 0; t < nthreads; ++t) {
    const long share = (count - done) / (nthreads - t);
    hop_job *jb = &jobs[t];
    jb->wav = wav; jb->n = n; jb->N = N; jb->hop = hop;
    jb->first = first_frame + done; jb->count = share;
    jb->kmin = kmin; jb->kmax = kmax; jb->use_fftw_api = use_fftw_api;
    jb->mags = mags ? mags + (size_t)done * (size_t)(N / 2) : NULL;
    jb->pbin = pitch_bin ? pitch_bin + done : NULL;
    jb->pmag = pitch_mag ? pitch_mag + done : NULL;
    jb->start = timed ? &start : NULL;
    jb->pick = loop_seconds != NULL;
    done += share;
    if (nthreads == 1) hop_worker(jb);
    else pthread_create(&th[t], NULL, hop_worker, jb);
  }
  int rc = 0;
  for (int t = 0; t < nthreads; ++t) {
    if (nthreads > 1) pthread_join(th[t], NULL);
    if (jobs[t].rc) rc = -1;
  }
  if (loop_seconds && rc == 0) { /* first thread into its frames .. last thread out */
    double b = jobs[0].t_begin, e = jobs[0].t_end;
    for (int t = 1; t < nthreads; ++t) {
      if (jobs[t].t_begin < b) b = jobs[t].t_begin;
      if (jobs[t].t_end > e) e = jobs[t].t_end;
    }
    *loop_seconds = e - b;
  }
  if (timed) pthread_barrier_destroy(&start);
  free(jobs);
  free(th);
  return rc;
}

int mxo_stft_hop_p(const float *wav, int n, int N, int hop, long first_frame, long count,
                   int kmin, int kmax, float *mags, int32_t *pitch_bin, float *pitch_mag,
                   int nthreads, int use_fftw_api) {
  return stft_hop_impl(wav, n, N, hop, first_frame, count, kmin, kmax, mags, pitch_bin, pitch_mag, nthreads,
                       use_fftw_api, NULL);
}

int mxo_stft_hop_timed(const float *wav, int n, int N, int hop, long first_frame, long count,
                       int kmin, int kmax, int nthreads, int use_fftw_api, double *loop_seconds) {
  return stft_hop_impl(wav, n, N, hop, first_frame, count, kmin, kmax, NULL, NULL, NULL, nthreads, use_fftw_api,
                       loop_seconds);
}

/* ======================================================================
 * SpecCache::populateTex colormap — spec-cache.cpp:77-96
 * ====================================================================== */
void mxo_colormap(const float *s, int nbins, float k, unsigned char *rgb) {
  for (int i = 0; i < nbins; ++i) {
    float tmp = s[i] * k; /* std::clamp(s[i]*k, 0.f, 255.f), spec-cache.cpp:79 */
    if (tmp < 0.f) tmp = 0.f; else if (255.f < tmp) tmp = 255.f;
    unsigned char *d = rgb + 3 * i;
    if (tmp < 255 / 3) { /* int 85 */
      d[0] = (unsigned char)tmp; d[1] = 0; d[2] = 0;
    } else if (tmp < 2 * 255 / 3) { /* int 170 */
      const double a = (tmp - 255 / 3) / (255 / 3) * 3.141592 / 2; /* float sub/div, then double */
      d[0] = (unsigned char)(tmp * cos(a));
      d[1] = (unsigned char)(tmp * sin(a));
      d[2] = 0;
    } else {
      const unsigned char l_k = (unsigned char)((tmp - 2 * 255 / 3) * 3);
      d[0] = l_k; d[1] = (unsigned char)tmp; d[2] = l_k;
    }
  }
}

/* ======================================================================
 * Time maps — app.cpp:1020-1122.  The reference memoises results in three
 * unordered_maps keyed by int (app.hpp:61-63); memo!=0 reproduces that.
 * ====================================================================== */

typedef struct imap { /* int -> 8 bytes, open addressing */
  int cap, used;
  int *keys;
  unsigned char *occ;
  double *vals;
} imap;

static void imap_init(imap *m) { m->cap = 0; m->used = 0; m->keys = NULL; m->occ = NULL; m->vals = NULL; }
static void imap_free(imap *m) { free(m->keys); free(m->occ); free(m->vals); imap_init(m); }
static unsigned imap_hash(int k) { unsigned x = (unsigned)k; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
static int imap_find(const imap *m, int key, double *val) {
  if (!m->cap) return 0;
  unsigned i = imap_hash(key) & (unsigned)(m->cap - 1);
  while (m->occ[i]) {
    if (m->keys[i] == key) { *val = m->vals[i]; return 1; }
    i = (i + 1) & (unsigned)(m->cap - 1);
  }
  return 0;
}
static void imap_put(imap *m, int key, double val);
static void imap_grow(imap *m) {
  imap old = *m;
  m->cap = old.cap ? old.cap * 2 : 1024;
  m->used = 0;
  m->keys = (int *)calloc((size_t)m->cap, sizeof(int));
  m->occ = (unsigned char *)calloc((size_t)m->cap, 1);
  m->vals = (double *)calloc((size_t)m->cap, sizeof(double));
  for (int i = 0; i < old.cap; ++i)
    if (old.occ[i]) imap_put(m, old.keys[i], old.vals[i]);
  free(old.keys); free(old.occ); free(old.vals);
}
static void imap_put(imap *m, int key, double val) {
  if ((m->used + 1) * 2 > m->cap) imap_grow(m);
  unsigned i = imap_hash(key) & (unsigned)(m->cap - 1);
  while (m->occ[i]) {
    if (m->keys[i] == key) { m->vals[i] = val; return; } /* operator[] = overwrite */
    i = (i + 1) & (unsigned)(m->cap - 1);
  }
  m->occ[i] = 1; m->keys[i] = key; m->vals[i] = val; m->used++;
}

struct mxo_timemap {
  mxo_marker *markers;
  int nmarkers;
  int sampleRate;
  long nsamples;
  int memo;
  imap s2t, t2s, t2pb;
};

mxo_timemap *mxo_timemap_new(const mxo_marker *markers, int nmarkers, int sampleRate,
                             long nsamples, int memo) {
  mxo_timemap *tm = (mxo_timemap *)calloc(1, sizeof(*tm));
  tm->markers = (mxo_marker *)malloc(sizeof(mxo_marker) * (size_t)(nmarkers > 0 ? nmarkers : 1));
  if (nmarkers > 0) memcpy(tm->markers, markers, sizeof(mxo_marker) * (size_t)nmarkers);
  tm->nmarkers = nmarkers;
  tm->sampleRate = sampleRate;
  tm->nsamples = nsamples;
  tm->memo = memo;
  imap_init(&tm->s2t); imap_init(&tm->t2s); imap_init(&tm->t2pb);
  return tm;
}
void mxo_timemap_free(mxo_timemap *tm) {
  if (!tm) return;
  imap_free(&tm->s2t); imap_free(&tm->t2s); imap_free(&tm->t2pb);
  free(tm->markers);
  free(tm);
}

/* app.cpp:1020-1050 */
double mxo_sample2time(mxo_timemap *tm, int val) {
  const int sampleRate = tm->sampleRate;
  if (val <= 0) return 1. * val / sampleRate;
  double hit;
  if (tm->memo && imap_find(&tm->s2t, val, &hit)) return hit;
  int prevSample = 0;
  double prevTime = 0.0;
  for (int m = 0; m < tm->nmarkers; ++m) {
    const mxo_marker *mk = &tm->markers[m];
    const double rightTime = prevTime + 1.0 * (mk->sample - prevSample) / sampleRate + mk->dTime;
    if (val > prevSample && val <= mk->sample) {
      const double ret = prevTime + (val - prevSample) * (rightTime - prevTime) / (mk->sample - prevSample);
      if (tm->memo) imap_put(&tm->s2t, val, ret);
      return ret;
    }
    prevSample = mk->sample;
    prevTime = rightTime;
  }
  const double ret = prevTime + 1. * (val - prevSample) / sampleRate;
  if (tm->memo) imap_put(&tm->s2t, val, ret);
  return ret;
}

/* app.cpp:1052-1082 */
int mxo_time2sample(mxo_timemap *tm, double val) {
  const int sampleRate = tm->sampleRate;
  if (val <= 0) return (int)(val * sampleRate);
  const int key = (int)(val * sampleRate);
  double hit;
  if (tm->memo && imap_find(&tm->t2s, key, &hit)) return (int)hit;
  int prevSample = 0;
  double prevTime = 0.0;
  for (int m = 0; m < tm->nmarkers; ++m) {
    const mxo_marker *mk = &tm->markers[m];
    const double rightTime = prevTime + 1.0 * (mk->sample - prevSample) / sampleRate + mk->dTime;
    if (val > prevTime && val <= rightTime) {
      const int ret = (int)(prevSample + (val - prevTime) * (mk->sample - prevSample) / (rightTime - prevTime));
      if (tm->memo) imap_put(&tm->t2s, key, (double)ret);
      return ret;
    }
    prevSample = mk->sample;
    prevTime = rightTime;
  }
  const int ret = (int)(prevSample + (val - prevTime) * sampleRate);
  if (tm->memo) imap_put(&tm->t2s, key, (double)ret);
  return ret;
}

/* app.cpp:1084-1087 */
double mxo_duration(mxo_timemap *tm) { return mxo_sample2time(tm, (int)(tm->nsamples - 1)); }

/* app.cpp:1089-1122 */
float mxo_time2pitchbend(mxo_timemap *tm, double val) {
  const int sampleRate = tm->sampleRate;
  if (val <= 0) return 0;
  const int key = (int)(val * sampleRate);
  double hit;
  if (tm->memo && imap_find(&tm->t2pb, key, &hit)) return (float)hit;
  int prevSample = 0;
  double prevTime = 0.0;
  double prevPitchBend = 0.0;
  for (int m = 0; m < tm->nmarkers; ++m) {
    const mxo_marker *mk = &tm->markers[m];
    const double rightTime = prevTime + 1.0 * (mk->sample - prevSample) / sampleRate + mk->dTime;
    if (val > prevTime && val <= rightTime) {
      const float ret = (float)(prevPitchBend + (val - prevTime) * (mk->pitchBend - prevPitchBend) / (rightTime - prevTime));
      if (tm->memo) imap_put(&tm->t2pb, key, (double)ret);
      return ret;
    }
    prevSample = mk->sample;
    prevTime = rightTime;
    prevPitchBend = mk->pitchBend;
  }
  if (val > mxo_duration(tm)) return 0;
  const float ret = (float)(prevPitchBend + (val - prevTime) * (0 - prevPitchBend) / (mxo_duration(tm) - prevTime));
  if (tm->memo) imap_put(&tm->t2pb, key, (double)ret);
  return ret;
}

/* spec-cache.cpp:12 (key), :63-65 (range) */
void mxo_column_range(mxo_timemap *tm, double time, int width, double rangeTime, int *key,
                      int *start, int *end) {
  const int k = (int)(time * width / rangeTime);
  const double st = k * rangeTime / width;
  const double pixelSize = rangeTime / width;
  *key = k;
  *start = mxo_time2sample(tm, st);
  *end = mxo_time2sample(tm, st + pixelSize);
}

/* ======================================================================
 * Grain segmentation — App::preproc, app.cpp:153-235
 * ====================================================================== */
#define PREFERRED_GRAIN 1500 /* app.cpp:19 */

static int zero_crossing(const float *wav, long n, int idx, int lookAround) {
  /* app.cpp:167-181 (lookAround 7) and :202-216 (lookAround 3) */
  if (idx < lookAround) return 0;
  if (idx >= (int)(n - lookAround - 1)) return 0;
  for (int j = 0; j < lookAround; ++j) {
    if (wav[idx - j] >= 0) return 0;
    if (wav[idx + 1 + j] < 0) return 0;
  }
  return 1;
}

long mxo_grains(const float *wav, long n, int **starts_out, int **lens_out) {
  long cap = n / 700 + 16, cnt = 0;
  int *starts = (int *)malloc(sizeof(int) * (size_t)cap);
  int *lens = (int *)malloc(sizeof(int) * (size_t)cap);
  /* app.cpp:161 compares against (int)(size_t)(n-1501): for n<1501 that wraps in
   * the reference; such inputs are outside the tested domain. */
  if (n >= PREFERRED_GRAIN + 1) {
    int start = 0;
    while (start < (int)(n - PREFERRED_GRAIN - 1)) {
      int found = 0;
      for (int i = 0; i < PREFERRED_GRAIN; ++i) {
        const int idx = start + PREFERRED_GRAIN + (i % 2 == 0 ? i / 2 : -i / 2); /* app.cpp:166 */
        if (zero_crossing(wav, n, idx, 7)) {
          if (cnt == cap) { cap *= 2; starts = realloc(starts, sizeof(int) * (size_t)cap); lens = realloc(lens, sizeof(int) * (size_t)cap); }
          starts[cnt] = start; lens[cnt] = idx - start; ++cnt;
          start = idx;
          found = 1;
          break;
        }
      }
      if (!found) {
        for (int i = start + PREFERRED_GRAIN + PREFERRED_GRAIN / 2; i < (int)(n - 1); ++i) { /* app.cpp:198-200 */
          if (zero_crossing(wav, n, i, 3)) {
            if (cnt == cap) { cap *= 2; starts = realloc(starts, sizeof(int) * (size_t)cap); lens = realloc(lens, sizeof(int) * (size_t)cap); }
            starts[cnt] = start; lens[cnt] = i - start; ++cnt;
            start = i;
            found = 1;
            break;
          }
        }
        if (!found) break;
      }
    }
  }
  *starts_out = starts;
  *lens_out = lens;
  return cnt;
}

void mxo_free(void *p) { free(p); }

/* ======================================================================
 * App::process / App::exportWav — app.cpp:294-345, 1194-1215
 * ====================================================================== */

/* std::map::lower_bound over the grain keys (app.cpp:300, :324) */
static long grain_lower_bound(const int *starts, long cnt, int sample) {
  long lo = 0, hi = cnt;
  while (lo < hi) {
    const long mid = lo + (hi - lo) / 2;
    if (starts[mid] < sample) lo = mid + 1; else hi = mid;
  }
  return lo;
}

typedef struct fvec { float *p; long n, cap; } fvec;
static void fvec_push(fvec *v, float x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1 << 16; v->p = (float *)realloc(v->p, sizeof(float) * (size_t)v->cap); }
  v->p[v->n++] = x;
}

/* need < 0: App::exportWav's loop (app.cpp:1200-1207), cursor from 0 until process() returns 0.
 * need >= 0: App::playback's refill loop (app.cpp:272-274): tmpCursor = cursor0;
 *            while (restWav.size() < need) tmpCursor += process(tmpCursor, restWav);  (restWav empty at entry) */
static int run_chain(const float *wav, long n, int sampleRate, const mxo_marker *markers,
                     int nmarkers, int memo, double cursor0, long need, mxo_export *out) {
  int *gs = NULL, *gl = NULL;
  const long ng = mxo_grains(wav, n, &gs, &gl);
  mxo_timemap *tm = mxo_timemap_new(markers, nmarkers, sampleRate, n, memo);
  fvec pcm = {NULL, 0, 0};
  long scap = ng + 16, ns = 0;
  mxo_step *steps = (mxo_step *)malloc(sizeof(mxo_step) * (size_t)scap);
  const float bias = 0.f; /* app.hpp:66, never assigned */

  double cursor = cursor0; /* app.cpp:1201 / :272 */
  for (;;) {
    if (need >= 0 && pcm.n >= need) break; /* app.cpp:273 */
    /* ---- App::process(cursor, pcm), app.cpp:294-345 ---- */
    const float pitchBend = mxo_time2pitchbend(tm, cursor);
    const float rate = powf(2, pitchBend / 12);
    const long g1 = grain_lower_bound(gs, ng, mxo_time2sample(tm, cursor));
    if (g1 == ng) { /* app.cpp:303-309 */
      for (int i = 0; i < PREFERRED_GRAIN; ++i) fvec_push(&pcm, 0.f);
      if (need >= 0) continue; /* playback: returns 0, tmpCursor stays, the while loop asks again */
      break; /* returns 0 -> exportWav's dt<=0 break, app.cpp:1204 */
    }
    const float *grain = wav + gs[g1];
    const size_t L = (size_t)gl[g1];
    int sz = 0;
    for (int i = 0;; ++i) { /* app.cpp:313-322 */
      double idxF;
      modf((double)(i * rate + bias), &idxF);
      const size_t idx = (size_t)idxF;
      if (idx >= L) break;
      ++sz;
    }
    float next = 0.f;
    {
      const long g2 = grain_lower_bound(gs, ng, mxo_time2sample(tm, cursor + 1. * sz / sampleRate));
      if (g2 != ng) next = wav[gs[g2]]; /* .front(), app.cpp:328 */
    }
    if (ns == scap) { scap *= 2; steps = (mxo_step *)realloc(steps, sizeof(mxo_step) * (size_t)scap); }
    mxo_step *st = &steps[ns++];
    st->cursor = cursor; st->grain_start = gs[g1]; st->grain_len = gl[g1];
    st->rate = rate; st->next_first = next; st->out_offset = pcm.n;

    int sz2 = 0;
    for (int i = 0;; ++i) { /* app.cpp:332-343 */
      float idxF;
      const float curBias = modff(i * rate + bias, &idxF);
      const size_t idx = (size_t)idxF;
      if (idx >= L) break;
      fvec_push(&pcm, (1.f - curBias) * grain[idx] + curBias * (idx + 1 < L ? grain[idx + 1] : next));
      ++sz2;
    }
    st->sz = sz2;
    const double dt = 1. * sz2 / sampleRate; /* app.cpp:344 */
    if (need < 0 && dt <= 0.) break; /* app.cpp:1204 */
    cursor += dt; /* app.cpp:1206 / :274 */
  }

  out->cursor_end = cursor;
  out->nsteps = ns;
  out->steps = steps;
  out->nsamples = pcm.n;
  out->pcm = pcm.p;
  mxo_timemap_free(tm);
  free(gs);
  free(gl);
  return 0;
}

int mxo_export_run(const float *wav, long n, int sampleRate, const mxo_marker *markers,
                   int nmarkers, int memo, mxo_export *out) {
  return run_chain(wav, n, sampleRate, markers, nmarkers, memo, 0., -1, out);
}

int mxo_playback_fill(const float *wav, long n, int sampleRate, const mxo_marker *markers,
                      int nmarkers, int memo, double cursor0, long need, mxo_export *out) {
  if (need < 0) return -1;
  return run_chain(wav, n, sampleRate, markers, nmarkers, memo, cursor0, need, out);
}

void mxo_export_free(mxo_export *e) {
  free(e->steps);
  free(e->pcm);
  e->steps = NULL;
  e->pcm = NULL;
}

/* app.cpp:1209-1212: static_cast<int16_t>(pcm[i] * 32767.) — double multiply,
 * truncation toward zero. */
void mxo_pcm_to_i16(const float *pcm, long m, int16_t *out) {
  for (long i = 0; i < m; ++i) out[i] = (int16_t)(pcm[i] * 32767.);
}

/* ======================================================================
 * saveWav — save-wav.cpp:17-48 (writeWord :8-13)
 * ====================================================================== */
static void put_le(unsigned char *b, uint64_t v, int size) {
  for (int i = 0; i < size; ++i, v >>= 8) b[i] = (unsigned char)(v & 0xFF);
}

long mxo_wav_bytes(const int16_t *pcm, long m, int sampleRate, unsigned char *buf) {
  memcpy(buf, "RIFF----WAVEfmt ", 16);                          /* :22 */
  put_le(buf + 16, 16, 4);                                      /* :23 */
  put_le(buf + 20, 1, 2);                                       /* :24 */
  put_le(buf + 22, 1, 2);                                       /* :25 */
  put_le(buf + 24, (uint64_t)(int64_t)sampleRate, 4);           /* :26 */
  put_le(buf + 28, (uint64_t)(int64_t)((sampleRate * 16 * 1) / 8), 4); /* :27 */
  put_le(buf + 32, 2, 2);                                       /* :28 */
  put_le(buf + 34, 16, 2);                                      /* :29 */
  const size_t dataChunkPos = 36;                               /* :32 */
  memcpy(buf + 36, "data----", 8);                              /* :33 */
  for (long i = 0; i < m; ++i) put_le(buf + 44 + 2 * i, (uint64_t)(int64_t)pcm[i], 2); /* :35-36 */
  const size_t fileLength = 44 + 2 * (size_t)m;                 /* :39 */
  long len = (long)fileLength;
  /* :42-43 — writeWord(f, size_t) with the default size=sizeof(Word)=8:
   * eight bytes land at offset 40, clobbering PCM samples 0 and 1 (and
   * extending a file shorter than 48 bytes). */
  put_le(buf + dataChunkPos + 4, (uint64_t)(fileLength - dataChunkPos + 8), 8);
  if (len < 48) len = 48;
  put_le(buf + 4, (uint64_t)(fileLength - 8), 4);               /* :46-47 */
  return len;
}

int mxo_save_wav(const char *path, const int16_t *pcm, long m, int sampleRate) {
  unsigned char *buf = (unsigned char *)malloc(48 + 2 * (size_t)m);
  if (!buf) return -1;
  const long len = mxo_wav_bytes(pcm, m, sampleRate, buf);
  FILE *f = fopen(path, "wb");
  if (!f) { free(buf); return -1; }
  const size_t w = fwrite(buf, 1, (size_t)len, f);
  fclose(f);
  free(buf);
  return w == (size_t)len ? 0 : -1;
}

/* ======================================================================
 * Waveform min/max pyramid — App::calcPicks / getMinMaxFromRange, app.cpp:347-426
 * ====================================================================== */
int mxo_calc_picks(const float *wav, long n, float *out, long *counts, int max_levels) {
  int lvl = 0;
  size_t size = (size_t)n;
  if (size <= ((size_t)1 << (lvl + 1))) return 0; /* app.cpp:352 */
  long cnt = (long)(size / ((size_t)1 << (lvl + 1)));
  for (long i = 0; i < cnt; ++i) { /* app.cpp:356-361 */
    const float a = wav[i * 2], b = wav[i * 2 + 1];
    out[2 * i] = b < a ? b : a;     /* std::min(a, b) */
    out[2 * i + 1] = a < b ? b : a; /* std::max(a, b) */
  }
  counts[0] = cnt;
  const float *prev = out;
  float *cur = out + 2 * cnt;
  for (;;) { /* app.cpp:363-376 */
    ++lvl;
    if (size <= ((size_t)1 << (lvl + 1)) || lvl >= max_levels) break;
    cnt = (long)(size / ((size_t)1 << (lvl + 1)));
    for (long i = 0; i < cnt; ++i) {
      const float m0 = prev[2 * (2 * i)], m1 = prev[2 * (2 * i + 1)];
      const float x0 = prev[2 * (2 * i) + 1], x1 = prev[2 * (2 * i + 1) + 1];
      cur[2 * i] = m1 < m0 ? m1 : m0;
      cur[2 * i + 1] = x0 < x1 ? x1 : x0;
    }
    counts[lvl] = cnt;
    prev = cur;
    cur += 2 * cnt;
  }
  return lvl;
}

static void minmax_rec(const float *wav, long n, const float *const *lv, const long *counts, int nlevels,
                       int start, int end, float *mn, float *mx) {
  if (start >= end) { /* app.cpp:382-387 */
    if (start >= 0 && start < (int)n) { *mn = wav[start]; *mx = wav[start]; } else { *mn = 0.f; *mx = 0.f; }
    return;
  }
  if (start < 0 || end < 0) { *mn = 0.f; *mx = 0.f; return; }
  if (start >= (int)n || end >= (int)n) { *mn = 0.f; *mx = 0.f; return; }
  if (end - start == 1) { *mn = wav[start]; *mx = wav[start]; return; }
  const size_t lvl = (size_t)log2((double)(end - start)); /* app.cpp:399 */
  const int lvlStart = start / (1 << lvl);
  float a = 0.f, b = 0.f; /* app.cpp:402-408 */
  if (!(lvl - 1 >= (size_t)nlevels) && !(lvlStart >= (int)counts[lvl - 1])) {
    a = lv[lvl - 1][2 * lvlStart];
    b = lv[lvl - 1][2 * lvlStart + 1];
  }
  const int leftEnd = lvlStart * (1 << lvl);
  if (leftEnd >= start) { /* app.cpp:411-416 */
    float l0, l1;
    minmax_rec(wav, n, lv, counts, nlevels, start, leftEnd, &l0, &l1);
    a = l0 < a ? l0 : a;
    b = b < l1 ? l1 : b;
  }
  const int rightStart = (lvlStart + 1) * (1 << lvl);
  if (rightStart < end) { /* app.cpp:419-424 */
    float r0, r1;
    minmax_rec(wav, n, lv, counts, nlevels, rightStart, end, &r0, &r1);
    a = r0 < a ? r0 : a;
    b = b < r1 ? r1 : b;
  }
  *mn = a;
  *mx = b;
}

void mxo_minmax_range(const float *wav, long n, const float *picks, const long *counts, int nlevels,
                      int start, int end, float *mn, float *mx) {
  const float *lv[64];
  const float *p = picks;
  for (int l = 0; l < nlevels && l < 64; ++l) { lv[l] = p; p += 2 * counts[l]; }
  minmax_rec(wav, n, lv, counts, nlevels, start, end, mn, mx);
}

/* ======================================================================
 * Synthetic input — SURVEY.md §8(d): closed-form linear sine sweep.
 * ====================================================================== */
void mxo_sweep(float *out, long n, int sampleRate, double f0, double f1, double amp) {
  const double two_pi = 6.283185307179586476925286766559;
  const double T = (double)n / sampleRate;
  for (long i = 0; i < n; ++i) {
    const double t = (double)i / sampleRate;
    out[i] = (float)(amp * sin(two_pi * (f0 * t + (f1 - f0) * t * t / (2 * T))));
  }
}
