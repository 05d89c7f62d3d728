// grain_chain.hip — the grain chain of App::preproc (reference app.cpp:153-235) on the device.
//
// The reference walks   start -> idx(start)   from 0: idx = the zero crossing (look-around 7, app.cpp:167-181) nearest to
// start + 1500 within +-749, the later one on ties (candidate order app.cpp:164-166); if there is none, the first
// look-around-3 crossing at or after start + 2250 (app.cpp:198-228); it stops when start >= n - 1501 or nothing is found.
// A serial chain of ~n/1500 steps — but idx() is a pure function of `start`, and every start but 0 is itself a
// crossing.  So:
//   1. the two predicate bitmaps (zc_kernel, resynth_kernels.hip: bit-parallel, one read of the audio);
//   2. every look-around-3 bit (a superset of the look-around-7 bits) is a node, numbered by its rank (popcount scan);
//   3. succ(node) = rank of idx(position) for ALL nodes at once (one thread per bitmap word; the +-749 window is
//      24 words, the fallback search skips empty words through the rank table);
//   4. binary lifting J[l] = J[l-1] o J[l-1]  (succ is monotone, so the gathers are nearly coalesced);
//   5. one thread measures the chain from the head (start 0), then grain g = the (g)-fold successor of the head,
//      found by the bits of g: positions, lengths and first samples of all grains in parallel.
// Exactly the reference's chain for any input (no speculation), and the host only receives the grain table.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace mx {

namespace {

constexpr uint32_t kEnd = 0xffffffffu;
constexpr int kPref = 1500;     // preferredGrainSize (app.cpp:19)
constexpr int kReach = 749;     // candidates start + 1500 +- 749 (app.cpp:164-166)
constexpr int kScanWords = 1024;  // bitmap words per scan block

struct ChainArgs {
  const float *audio;   // padded image
  int64_t n;
  int64_t nwords;
  const uint64_t *zc7, *zc3;
  uint32_t *wbase;      // [nwords]   rank of the word's first bit inside its scan block
  uint32_t *bsum;       // [nblocks]  exclusive scan of the scan blocks' totals; bsum[nblocks] = number of nodes
  uint32_t *pos;        // [nodes]    sample index of node r
  uint32_t *J;          // [levels][nodes] successor tables
  int levels;
  uint32_t nodes_cap;
  uint32_t *head;       // [4] {succ of the head (start 0), grain count, node count, unused}
  int32_t *starts, *lens;  // [cap] outputs
  float *firsts;
  uint32_t out_cap;
};

__device__ __forceinline__ uint32_t rank_of(const ChainArgs &a, int64_t p) {  // rank of set bit p of zc3
  const int64_t w = p >> 6;
  const uint64_t below = a.zc3[w] & ((1ull << (p & 63)) - 1ull);
  return a.bsum[w / kScanWords] + a.wbase[w] + (uint32_t)__popcll(below);
}
__device__ __forceinline__ uint32_t word_rank(const ChainArgs &a, int64_t w) {  // zc3 bits in words [0, w)
  return w >= a.nwords ? a.bsum[(a.nwords + kScanWords - 1) / kScanWords] : a.bsum[w / kScanWords] + a.wbase[w];
}

// idx(start) of the reference, or -1 (chain ends).  start is 0 or a look-around-3 bit.
__device__ int64_t next_start(const ChainArgs &a, int64_t start) {
  if (!(start < a.n - kPref - 1)) return -1;  // app.cpp:161
  const int64_t c = start + kPref;
  int64_t lo = c - kReach, hi = c + kReach;
  if (lo < 0) lo = 0;
  if (hi > a.n - 1) hi = a.n - 1;
  // nearest set bit of zc7 to c in [lo, hi]: upwards from c and downwards from c, word by word
  int64_t up = -1, dn = -1;
  {
    int64_t w = c >> 6;
    uint64_t cur = a.zc7[w] & (~0ull << (c & 63));
    const int64_t wend = hi >> 6;
    for (;;) {
      if (cur) {
        const int64_t p = (w << 6) + __ffsll((long long)cur) - 1;
        up = p <= hi ? p : -1;
        break;
      }
      if (++w > wend || w >= a.nwords) break;
      cur = a.zc7[w];
    }
  }
  {
    int64_t w = c >> 6;
    const int sh = 63 - (int)(c & 63);
    uint64_t cur = a.zc7[w] & (~0ull >> sh);  // (c <= n - 2: the word exists)
    const int64_t wbeg = lo >> 6;
    for (;;) {
      if (cur) {
        const int64_t p = (w << 6) + 63 - __clzll((long long)cur);
        dn = p >= lo ? p : -1;
        break;
      }
      if (--w < wbeg) break;
      cur = a.zc7[w];
    }
  }
  if (up >= 0 && dn >= 0) return (up - c <= c - dn) ? up : dn;  // nearest first, the later index on ties
  if (up >= 0) return up;
  if (dn >= 0) return dn;
  // fallback: the first look-around-3 bit in [start + 2250, n - 2]
  const int64_t f0 = start + kPref + kPref / 2, f1 = a.n - 2;
  if (f0 > f1) return -1;
  int64_t w = f0 >> 6;
  uint64_t cur = a.zc3[w] & (~0ull << (f0 & 63));
  if (!cur) {
    // skip the empty words: first word > w whose rank base exceeds the rank at the end of w (binary search)
    const uint32_t r0 = word_rank(a, w + 1);
    int64_t lo_w = w + 1, hi_w = a.nwords;  // find the smallest x in (w, nwords] with word_rank(x) > r0; the bit is in word x-1
    if (word_rank(a, hi_w) <= r0) return -1;
    while (lo_w < hi_w) {
      const int64_t mid = (lo_w + hi_w) >> 1;
      if (word_rank(a, mid) > r0) hi_w = mid;
      else lo_w = mid + 1;
    }
    w = lo_w - 1;
    cur = a.zc3[w];
  }
  const int64_t p = (w << 6) + __ffsll((long long)cur) - 1;
  return p <= f1 ? p : -1;
}

// ---- 2: ranks -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chain_scan_a(const ChainArgs a) {
  __shared__ uint32_t part[256];
  const int64_t w0 = (int64_t)blockIdx.x * kScanWords + threadIdx.x * 4;
  uint32_t c[4], s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c[i] = (w0 + i < a.nwords) ? (uint32_t)__popcll(a.zc3[w0 + i]) : 0u;
    s += c[i];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {  // inclusive scan of the 256 partial sums
    const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - s;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (w0 + i < a.nwords) a.wbase[w0 + i] = run;
    run += c[i];
  }
  if (threadIdx.x == 255) a.bsum[blockIdx.x] = part[255];
}
__global__ __launch_bounds__(1024) void chain_scan_b(const ChainArgs a, int nblocks) {  // exclusive scan of bsum, one workgroup
  __shared__ uint32_t part[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < nblocks ? a.bsum[i] : 0u;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const uint32_t u = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
      __syncthreads();
      part[threadIdx.x] += u;
      __syncthreads();
    }
    if (i < nblocks) a.bsum[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.bsum[nblocks] = carry;  // number of nodes
    a.head[2] = carry;
  }
}

// ---- 3: successors ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chain_succ(const ChainArgs a) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w == 0) {  // the head of the chain is start 0 (never a crossing itself: app.cpp:159)
    const int64_t p = next_start(a, 0);
    a.head[0] = p < 0 ? kEnd : rank_of(a, p);
  }
  if (w >= a.nwords) return;
  uint64_t bits = a.zc3[w];
  if (!bits) return;
  uint32_t r = a.bsum[w / kScanWords] + a.wbase[w];
  if (r >= a.nodes_cap) return;  // (guarded by the host: the tables are sized from an upper bound)
  while (bits) {
    const int b = __ffsll((long long)bits) - 1;
    bits &= bits - 1;
    const int64_t start = (w << 6) + b;
    const int64_t p = next_start(a, start);
    if (r < a.nodes_cap) {
      a.pos[r] = (uint32_t)start;
      a.J[r] = p < 0 ? kEnd : rank_of(a, p);
    }
    ++r;
  }
}

// ---- 4: lifting ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chain_lift(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t nodes) {
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= nodes) return;
  const uint32_t m = src[v];
  dst[v] = m == kEnd ? kEnd : src[m];
}

// ---- 5: the chain from the head ---------------------------------------------------------------------------------
__global__ void chain_count(const ChainArgs a) {
  // nodes on the chain after the head: first = head[0]; count them by descending the levels
  uint32_t v = a.head[0], cnt = 0;
  const uint32_t nodes = a.head[2];
  if (v != kEnd) {
    cnt = 1;
    for (int l = a.levels - 1; l >= 0; --l) {
      const uint32_t m = a.J[(size_t)l * a.nodes_cap + v];
      if (m != kEnd) {
        v = m;
        cnt += 1u << l;
      }
    }
  }
  (void)nodes;
  a.head[1] = cnt;  // grains = chain nodes after the head: grain g spans [node g, node g+1), node 0 = start 0
}
__global__ __launch_bounds__(256) void chain_expand(const ChainArgs a) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;  // grain g = [P_g, P_{g+1}); P_0 = 0, P_k = (k-1)-fold successor of head[0]
  const uint32_t cnt = a.head[1];
  if (g >= cnt || g >= a.out_cap) return;
  auto node_at = [&](uint32_t k) -> uint32_t {  // position of chain node k >= 1
    uint32_t v = a.head[0];
    uint32_t steps = k - 1;
    for (int l = 0; steps; ++l, steps >>= 1)
      if (steps & 1u) v = a.J[(size_t)l * a.nodes_cap + v];
    return a.pos[v];
  };
  const uint32_t p0 = g == 0 ? 0u : node_at(g);
  const uint32_t p1 = node_at(g + 1);
  a.starts[g] = (int32_t)p0;
  a.lens[g] = (int32_t)(p1 - p0);
  a.firsts[g] = a.audio[MX_AUDIO_PAD + p0];
}

}  // namespace

// ---- host side: two phases around one small read-back (the node count sizes the lifting tables) ---------------
static ChainArgs chain_layout(const float *audio_padded, int64_t n, const uint64_t *zc7, const uint64_t *zc3, void *rank_scratch) {
  ChainArgs a{};
  a.audio = audio_padded;
  a.n = n;
  a.nwords = (n + 63) >> 6;
  const int64_t nblocks = (a.nwords + kScanWords - 1) / kScanWords;
  a.zc7 = zc7;
  a.zc3 = zc3;
  char *p = static_cast<char *>(rank_scratch);
  a.head = reinterpret_cast<uint32_t *>(p);
  a.bsum = reinterpret_cast<uint32_t *>(p + 64);
  a.wbase = reinterpret_cast<uint32_t *>(p + 64 + (((size_t)(nblocks + 1) * 4 + 63) & ~(size_t)63));
  return a;
}

size_t grain_rank_scratch_bytes(int64_t n) {
  const int64_t nwords = (n + 63) >> 6;
  const int64_t nblocks = (nwords + kScanWords - 1) / kScanWords;
  return 64 + (((size_t)(nblocks + 1) * 4 + 63) & ~(size_t)63) + (size_t)nwords * 4 + 64;
}

// Phase 1: ranks of the look-around-3 bits.  rank_scratch: grain_rank_scratch_bytes(n); its first 16 bytes are the
// header {head successor, grain count, NODE COUNT, -} — read word 2 back before phase 2.
hipError_t launch_grain_rank(const float *audio_padded, int64_t n, const uint64_t *zc7, const uint64_t *zc3, void *rank_scratch,
                             hipStream_t s) {
  if ((uint64_t)n >= 0xfffffff0ull) return hipErrorInvalidValue;  // positions are 32-bit, like the reference's int
  const ChainArgs a = chain_layout(audio_padded, n, zc7, zc3, rank_scratch);
  const int64_t nblocks = (a.nwords + kScanWords - 1) / kScanWords;
  hipError_t e = hipMemsetAsync(a.head, 0, 16, s);
  if (e != hipSuccess || a.nwords == 0) return e;
  hipLaunchKernelGGL(chain_scan_a, dim3((unsigned)nblocks), dim3(256), 0, s, a);
  hipLaunchKernelGGL(chain_scan_b, dim3(1), dim3(1024), 0, s, a, (int)nblocks);
  return hipGetLastError();
}

void grain_chain_sizes(int64_t n, uint32_t nodes, int *levels_out, uint32_t *out_cap_out, size_t *bytes_out) {
  const uint64_t out_cap = (uint64_t)n / 751 + 8;  // a grain is at least 751 samples long (app.cpp:164-166)
  int levels = 1;
  while ((1ull << levels) < out_cap + 1) ++levels;
  *levels_out = levels;
  *out_cap_out = (uint32_t)out_cap;
  *bytes_out = (size_t)(nodes + 16) * 4 * (size_t)(levels + 1) + (size_t)out_cap * 12 + 1024;
}

// Phase 2: successors, lifting, expansion.  chain_scratch: grain_chain_sizes(n, nodes).bytes.  The grain count ends up
// in header word 1, the grain table (starts, lens, first samples; out_cap entries each) at the returned pointers.
hipError_t launch_grain_chain(const float *audio_padded, int64_t n, const uint64_t *zc7, const uint64_t *zc3, void *rank_scratch,
                              uint32_t nodes, void *chain_scratch, int32_t **d_starts, int32_t **d_lens, float **d_firsts,
                              hipStream_t s) {
  ChainArgs a = chain_layout(audio_padded, n, zc7, zc3, rank_scratch);
  int levels;
  uint32_t out_cap;
  size_t bytes;
  grain_chain_sizes(n, nodes, &levels, &out_cap, &bytes);
  const uint32_t cap = nodes + 16;
  char *p = static_cast<char *>(chain_scratch);
  a.pos = reinterpret_cast<uint32_t *>(p);
  a.J = reinterpret_cast<uint32_t *>(p + (size_t)cap * 4);
  char *q = p + (size_t)cap * 4 * (size_t)(levels + 1);
  a.starts = reinterpret_cast<int32_t *>(q);
  a.lens = reinterpret_cast<int32_t *>(q + (size_t)out_cap * 4);
  a.firsts = reinterpret_cast<float *>(q + (size_t)out_cap * 8);
  a.levels = levels;
  a.nodes_cap = cap;
  a.out_cap = out_cap;
  *d_starts = a.starts;
  *d_lens = a.lens;
  *d_firsts = a.firsts;
  if (a.nwords == 0) return hipSuccess;
  hipLaunchKernelGGL(chain_succ, dim3((unsigned)((a.nwords + 255) / 256)), dim3(256), 0, s, a);
  if (nodes)
    for (int l = 1; l < levels; ++l)
      hipLaunchKernelGGL(chain_lift, dim3((nodes + 255) / 256), dim3(256), 0, s, a.J + (size_t)(l - 1) * cap,
                         a.J + (size_t)l * cap, nodes);
  hipLaunchKernelGGL(chain_count, dim3(1), dim3(1), 0, s, a);
  hipLaunchKernelGGL(chain_expand, dim3((out_cap + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace mx
