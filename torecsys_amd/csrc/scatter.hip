// K1 backward: row-bucketed index (CSR) + atomics-free segmented gradient reduction.
//
// nn.Embedding's dense gradient (sparse=False) is "zero V x E, then index_add".  Here the B*N lookups
// are bucketed by destination row once per batch (count -> exclusive scan -> fill; int32 atomics
// only on the 4-byte counters), then ONE pass writes every row of the gradient exactly once:
// a group of L lanes (L = 16-byte vectors per row) owns a table row, walks its bucket, accumulates in
// fp32 and stores once (zeros for rows nobody looked up).  No zero-fill pass, no read-modify-write,
// no float atomics; the FM second-order backward dx = g*(S - x) is folded into the same walk:
//   sum_p g_fm[b_p]*(S[b_p] - W[r]) = sum_p g_fm[b_p]*S[b_p]  -  W[r] * sum_p g_fm[b_p].
// HBM-bound: reads B*N*(E*s + 4) (+ 2*B*E*s L2-resident FM operands), writes V*E*s.
#include <stdlib.h>

#include <algorithm>

#include "trs_common.hpp"

#ifndef TRS_SCATTER_NT_LOADS
#define TRS_SCATTER_NT_LOADS 1      // nontemporal loads of the block gradient in the bucket walks (0: plain loads)
#endif

namespace trs {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;
// rows with more than LONG_ROW lookups go to a queue and are reduced by whole waves (Zipf-hot rows)
constexpr int LONG_ROW = 64;
// element path (rows that are not 16-byte multiples, e.g. the E = 1 first-order table): one THREAD walks a row's bucket,
// so already moderately hot rows stall their wave; rows above this go to the queue and are reduced by whole waves
constexpr int LONG_ROW_ELEM = 32;
// very hot rows (a Zipf head row collects thousands of lookups) are cut into chunks of LONG_CHUNK lookups that
// different waves reduce; a second pass adds the chunk partials of a row.  Queue entries are (row, chunk).
// 256 = four rounds of a wave at E = 64 bf16 (8 lane groups x 8 lookups in flight); measured on the Zipf(1.05) DeepFM
// step, same box alternately: 1024 per workgroup 1.292 ms, 1024 per wave 1.294, 256 per wave 1.267, 128 per wave 1.276
constexpr int LONG_CHUNK = 256;
// element path (rows that are not whole 16-byte vectors, e.g. the E = 1 first-order table): a queued row is reduced by ONE
// wave; rows with more than ELEM_SPLIT lookups (a 4-row field collects 16 384 of a 65 536-sample batch) are cut into
// chunks of ELEM_SPLIT lookups, one wave each, whose partial sums meet in fp32 scratch (atomics) and are finished by a
// third small kernel.  Round 4: with one wave per row the criteo-skewed layout spent 115 us here (64 rounds of two
// dependent latencies on a single wave).
constexpr int ELEM_SPLIT = 2048;

// 4 independent lookups per thread per iteration: 4 index loads, then 4 returning atomics in flight
template <typename IdxT>
__global__ __launch_bounds__(256) void csr_count_kernel(const IdxT* __restrict__ idx,
                                                        const int64_t* __restrict__ offsets, int64_t BN, int N,
                                                        int64_t V, int32_t* __restrict__ count,
                                                        int32_t* __restrict__ slot, int32_t* __restrict__ err_flag,
                                                        const int32_t* __restrict__ gate) {
  constexpr int U = 4;
  if (gate != nullptr && *gate == 0) return;      // the partitioned build (below) handles this batch
  const unsigned n_items = (unsigned)BN, uN = (unsigned)N;
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned p0 = blockIdx.x * blockDim.x + threadIdx.x; p0 < n_items; p0 += stride * U) {
    int64_t r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned p = p0 + u * stride;
      r[u] = -1;
      if (p < n_items) {
        r[u] = load_row_id(idx, offsets, (int64_t)p, (int)(p % uN));
        if (r[u] < 0 || r[u] >= V) {
          if (err_flag != nullptr) *err_flag = 1;
          r[u] = -1;
        }
      }
    }
    int sl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) sl[u] = r[u] >= 0 ? atomicAdd(&count[r[u]], 1) : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned p = p0 + u * stride;
      if (p < n_items) slot[p] = sl[u];
    }
  }
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* total, int* lds /* >= 8 ints */) {
  // inclusive scan inside the wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    const int s = lds[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums_kernel(const int32_t* __restrict__ data, int64_t n,
                                                                      int32_t* __restrict__ tile_sums) {
  __shared__ int lds[8];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) s += data[base + i];
  int tot;
  block_exclusive_scan(s, &tot, lds);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(int32_t* __restrict__ tile_sums, int ntiles) {
  __shared__ int lds[8];
  int carry = 0;
  for (int t0 = 0; t0 < ntiles; t0 += SCAN_THREADS) {
    const int i = t0 + threadIdx.x;
    const int v = i < ntiles ? tile_sums[i] : 0;
    int tot;
    const int ex = block_exclusive_scan(v, &tot, lds);
    if (i < ntiles) tile_sums[i] = carry + ex;
    carry += tot;
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(int32_t* __restrict__ data, int64_t n,
                                                                  const int32_t* __restrict__ tile_sums) {
  __shared__ int lds[8];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = base + i < n ? data[base + i] : 0;
    s += v[i];
  }
  int tot;
  int run = block_exclusive_scan(s, &tot, lds) + tile_sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) data[base + i] = run;
    run += v[i];
  }
}

// Exclusive scan of data[0..n) in ONE pass (decoupled look-back): a workgroup draws its tile from a ticket counter
// (status[ntiles]; progress therefore never depends on the order blockIdx values are dispatched in: whoever holds tile t
// started after the holders of tiles < t drew theirs), scans the tile, publishes its aggregate in status[t] (bit 30 =
// aggregate only, bit 31 = inclusive prefix; the caller guarantees counts < 2^30) and finds its exclusive prefix by
// walking back over its predecessors' status words until it meets an inclusive one.  The status words are written and
// polled with relaxed agent-scope atomics (sc1: served by L2, never by a CU's own L1, and visible across XCDs); the
// words carry their own payload, so no other memory has to be ordered with them.  Replaces tile-sums -> one-workgroup
// scan of the sums -> apply (three launches, the middle one a single workgroup).  status[0..ntiles] must be zero on entry.
__global__ __launch_bounds__(SCAN_THREADS) void scan_onepass_kernel(int32_t* __restrict__ data, int64_t n,
                                                                    unsigned* __restrict__ status, int ntiles) {
  __shared__ int lds[8];
  __shared__ int s_prefix;
  __shared__ int s_tile;
  if (threadIdx.x == 0)
    s_tile = (int)__hip_atomic_fetch_add(&status[ntiles], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int tile = s_tile;
  const int64_t base = (int64_t)tile * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = base + i < n ? data[base + i] : 0;
    s += v[i];
  }
  int tot;
  const int ex = block_exclusive_scan(s, &tot, lds);
  // look-back by the whole first wave: lane l polls predecessor tile-1-l (-64, -128, ... in later rounds), the nearest
  // inclusive word among the 64 ends the walk (tiles before the start count as inclusive zeros) and the aggregates in
  // front of it are summed across the lanes -- one L2 round trip per 64 predecessors instead of one per predecessor (a
  // single thread walking ~245 tiles of a 1 M-row table: 36-40 us inside the DeepFM step)
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    if (tile == 0) {
      if (lane == 0) {
        __hip_atomic_store(&status[0], 0x80000000u | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_prefix = 0;
      }
    } else {
      if (lane == 0)
        __hip_atomic_store(&status[tile], 0x40000000u | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int run = 0;
      for (int hi = tile - 1;; hi -= 64) {
        const int p = hi - lane;
        unsigned w = 0x80000000u;
        if (p >= 0) {
          do {
            w = __hip_atomic_load(&status[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w & 0xc0000000u) == 0u) __builtin_amdgcn_s_sleep(1);
          } while ((w & 0xc0000000u) == 0u);
        }
        const unsigned long long incl = __ballot((w & 0x80000000u) != 0u);
        const int first = incl ? __ffsll((long long)incl) - 1 : 64;      // nearest predecessor with an inclusive prefix
        int c = lane <= first ? (int)(w & 0x3fffffffu) : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        run += c;
        if (first < 64) break;
      }
      if (lane == 0) {
        s_prefix = run;
        __hip_atomic_store(&status[tile], 0x80000000u | (unsigned)(run + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  __syncthreads();
  int run = s_prefix + ex;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) data[base + i] = run;
    run += v[i];
  }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void csr_fill_kernel(const IdxT* __restrict__ idx,
                                                       const int64_t* __restrict__ offsets, int64_t BN, int N,
                                                       const int32_t* __restrict__ row_start,
                                                       const int32_t* __restrict__ slot, int32_t* __restrict__ perm,
                                                       const int32_t* __restrict__ gate) {
  constexpr int U = 4;
  if (gate != nullptr && *gate == 0) return;
  const unsigned n_items = (unsigned)BN, uN = (unsigned)N;
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned p0 = blockIdx.x * blockDim.x + threadIdx.x; p0 < n_items; p0 += stride * U) {
    int sl[U];
    int64_t r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned p = p0 + u * stride;
      sl[u] = p < n_items ? slot[p] : -1;
      r[u] = sl[u] >= 0 ? load_row_id(idx, offsets, (int64_t)p, (int)(p % uN)) : 0;
    }
    int base[U];
#pragma unroll
    for (int u = 0; u < U; ++u) base[u] = sl[u] >= 0 ? row_start[r[u]] : 0;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (sl[u] >= 0) perm[base[u] + sl[u]] = (int32_t)(p0 + u * stride);
  }
}

// ---- partitioned build: per-field row ranges, LDS counters ------------------------------------------------
// The global-atomic build above is bound by the rate of returning atomics at the memory side (~25 G/s:
// ~100 us per pass at 2.5 M lookups).  When the table is the concatenation of per-field ranges
// (MultiIndicesEmbedding: field n owns rows [offsets[n], offsets[n+1])), the row space is cut into chunks of
// CSR2_CHUNK rows that never straddle a field; one workgroup owns a chunk, keeps its counters in LDS and
// scans only its field's column of the batch (transposed to (N,B) int32 row ids by a first pass), so all
// atomics are LDS atomics and the global traffic is coalesced.  A lookup outside its field's range (legal for
// nn.Embedding as long as idx+offset < V) or non-monotonic offsets set flags[0]; the workgroups of the
// partitioned kernels then exit and the gated global-atomic kernels run instead -- no host round trip.
constexpr int CSR2_CHUNK = 15360;      // 60 KB of int32 counters
constexpr int CSR2_THREADS = 1024;
constexpr int CSR2_TINY = 32;          // chunks of at most this many rows use privatised counters
constexpr int CSR2_STAGE = 24576;      // positions of perm a fill workgroup can stage in LDS (96 KB)
constexpr int CSR2_COPIES = 16;
constexpr int CSR2_TB = 128;           // samples per transpose tile
constexpr int CSR2_MAX_FIELDS = 120;   // transpose tile (N x 129 int32) stays under 64 KB

// Field ranges may reach outside [0, V): the owner-side build of a row-sharded table passes offsets shifted by the first
// row of its shard (fields below the shard start at negative rows, fields above it past V); lookups that land outside
// are skipped by the row-id pass, and the chunks are cut from the part of every field's range that lies inside.
__device__ __forceinline__ int64_t csr2_clamp(int64_t r, int64_t V) { return r < 0 ? 0 : (r > V ? V : r); }

template <typename IdxT>
__global__ __launch_bounds__(256) void csr2_rowid_kernel(const IdxT* __restrict__ idx,
                                                         const int64_t* __restrict__ offsets, int64_t B, int N,
                                                         int64_t V, int32_t* __restrict__ rowT,
                                                         int32_t* __restrict__ flags, int32_t* __restrict__ err_flag,
                                                         int max_items, int chunk, int need_cover) {
  extern __shared__ int32_t tile[];     // [N][CSR2_TB + 1]
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int64_t items = 0;
    for (int n = 0; n < N; ++n) {
      const int64_t lo = csr2_clamp(offsets[n], V), hi = n + 1 < N ? csr2_clamp(offsets[n + 1], V) : V;
      if (hi > lo) items += (hi - lo + chunk - 1) / chunk;
    }
    // need_cover (the build that does not zero the counters first): rows in front of the first field belong to no chunk
    if (items > max_items || (need_cover && offsets[0] > 0)) flags[0] = 1;
  }
  const int64_t b0 = (int64_t)blockIdx.x * CSR2_TB;
  const int nb = (int)((B - b0) < CSR2_TB ? (B - b0) : CSR2_TB);
  const int total = nb * N;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int bl = e / N, n = e - bl * N;
    const int64_t lo = offsets[n], hi = n + 1 < N ? offsets[n + 1] : V;
    int64_t r = (int64_t)idx[b0 * N + e] + lo;
    if (r < 0 || r >= V) {
      if (err_flag != nullptr) *err_flag = 1;
      r = -1;
    } else if (r < lo || r >= hi) {
      flags[0] = 1;
    }
    tile[n * (CSR2_TB + 1) + bl] = (int32_t)r;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * CSR2_TB; e += 256) {
    const int n = e / CSR2_TB, bl = e - n * CSR2_TB;
    if (bl < nb) rowT[(int64_t)n * B + b0 + bl] = tile[n * (CSR2_TB + 1) + bl];
  }
}

// which (field, chunk) does workgroup `item` own?  returns false when there is none
__device__ __forceinline__ bool csr2_item(const int64_t* __restrict__ offsets, int N, int64_t V, int item, int chunk,
                                          int* field, int64_t* base, int* len) {
  int64_t acc = 0;
  for (int n = 0; n < N; ++n) {
    const int64_t lo = csr2_clamp(offsets[n], V), hi = n + 1 < N ? csr2_clamp(offsets[n + 1], V) : V;
    if (hi <= lo) continue;
    const int64_t nch = (hi - lo + chunk - 1) / chunk;
    if (item < acc + nch) {
      const int64_t c = item - acc;
      *field = n;
      *base = lo + c * chunk;
      *len = (int)((hi - *base) < chunk ? (hi - *base) : chunk);
      return true;
    }
    acc += nch;
  }
  return false;
}

template <bool FILL>
__global__ __launch_bounds__(CSR2_THREADS) void csr2_pass_kernel(const int32_t* __restrict__ rowT,
                                                                 const int64_t* __restrict__ offsets, int64_t B, int N,
                                                                 int64_t V, int32_t* __restrict__ row_start,
                                                                 int32_t* __restrict__ perm,
                                                                 const int32_t* __restrict__ flags, int chunk,
                                                                 int stage_cap) {
  __shared__ int32_t ctr[CSR2_CHUNK];
  __shared__ int s_field, s_len, s_ok;
  __shared__ int64_t s_base;
  if (flags[0] != 0) return;
  if (threadIdx.x == 0) {
    int f = 0, l = 0;
    int64_t bs = 0;
    s_ok = csr2_item(offsets, N, V, (int)blockIdx.x, chunk, &f, &bs, &l) ? 1 : 0;
    s_field = f; s_len = l; s_base = bs;
  }
  __syncthreads();
  if (!s_ok) return;
  const int n = s_field, len = s_len;
  const int64_t base = s_base;
  for (int i = threadIdx.x; i < len; i += CSR2_THREADS) ctr[i] = FILL ? row_start[base + i] : 0;
  // fill pass: the chunk's rows own ONE contiguous piece of perm, [row_start[base], row_start[base + len]).  When it fits
  // the stage (dynamic LDS behind the counters) the positions are scattered into LDS and the piece leaves as contiguous
  // stores -- the pass is bound by its scattered 4-byte global stores otherwise (profiles/r05_logs/ab_csr_rank_fill.txt)
  extern __shared__ int32_t stage[];
  int seg0 = 0, seg_len = 0;
  if (FILL && stage_cap > 0 && len > CSR2_TINY) {
    seg0 = row_start[base];
    seg_len = row_start[base + len] - seg0;
  }
  const bool staged = FILL && stage_cap > 0 && len > CSR2_TINY && seg_len <= stage_cap;
  __syncthreads();
  const int32_t* col = rowT + (int64_t)n * B;
  const int32_t ibase = (int32_t)base;
  constexpr int U = 8;
  if (len <= CSR2_TINY) {
    // A field of a handful of rows (criteo's bucketised numeric fields: 4, 5, 6, 8 ... rows): all 64 lanes of every wave
    // hit the same few counters, and an LDS atomic on one address retires about one lane per clock (csr2_pass_kernel<true>
    // 150 -> 204 us on the skewed layout).  The counters are PRIVATISED here: CSR2_COPIES copies of the chunk's counters,
    // a lane uses copy (lane & 15), so the 64 lanes of a wave spread over 16 x len addresses.  Count pass: add up the
    // copies.  Fill pass: count into the copies first, turn them into start positions (copy c of row v starts behind
    // copies 0..c-1), then walk the column again with returning atomics on the lane's own copy.  (A ballot per row value
    // instead of atomics was tried first: 8 x len dependent ballot rounds per wave iteration, 341 us.)
    __shared__ int32_t ctrp[CSR2_COPIES][CSR2_TINY];
    const int copy = threadIdx.x & (CSR2_COPIES - 1);
    for (int i = threadIdx.x; i < CSR2_COPIES * CSR2_TINY; i += CSR2_THREADS) (&ctrp[0][0])[i] = 0;
    __syncthreads();
    for (int64_t b0 = threadIdx.x; b0 < B; b0 += (int64_t)CSR2_THREADS * U) {
      int32_t r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t b = b0 + (int64_t)u * CSR2_THREADS;
        r[u] = b < B ? col[b] : -1;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned d = (unsigned)(r[u] - ibase);
        if (r[u] >= 0 && d < (unsigned)len) atomicAdd(&ctrp[copy][d], 1);
      }
    }
    __syncthreads();
    if (!FILL) {
      for (int v = threadIdx.x; v < len; v += CSR2_THREADS) {
        int t = 0;
        for (int c = 0; c < CSR2_COPIES; ++c) t += ctrp[c][v];
        row_start[base + v] = t;
      }
      return;
    }
    for (int v = threadIdx.x; v < len; v += CSR2_THREADS) {      // counts -> start positions, per row over the copies
      int run = ctr[v];                                         // = row_start[base + v]
      for (int c = 0; c < CSR2_COPIES; ++c) {
        const int t = ctrp[c][v];
        ctrp[c][v] = run;
        run += t;
      }
    }
    __syncthreads();
    for (int64_t b0 = threadIdx.x; b0 < B; b0 += (int64_t)CSR2_THREADS * U) {
      int32_t r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t b = b0 + (int64_t)u * CSR2_THREADS;
        r[u] = b < B ? col[b] : -1;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned d = (unsigned)(r[u] - ibase);
        if (r[u] >= 0 && d < (unsigned)len) {
          const int pos = atomicAdd(&ctrp[copy][d], 1);
          perm[pos] = (int32_t)((b0 + (int64_t)u * CSR2_THREADS) * N + n);
        }
      }
    }
    return;
  }
  for (int64_t b0 = threadIdx.x; b0 < B; b0 += (int64_t)CSR2_THREADS * U) {
    int32_t r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t b = b0 + (int64_t)u * CSR2_THREADS;
      r[u] = b < B ? col[b] : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned d = (unsigned)(r[u] - ibase);
      if (r[u] >= 0 && d < (unsigned)len) {
        if (FILL) {
          const int pos = atomicAdd(&ctr[d], 1);
          const int32_t p = (int32_t)((b0 + (int64_t)u * CSR2_THREADS) * N + n);
          if (staged) stage[pos - seg0] = p;
          else perm[pos] = p;
        } else {
          atomicAdd(&ctr[d], 1);
        }
      }
    }
  }
  if (staged) {
    __syncthreads();
    for (int i = threadIdx.x; i < seg_len; i += CSR2_THREADS) perm[seg0 + i] = stage[i];
  }
  if (!FILL) {
    __syncthreads();
    for (int i = threadIdx.x; i < len; i += CSR2_THREADS) row_start[base + i] = ctr[i];
  }
}

// Where a finished row sum goes: written out as the gradient row (mode 0), or applied in place to the table
// row by a fused sparse optimizer -- SGD (mode 1) or Adagrad (mode 2).  Both are exactly equivalent to their
// dense torch.optim counterparts (no momentum / weight decay): rows nobody looked up have a zero gradient
// and are not touched, so neither the dense V x E gradient nor a dense optimizer pass over the table exists.
struct RowSink {
  int mode;      // 0 write gradient, 1 SGD, 2 Adagrad, 3 Adam (lazy / SparseAdam semantics)
  float lr, eps; // Adam: lr = lr * sqrt(1 - beta2^t) / (1 - beta1^t), computed by the caller
  float* state;  // Adagrad accumulator / Adam exp_avg (V x E fp32) or null
  float beta1 = 0.f, beta2 = 0.f;
  float* state2 = nullptr;  // Adam exp_avg_sq
  // Optional: the bucketed rows are a COMPACT list of distinct table rows (row_map[r] = table row of bucket row r):
  // the owner side of a row-sharded table, whose 125 M-row shard cannot afford a V-sized bucket index per step.
  const int32_t* row_map = nullptr;
  int64_t map_rows = 0;     // rows of the table row_map points into: an entry outside [0, map_rows) is skipped
};
// table row a bucketed row lands on; -1 = a row_map entry outside the table (the sinks skip negative positions)
__device__ __forceinline__ int64_t sink_row(const RowSink& k, int64_t r) {
  if (k.row_map == nullptr) return r;
  const int64_t m = (int64_t)k.row_map[r];
  return (m >= 0 && m < k.map_rows) ? m : -1;
}

template <typename T>
__device__ __forceinline__ void sink_vec(const RowSink& k, uint4* __restrict__ out, int64_t vec, const float* acc,
                                         bool touched) {
  constexpr int VE = Vec16<T>::VE;
  if (vec < 0) return;
  if (k.mode == 0) {
    // dense gradient table: written once, read by the optimizer much later -- stream it past L2 (which the [g*S | g]
    // rows of the FM term want to keep)
    store_stream(&out[vec], Vec16<T>::pack(acc));
    return;
  }
  if (!touched) return;
  float w[VE];
  Vec16<T>::unpack(out[vec], w);
  if (k.mode == 1) {
#pragma unroll
    for (int i = 0; i < VE; ++i) w[i] = fmaf(-k.lr, acc[i], w[i]);
  } else if (k.mode == 2) {
    float* st = k.state + vec * VE;
#pragma unroll
    for (int i = 0; i < VE; ++i) {
      const float s2 = fmaf(acc[i], acc[i], st[i]);
      st[i] = s2;
      w[i] -= k.lr * acc[i] / (sqrtf(s2) + k.eps);
    }
  } else {
    float* m1 = k.state + vec * VE;
    float* m2 = k.state2 + vec * VE;
#pragma unroll
    for (int i = 0; i < VE; ++i) {
      const float a = m1[i] + (acc[i] - m1[i]) * (1.f - k.beta1);
      const float v = m2[i] + (acc[i] * acc[i] - m2[i]) * (1.f - k.beta2);
      m1[i] = a;
      m2[i] = v;
      w[i] -= k.lr * a / (sqrtf(v) + k.eps);
    }
  }
  out[vec] = Vec16<T>::pack(w);
}

template <typename T>
__device__ __forceinline__ void sink_elem(const RowSink& k, T* __restrict__ out, int64_t idx, float acc, bool touched) {
  if (idx < 0) return;
  if (k.mode == 0) {
    out[idx] = from_f32<T>(acc);
    return;
  }
  if (!touched) return;
  float w = to_f32(out[idx]);
  if (k.mode == 1) {
    w = fmaf(-k.lr, acc, w);
  } else if (k.mode == 2) {
    const float s2 = fmaf(acc, acc, k.state[idx]);
    k.state[idx] = s2;
    w -= k.lr * acc / (sqrtf(s2) + k.eps);
  } else {
    const float a = k.state[idx] + (acc - k.state[idx]) * (1.f - k.beta1);
    const float v = k.state2[idx] + (acc * acc - k.state2[idx]) * (1.f - k.beta2);
    k.state[idx] = a;
    k.state2[idx] = v;
    w -= k.lr * a / (sqrtf(v) + k.eps);
  }
  out[idx] = from_f32<T>(w);
}

// ---------------------------------------------------------------------------------------------
// segmented reduction: one L-lane group per table row (rows with > LONG_ROW lookups are deferred
// to a queue and reduced by whole waves afterwards)
// HAS_F1: a companion E = 1 table (the first-order term of the same lookups) rides in the same walk: g_first holds one
// value per lookup (B*N), every lane of the group adds the same ones into *f1
// Where the FM term of a lookup of sample b is read from (16-byte vectors, table dtype): t[b*tstr + lane] = g*S and
// g[b*gstr + (gstr > 1 ? lane : 0)] = g.  Full g_fm rows: one [g*S | g] row of 2L vectors per sample (tstr = gstr = 2L,
// g = t + L).  g_fm constant along E (the gradient of a sum over E, what the reference's FM / DeepFM models feed back):
// g*S rows of L vectors and ONE vector per sample holding g in every element -- 144 instead of 256 bytes per lookup,
// the g vectors (16 B x B) stay in L2.
struct FmSrc {
  const uint4* t;
  const uint4* g;
  int tstr, gstr;
  const void* g1 = nullptr;   // SCAL kernels: the per-sample gradient scalars themselves (B values of the table dtype)
};

// SCAL: the FM gradient is one scalar per sample (g constant along E): it is read as that scalar (2-4 bytes, L2-resident)
// instead of a 16-byte vector of copies, and its sum over the bucket is ONE register (gsum[0]) instead of VE
template <typename T, int LOG2L, bool HAS_G, bool HAS_FM, int CH = 4, bool HAS_F1 = false, bool SCAL = false>
__device__ __forceinline__ void accumulate_bucket(float* acc, float* gsum, const uint4* __restrict__ g_rows,
                                                  const uint4* __restrict__ g_fm, const float* __restrict__ fm_sum,
                                                  const int32_t* __restrict__ perm, int beg, int end, int step,
                                                  int N, int64_t gbs, int lane_v, FmSrc fs,
                                                  const T* __restrict__ g_first = nullptr, float* f1 = nullptr) {
  constexpr int L = 1 << LOG2L;
  constexpr int VE = Vec16<T>::VE;
  // CH lookups in flight per lane; the bucket positions of the NEXT round are fetched while this round's gradient rows
  // are on their way (two dependent latencies per round otherwise: perm, then the rows it points at)
  int pn[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) pn[c] = (beg + c * step) < end ? perm[beg + c * step] : -1;
  for (int q = beg; q < end; q += CH * step) {
    int p[CH];
    uint4 gv[CH], fv[CH], tv[CH];
    float fr[CH], gs[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) p[c] = pn[c];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int qn = q + (CH + c) * step;
      pn[c] = qn < end ? perm[qn] : -1;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      gv[c] = make_uint4(0, 0, 0, 0);
      fv[c] = make_uint4(0, 0, 0, 0);
      tv[c] = make_uint4(0, 0, 0, 0);
      fr[c] = 0.f;
      gs[c] = 0.f;
      if (p[c] >= 0) {
        if (HAS_F1) fr[c] = to_f32(g_first[p[c]]);
        if (HAS_G) {
          int64_t row = p[c];
          if (gbs != N) {  // g_rows is a strided slice: sample b starts at row b*gbs
            const unsigned b = (unsigned)p[c] / (unsigned)N;
            row = (int64_t)b * gbs + ((unsigned)p[c] - b * (unsigned)N);
          }
#if TRS_SCATTER_NT_LOADS
          gv[c] = load_stream(&g_rows[row * L + lane_v]);
#else
          gv[c] = g_rows[row * L + lane_v];
#endif
        }
        if (HAS_FM) {
          const int64_t b = (int64_t)((unsigned)p[c] / (unsigned)N);
          if (fm_sum != nullptr) {  // FM mode: see FmSrc
            tv[c] = fs.t[b * fs.tstr + lane_v];
            if (SCAL) gs[c] = to_f32(static_cast<const T*>(fs.g1)[b]);
            else fv[c] = fs.g[b * fs.gstr + (fs.gstr > 1 ? lane_v : 0)];
          } else {
            fv[c] = g_fm[b * L + lane_v];
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (HAS_F1) *f1 += fr[c];
      if (p[c] >= 0) {
        if (HAS_G) {
          float x[VE];
          Vec16<T>::unpack(gv[c], x);
#pragma unroll
          for (int k = 0; k < VE; ++k) acc[k] += x[k];
        }
        if (HAS_FM) {
          float gf[VE];
          if (!SCAL) Vec16<T>::unpack(fv[c], gf);
          if (fm_sum != nullptr) {
            float tg[VE];
            Vec16<T>::unpack(tv[c], tg);
#pragma unroll
            for (int k = 0; k < VE; ++k) acc[k] += tg[k];
            if (SCAL) {
              gsum[0] += gs[c];
            } else {
#pragma unroll
              for (int k = 0; k < VE; ++k) gsum[k] += gf[k];
            }
          } else {  // plain per-sample broadcast gradient (first-order sum): no FM weighting
#pragma unroll
            for (int k = 0; k < VE; ++k) acc[k] += gf[k];
          }
        }
      }
    }
  }
}

template <typename T, int LOG2L, bool HAS_G, bool HAS_FM, bool HAS_F1 = false, bool SCAL = false, int CHF = 2>
__global__ __launch_bounds__(256) void scatter_rows_group_kernel(
    const uint4* __restrict__ g_rows, const uint4* __restrict__ g_fm, const float* __restrict__ fm_sum,
    const uint4* __restrict__ table, const int32_t* __restrict__ row_start, const int32_t* __restrict__ perm,
    int64_t V, int N, int64_t gbs, int64_t padding_row, uint4* __restrict__ grad,
    int32_t* __restrict__ long_rows /* [0]=count */, RowSink sink, FmSrc fs, const T* __restrict__ g_first = nullptr,
    T* __restrict__ grad_first = nullptr) {
  constexpr int L = 1 << LOG2L;
  constexpr int VE = Vec16<T>::VE;
  const int lane_v = threadIdx.x & (L - 1);
  const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> LOG2L;
  int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> LOG2L;
  int nbeg = 0, nend = 0;
  if (r < V) { nbeg = row_start[r]; nend = row_start[r + 1]; }
  for (; r < V; r += groups) {
    const int beg = nbeg, end = nend;
    // the bucket bounds of the NEXT row and this row's table values (FM term) are requested before the dependent
    // perm -> gradient-row chain of this bucket starts, so their latency hides behind it
    if (r + groups < V) { nbeg = row_start[r + groups]; nend = row_start[r + groups + 1]; }
    uint4 wraw = make_uint4(0, 0, 0, 0);
    if (HAS_FM && fm_sum != nullptr) wraw = table[r * L + lane_v];
    float acc[VE], gsum[VE];
#pragma unroll
    for (int k = 0; k < VE; ++k) { acc[k] = 0.f; gsum[k] = 0.f; }
    if (end - beg > LONG_ROW && r != padding_row) {
      if (lane_v == 0) {
        const int nch = (end - beg + LONG_CHUNK - 1) / LONG_CHUNK;
        const int slot = atomicAdd(&long_rows[0], nch);      // the chunks of a row are adjacent in the queue
        for (int c = 0; c < nch; ++c) {
          long_rows[1 + 2 * (slot + c)] = (int32_t)r;
          long_rows[2 + 2 * (slot + c)] = c;
        }
      }
      continue;
    }
    float f1 = 0.f;
    if (r != padding_row) {
      accumulate_bucket<T, LOG2L, HAS_G, HAS_FM, (HAS_FM ? CHF : 4), HAS_F1, SCAL>(acc, gsum, g_rows, g_fm, fm_sum, perm, beg,
                                                                                end, 1, N, gbs, lane_v, fs, g_first, &f1);
      if (HAS_FM && fm_sum != nullptr && end > beg) {
        float w[VE];
        Vec16<T>::unpack(wraw, w);
#pragma unroll
        for (int k = 0; k < VE; ++k) acc[k] = fmaf(-w[k], gsum[SCAL ? 0 : k], acc[k]);
      }
    }
    sink_vec<T>(sink, grad, sink_row(sink, r) * L + lane_v, acc, end > beg && r != padding_row);
    if (HAS_F1 && lane_v == 0) grad_first[r] = from_f32<T>(f1);      // dense companion gradient: every row written
  }
}

// The models' case, lean: FM gradient constant along E (one scalar per sample), rows of whole 16-byte vectors, no
// companion table.  Same walk as scatter_rows_group_kernel, written for 64 registers (8 waves per SIMD: the walk is a
// chain of dependent latencies -- bucket bounds -> bucket positions -> gradient rows -- and its throughput is the number
// of rows in flight; the generic FM kernel needs 92 registers = 5 waves per SIMD and ran 193 us where the plain walk,
// at 8 waves, takes 105): 32-bit element offsets from uniform bases (BN * L < 2^31 is checked by the host), the sample
// index by a multiply-shift with a host-computed reciprocal, the table row fetched with the bucket bounds.
template <typename T, int LOG2L, bool HAS_G>
__global__ __launch_bounds__(256, 8) void scatter_rows_fm1_kernel(
    const uint4* __restrict__ g_rows, const uint4* __restrict__ tg, const T* __restrict__ g1,
    const uint4* __restrict__ table, const int32_t* __restrict__ row_start, const int32_t* __restrict__ perm, int V,
    unsigned N, unsigned rcpN /* ceil(2^32 / N) */, int padding_row, uint4* __restrict__ grad,
    int32_t* __restrict__ long_rows /* [0]=count */, RowSink sink) {
  constexpr int L = 1 << LOG2L;
  constexpr int VE = Vec16<T>::VE;
  constexpr int CH = 2;      // 3 / 4 lookups in flight per lane cost 8 / 16 registers = 7 / 5 waves per SIMD: 171 / 191 us vs 163
  const unsigned lane_v = threadIdx.x & (L - 1);
  const int groups = (int)(((int64_t)gridDim.x * blockDim.x) >> LOG2L);
  int r = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> LOG2L);
  int nbeg = 0, nend = 0;
  if (r < V) { nbeg = row_start[r]; nend = row_start[r + 1]; }
  for (; r < V; r += groups) {
    const int beg = nbeg, end = nend;
    if (r + groups < V) { nbeg = row_start[r + groups]; nend = row_start[r + groups + 1]; }
    const uint4 wraw = table[(unsigned)r * L + lane_v];
    float acc[VE], gsum = 0.f;
#pragma unroll
    for (int k = 0; k < VE; ++k) acc[k] = 0.f;
    const bool live = r != padding_row;
    if (end - beg > LONG_ROW && live) {
      if (lane_v == 0) {
        const int nch = (end - beg + LONG_CHUNK - 1) / LONG_CHUNK;
        const int slot = atomicAdd(&long_rows[0], nch);
        for (int c = 0; c < nch; ++c) {
          long_rows[1 + 2 * (slot + c)] = r;
          long_rows[2 + 2 * (slot + c)] = c;
        }
      }
      continue;
    }
    if (live) {
      int pn[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) pn[c] = (beg + c) < end ? perm[beg + c] : -1;
      for (int q = beg; q < end; q += CH) {
        int p[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) p[c] = pn[c];
#pragma unroll
        for (int c = 0; c < CH; ++c) pn[c] = (q + CH + c) < end ? perm[q + CH + c] : -1;
        uint4 gv[CH], tv[CH];
        float gs[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          gv[c] = make_uint4(0, 0, 0, 0);
          tv[c] = make_uint4(0, 0, 0, 0);
          gs[c] = 0.f;
          if (p[c] >= 0) {
            const unsigned b = __umulhi((unsigned)p[c], rcpN);      // p / N for p < 2^31, N < 2^16 (host-checked)
#if TRS_SCATTER_NT_LOADS
            if (HAS_G) gv[c] = load_stream(&g_rows[(unsigned)p[c] * L + lane_v]);      // read once: see load_stream
#else
            if (HAS_G) gv[c] = g_rows[(unsigned)p[c] * L + lane_v];
#endif
            tv[c] = tg[b * L + lane_v];
            gs[c] = to_f32(g1[b]);
          }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          float x[VE], t[VE];
          if (HAS_G) Vec16<T>::unpack(gv[c], x);
          Vec16<T>::unpack(tv[c], t);
#pragma unroll
          for (int k = 0; k < VE; ++k) acc[k] += HAS_G ? x[k] + t[k] : t[k];
          gsum += gs[c];
        }
      }
      if (end > beg) {
        float w[VE];
        Vec16<T>::unpack(wraw, w);
#pragma unroll
        for (int k = 0; k < VE; ++k) acc[k] = fmaf(-w[k], gsum, acc[k]);
      }
    }
    sink_vec<T>(sink, grad, sink_row(sink, r) * L + lane_v, acc, end > beg && live);
  }
}

// companion (E = 1) gradient of the hot rows the group kernel queued: one wave per queued row (its first chunk entry),
// lanes stride the whole bucket, wavefront reduction
template <typename T>
__global__ __launch_bounds__(256) void scatter_first_long_kernel(const T* __restrict__ g_first,
                                                                 const int32_t* __restrict__ row_start,
                                                                 const int32_t* __restrict__ perm,
                                                                 const int32_t* __restrict__ long_rows,
                                                                 T* __restrict__ grad_first) {
  const int nlong = long_rows[0];
  const int lane = threadIdx.x & 63;
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int i = wid; i < nlong; i += nwaves) {
    if (long_rows[2 + 2 * i] != 0) continue;
    const int64_t r = long_rows[1 + 2 * i];
    const int beg = row_start[r], end = row_start[r + 1];
    float acc = 0.f;
    constexpr int U = 4;
    for (int q = beg + lane; q < end; q += 64 * U) {
      int pp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) pp[u] = (q + 64 * u) < end ? perm[q + 64 * u] : -1;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (pp[u] >= 0) acc += to_f32(g_first[pp[u]]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) grad_first[r] = from_f32<T>(acc);
  }
}

// hot rows: one WAVE per (row, chunk) queue entry -- its 64 / L lane groups stride over the chunk with eight lookups in
// flight each, then the groups' partial sums are folded by shuffles: no LDS, no barrier.  (Round 5: the entry used to
// belong to a whole 256-thread workgroup with an LDS tree and six barriers; on Zipf(1.05) indices most queue entries are
// rows of 65 ... 300 lookups, for which 7 of its 8 lane-group rounds had nothing to do -- 92 us at the end of the step.)
// Rows of a single chunk are finished here; otherwise the chunk's partial sums go to scratch[entry][2][E] (fp32)
// and scatter_long_rows_finish_kernel adds the chunks of the row.
template <typename T, int LOG2L, bool HAS_G, bool HAS_FM, bool SCAL = false>
__global__ __launch_bounds__(256) void scatter_long_rows_kernel(
    const uint4* __restrict__ g_rows, const uint4* __restrict__ g_fm, const float* __restrict__ fm_sum,
    const uint4* __restrict__ table, const int32_t* __restrict__ row_start, const int32_t* __restrict__ perm, int N,
    int64_t gbs, uint4* __restrict__ grad, const int32_t* __restrict__ long_rows, float* __restrict__ scratch,
    RowSink sink, FmSrc fs) {
  constexpr int L = 1 << LOG2L;
  constexpr int VE = Vec16<T>::VE;
  constexpr int G = 64 / L;  // lane groups per wave
  const int lane = threadIdx.x & 63;
  const int lane_v = lane & (L - 1);
  const int grp = lane >> LOG2L;
  const int nlong = long_rows[0];
  const int waves = gridDim.x * (blockDim.x >> 6);
  for (int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < nlong; i += waves) {
    const int64_t r = long_rows[1 + 2 * i];
    const int c = long_rows[2 + 2 * i];
    const int rbeg = row_start[r], rend = row_start[r + 1];
    const int beg = rbeg + c * LONG_CHUNK, end = rend < beg + LONG_CHUNK ? rend : beg + LONG_CHUNK;
    const bool single = rend - rbeg <= LONG_CHUNK;
    float acc[VE], gsum[VE];
#pragma unroll
    for (int k = 0; k < VE; ++k) { acc[k] = 0.f; gsum[k] = 0.f; }
    accumulate_bucket<T, LOG2L, HAS_G, HAS_FM, 8, false, SCAL>(acc, gsum, g_rows, g_fm, fm_sum, perm, beg + grp, end, G, N, gbs,
                                                               lane_v, fs);
#pragma unroll
    for (int m = L; m < 64; m <<= 1) {
#pragma unroll
      for (int k = 0; k < VE; ++k) acc[k] += __shfl_xor(acc[k], m, 64);
#pragma unroll
      for (int k = 0; k < (SCAL ? 1 : VE); ++k) gsum[k] += __shfl_xor(gsum[k], m, 64);
    }
    if (SCAL) {
#pragma unroll
      for (int k = 1; k < VE; ++k) gsum[k] = gsum[0];
    }
    if (grp == 0) {
      if (single) {
        if (HAS_FM && fm_sum != nullptr) {
          float w[VE];
          Vec16<T>::unpack(table[r * L + lane_v], w);
#pragma unroll
          for (int k = 0; k < VE; ++k) acc[k] = fmaf(-w[k], gsum[k], acc[k]);
        }
        sink_vec<T>(sink, grad, sink_row(sink, r) * L + lane_v, acc, true);
      } else {
        float* sa = scratch + ((size_t)i * 2) * (L * VE) + lane_v * VE;
#pragma unroll
        for (int k = 0; k < VE; ++k) { sa[k] = acc[k]; sa[L * VE + k] = gsum[k]; }
      }
    }
  }
}

template <typename T, int LOG2L, bool HAS_FM>
__global__ __launch_bounds__(256) void scatter_long_rows_finish_kernel(
    const float* __restrict__ fm_sum, const uint4* __restrict__ table, const int32_t* __restrict__ row_start,
    uint4* __restrict__ grad, const int32_t* __restrict__ long_rows, const float* __restrict__ scratch, RowSink sink) {
  constexpr int L = 1 << LOG2L;
  constexpr int VE = Vec16<T>::VE;
  const int lane_v = threadIdx.x & (L - 1);
  const int nlong = long_rows[0];
  const int groups = (gridDim.x * blockDim.x) >> LOG2L;
  for (int i = (blockIdx.x * blockDim.x + threadIdx.x) >> LOG2L; i < nlong; i += groups) {
    if (long_rows[2 + 2 * i] != 0) continue;                 // only the first chunk of a row finishes it
    const int64_t r = long_rows[1 + 2 * i];
    const int len = row_start[r + 1] - row_start[r];
    if (len <= LONG_CHUNK) continue;                         // single-chunk rows were finished by the first pass
    const int nch = (len + LONG_CHUNK - 1) / LONG_CHUNK;
    float acc[VE], gsum[VE];
#pragma unroll
    for (int k = 0; k < VE; ++k) { acc[k] = 0.f; gsum[k] = 0.f; }
    // A Zipf head row or a row of a 4-row field collects 16 k-20 k lookups = 64-80 chunk partials, each a cross-XCD read
    // of ~1 k cycles: added one after the other this loop kept the kernel on the critical path for 42 us on the
    // criteo-skewed Zipf batch (profiles/r06_logs/skewed_zipf_timeline_before.md).  Eight partials are requested before
    // the first is added.  (The same sum spread over the lane groups of a wave with a shuffle tree at the end made hipcc
    // run for more than 20 minutes on this file; this form compiles in its usual 30 s.)
    constexpr int U = 8;
    int c = 0;
    for (; c + U <= nch; c += U) {
      float4 va[U][VE / 4], vg[U][VE / 4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float4* sa = reinterpret_cast<const float4*>(scratch + ((size_t)(i + c + u) * 2) * (L * VE) + lane_v * VE);
        const float4* sg = reinterpret_cast<const float4*>(scratch + ((size_t)(i + c + u) * 2 + 1) * (L * VE) + lane_v * VE);
#pragma unroll
        for (int k = 0; k < VE / 4; ++k) { va[u][k] = sa[k]; vg[u][k] = sg[k]; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < VE / 4; ++k) {
          acc[4 * k] += va[u][k].x; acc[4 * k + 1] += va[u][k].y; acc[4 * k + 2] += va[u][k].z; acc[4 * k + 3] += va[u][k].w;
          gsum[4 * k] += vg[u][k].x; gsum[4 * k + 1] += vg[u][k].y; gsum[4 * k + 2] += vg[u][k].z; gsum[4 * k + 3] += vg[u][k].w;
        }
    }
    for (; c < nch; ++c) {
      const float* sa = scratch + ((size_t)(i + c) * 2) * (L * VE) + lane_v * VE;
#pragma unroll
      for (int k = 0; k < VE; ++k) { acc[k] += sa[k]; gsum[k] += sa[L * VE + k]; }
    }
    if (HAS_FM && fm_sum != nullptr) {
      float w[VE];
      Vec16<T>::unpack(table[r * L + lane_v], w);
#pragma unroll
      for (int k = 0; k < VE; ++k) acc[k] = fmaf(-w[k], gsum[k], acc[k]);
    }
    sink_vec<T>(sink, grad, sink_row(sink, r) * L + lane_v, acc, true);
  }
}

// TG[b] = [ g_fm[b,:] * fm_sum[b,:]  |  g_fm[b,:] ]  (2*E values per sample, table dtype): the FM backward
// operands of one sample become ONE contiguous 2*E*s-byte read per lookup (instead of a g_fm row plus a
// 4*E-byte fp32 fm_sum row at two random addresses).
template <typename T>
__global__ __launch_bounds__(256) void build_tg_kernel(const uint4* __restrict__ g_fm, const float* __restrict__ fm_sum,
                                                       uint4* __restrict__ tg, int64_t B, int L) {
  constexpr int VE = Vec16<T>::VE;
  const int64_t total = B * L, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / L;
    const int lv = (int)(t - b * L);
    const uint4 gv = g_fm[t];
    float g[VE], o[VE];
    Vec16<T>::unpack(gv, g);
    const float* sp = fm_sum + t * VE;
#pragma unroll
    for (int k = 0; k < VE; ++k) o[k] = g[k] * sp[k];
    tg[b * 2 * L + lv] = Vec16<T>::pack(o);
    tg[b * 2 * L + L + lv] = gv;
  }
}

// g_fm constant along E: T[b] = g[b] * fm_sum[b,:] (L vectors per sample) and gvec[b] = g[b] in every element of one vector
template <typename T>
__global__ __launch_bounds__(256) void build_tg1_kernel(const T* __restrict__ g1, const float* __restrict__ fm_sum,
                                                        uint4* __restrict__ tg, uint4* __restrict__ gvec, int64_t B, int L) {
  constexpr int VE = Vec16<T>::VE;
  const int64_t total = B * L, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / L;
    const float g = to_f32(g1[b]);
    float o[VE], gg[VE];
    const float* sp = fm_sum + t * VE;
#pragma unroll
    for (int k = 0; k < VE; ++k) { o[k] = g * sp[k]; gg[k] = g; }
    tg[t] = Vec16<T>::pack(o);
    if (t == b * L) gvec[b] = Vec16<T>::pack(gg);
  }
}

// generic element path (any E): one thread per (row, e); hot rows are queued like in the group path
template <typename T>
__device__ __forceinline__ float elem_term(const T* __restrict__ g_rows, const T* __restrict__ g_fm,
                                           const float* __restrict__ fm_sum, int64_t p, int E, int N, int64_t gbs, int e,
                                           float* gsum, int gcols) {
  const int64_t b = p / N;
  float acc = 0.f;
  if (g_rows != nullptr) acc += to_f32(g_rows[(b * gbs + (p - b * N)) * E + e]);
  if (g_fm != nullptr) {
    const float gf = to_f32(g_fm[gcols == 1 ? b : b * E + e]);
    if (fm_sum != nullptr) {
      acc = fmaf(gf, fm_sum[b * E + e], acc);
      *gsum += gf;
    } else {
      acc += gf;
    }
  }
  return acc;
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_elem_kernel(
    const T* __restrict__ g_rows, const T* __restrict__ g_fm, const float* __restrict__ fm_sum,
    const T* __restrict__ table, const int32_t* __restrict__ row_start, const int32_t* __restrict__ perm, int64_t V,
    int E, int N, int64_t gbs, int64_t padding_row, T* __restrict__ grad, int32_t* __restrict__ long_rows,
    RowSink sink, int gcols, int32_t* __restrict__ split_rows /* [0] = count, then (row, chunk) pairs */,
    float* __restrict__ split_acc /* [entry][2][E] */) {
  const int64_t total = V * E;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool f32 = total < ((int64_t)1 << 32);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t r = udiv_fast(t, E, f32);
    const int e = (int)(t - r * E);
    float acc = 0.f, gsum = 0.f;
    bool touched = false;
    if (r != padding_row) {
      const int beg = row_start[r], end = row_start[r + 1];
      touched = end > beg;
      if (end - beg > ELEM_SPLIT) {
        if (e == 0) {
          const int nch = (end - beg + ELEM_SPLIT - 1) / ELEM_SPLIT;
          const int slot = atomicAdd(&split_rows[0], nch);      // the chunks of a row are adjacent in the queue
          for (int c = 0; c < nch; ++c) {
            split_rows[1 + 2 * (slot + c)] = (int32_t)r;
            split_rows[2 + 2 * (slot + c)] = c;
          }
          for (int k = 0; k < 2 * E; ++k) split_acc[(size_t)slot * 2 * E + k] = 0.f;      // the row's accumulators
        }
        continue;
      }
      if (end - beg > LONG_ROW_ELEM) {
        if (e == 0) {
          const int slot = atomicAdd(&long_rows[0], 1);
          long_rows[1 + slot] = (int32_t)r;
        }
        continue;
      }
      for (int q = beg; q < end; ++q) acc += elem_term<T>(g_rows, g_fm, fm_sum, perm[q], E, N, gbs, e, &gsum, gcols);
      if (g_fm != nullptr && fm_sum != nullptr && end > beg) acc = fmaf(-to_f32(table[t]), gsum, acc);
    }
    sink_elem<T>(sink, grad, sink_row(sink, r) * E + e, acc, touched);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_long_rows_elem_kernel(
    const T* __restrict__ g_rows, const T* __restrict__ g_fm, const float* __restrict__ fm_sum,
    const T* __restrict__ table, const int32_t* __restrict__ row_start, const int32_t* __restrict__ perm, int E, int N,
    int64_t gbs, T* __restrict__ grad, const int32_t* __restrict__ long_rows, RowSink sink, int gcols) {
  // one wave per queued row: lanes stride the bucket, wavefront reduction, no workgroup barriers
  const int nlong = long_rows[0];
  const int lane = threadIdx.x & 63;
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int i = wid; i < nlong; i += nwaves) {
    const int64_t r = long_rows[1 + i];
    const int beg = row_start[r], end = row_start[r + 1];
    for (int e = 0; e < E; ++e) {
      float acc = 0.f, gsum = 0.f;
      // 4 bucket positions per lane per round: their perm entries are fetched first, then the 4 gradient values --
      // a row of 13 000 lookups (the Zipf head of a field) is otherwise 200 rounds of two dependent latencies
      constexpr int U = 4;
      for (int q = beg + lane; q < end; q += 64 * U) {
        int pp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) pp[u] = (q + 64 * u) < end ? perm[q + 64 * u] : -1;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (pp[u] >= 0) acc += elem_term<T>(g_rows, g_fm, fm_sum, pp[u], E, N, gbs, e, &gsum, gcols);
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        acc += __shfl_xor(acc, m, 64);
        gsum += __shfl_xor(gsum, m, 64);
      }
      if (lane == 0) {
        if (g_fm != nullptr && fm_sum != nullptr) acc = fmaf(-to_f32(table[r * E + e]), gsum, acc);
        sink_elem<T>(sink, grad, sink_row(sink, r) * E + e, acc, true);
      }
    }
  }
}

// very hot rows of the element path: one wave per (row, chunk of ELEM_SPLIT lookups) entry, partial sums added into the
// row's fp32 accumulators (the first entry of the row owns them)
template <typename T>
__global__ __launch_bounds__(256) void scatter_split_rows_elem_kernel(
    const T* __restrict__ g_rows, const T* __restrict__ g_fm, const float* __restrict__ fm_sum,
    const int32_t* __restrict__ row_start, const int32_t* __restrict__ perm, int E, int N, int64_t gbs,
    const int32_t* __restrict__ split_rows, float* __restrict__ split_acc, int gcols) {
  const int nent = split_rows[0];
  const int lane = threadIdx.x & 63;
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int i = wid; i < nent; i += nwaves) {
    const int64_t r = split_rows[1 + 2 * i];
    const int c = split_rows[2 + 2 * i];
    const int rbeg = row_start[r], rend = row_start[r + 1];
    const int beg = rbeg + c * ELEM_SPLIT, end = rend < beg + ELEM_SPLIT ? rend : beg + ELEM_SPLIT;
    float* racc = split_acc + (size_t)(i - c) * 2 * E;
    for (int e = 0; e < E; ++e) {
      float acc = 0.f, gsum = 0.f;
      constexpr int U = 4;
      for (int q = beg + lane; q < end; q += 64 * U) {
        int pp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) pp[u] = (q + 64 * u) < end ? perm[q + 64 * u] : -1;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (pp[u] >= 0) acc += elem_term<T>(g_rows, g_fm, fm_sum, pp[u], E, N, gbs, e, &gsum, gcols);
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        acc += __shfl_xor(acc, m, 64);
        gsum += __shfl_xor(gsum, m, 64);
      }
      if (lane == 0) {
        atomicAdd(&racc[e], acc);
        atomicAdd(&racc[E + e], gsum);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_split_rows_finish_elem_kernel(
    const T* __restrict__ g_fm, const float* __restrict__ fm_sum, const T* __restrict__ table, int E,
    T* __restrict__ grad, const int32_t* __restrict__ split_rows, const float* __restrict__ split_acc, RowSink sink) {
  const int nent = split_rows[0];
  const int64_t total = (int64_t)nent * E;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(t / E), e = (int)(t - (int64_t)i * E);
    if (split_rows[2 + 2 * i] != 0) continue;                 // the first chunk of a row finishes it
    const int64_t r = split_rows[1 + 2 * i];
    float acc = split_acc[(size_t)i * 2 * E + e];
    if (g_fm != nullptr && fm_sum != nullptr) acc = fmaf(-to_f32(table[r * E + e]), split_acc[(size_t)i * 2 * E + E + e], acc);
    sink_elem<T>(sink, grad, sink_row(sink, r) * E + e, acc, true);
  }
}

static int log2_lanes_sc(int row_bytes) {
  if (row_bytes % 16 != 0) return -1;
  const int L = row_bytes / 16;
  if (!is_pow2(L) || L > 64) return -1;
  int l = 0;
  while ((1 << l) < L) ++l;
  return l;
}

template <typename T, int LOG2L>
static void scatter_group_launch(const void* g_rows, const void* g_fm, const float* fm_sum, const void* table,
                                 const int32_t* row_start, const int32_t* perm, int64_t V, int N, int64_t gbs,
                                 int64_t padding_row, void* grad, int32_t* long_rows, void* tg, float* scratch,
                                 int64_t B, RowSink sink, hipStream_t s, int gcols, const void* g_first = nullptr,
                                 void* grad_first = nullptr) {
  const int L = 1 << LOG2L;
  FmSrc fs{nullptr, nullptr, 0, 0};
  bool scal = false;
  static const int variant = getenv("TRS_SCATTER_VARIANT") ? atoi(getenv("TRS_SCATTER_VARIANT")) : 1;   // 0: the generic FM walk (A/B)
  if (g_fm != nullptr && fm_sum != nullptr) {
    if (gcols == 1) {      // (L == 1 too: a (B,1) buffer must never be read as B 16-byte rows)
      uint4* gvec = (uint4*)tg + B * L;
      hipLaunchKernelGGL((build_tg1_kernel<T>), dim3(stream_grid(B * L, 256, 4096)), dim3(256), 0, s, (const T*)g_fm,
                         fm_sum, (uint4*)tg, gvec, B, L);
      fs = FmSrc{(const uint4*)tg, gvec, L, 1};
      fs.g1 = g_fm;
      scal = variant != 0 && g_first == nullptr;
    } else {
      hipLaunchKernelGGL((build_tg_kernel<T>), dim3(stream_grid(B * L, 256, 4096)), dim3(256), 0, s, (const uint4*)g_fm,
                         fm_sum, (uint4*)tg, B, L);
      fs = FmSrc{(const uint4*)tg, (const uint4*)tg + L, 2 * L, 2 * L};
    }
    g_fm = tg;
  }
  const int grid = stream_grid(V * L, 256, 256 * 32);
  const bool hg = g_rows != nullptr, hf = g_fm != nullptr;
#define TRS_SC_TAIL(HG, HF, SC)                                                                                 \
  do {                                                                                                          \
    hipLaunchKernelGGL((scatter_long_rows_kernel<T, LOG2L, HG, HF, SC>), dim3(2048), dim3(256), 0, s,            \
                       (const uint4*)g_rows, (const uint4*)g_fm, fm_sum, (const uint4*)table, row_start, perm, N, \
                       gbs, (uint4*)grad, long_rows, scratch, sink, fs);                                              \
    hipLaunchKernelGGL((scatter_long_rows_finish_kernel<T, LOG2L, HF>), dim3(64), dim3(256), 0, s, fm_sum,           \
                       (const uint4*)table, row_start, (uint4*)grad, long_rows, scratch, sink);                       \
  } while (0)
#define TRS_SC(HG, HF)                                                                                          \
  do {                                                                                                          \
    if (g_first != nullptr)                                                                                     \
      hipLaunchKernelGGL((scatter_rows_group_kernel<T, LOG2L, HG, HF, true>), dim3(grid), dim3(256), 0, s,       \
                         (const uint4*)g_rows, (const uint4*)g_fm, fm_sum, (const uint4*)table, row_start, perm, \
                         V, N, gbs, padding_row, (uint4*)grad, long_rows, sink, fs, (const T*)g_first,           \
                         (T*)grad_first);                                                                        \
    else                                                                                                        \
      hipLaunchKernelGGL((scatter_rows_group_kernel<T, LOG2L, HG, HF>), dim3(grid), dim3(256), 0, s,             \
                         (const uint4*)g_rows, (const uint4*)g_fm, fm_sum, (const uint4*)table, row_start, perm, \
                         V, N, gbs, padding_row, (uint4*)grad, long_rows, sink, fs);                             \
    TRS_SC_TAIL(HG, HF, false);                                                                                 \
  } while (0)
#define TRS_SCS(HG, CHF_)                                                                                       \
  do {                                                                                                          \
    hipLaunchKernelGGL((scatter_rows_group_kernel<T, LOG2L, HG, true, false, true, CHF_>), dim3(grid), dim3(256), 0, s, \
                       (const uint4*)g_rows, (const uint4*)g_fm, fm_sum, (const uint4*)table, row_start, perm,   \
                       V, N, gbs, padding_row, (uint4*)grad, long_rows, sink, fs);                               \
    TRS_SC_TAIL(HG, true, true);                                                                                \
  } while (0)
  const bool lean = scal && variant == 1 && (B * N) * (int64_t)L < ((int64_t)1 << 31) && V * (int64_t)L < ((int64_t)1 << 31) &&
                    N >= 2 && (B * N) * (int64_t)N < ((int64_t)1 << 32) && gbs == N;   // (mulhi by ceil(2^32/N) is exact)
  if (lean) {
    const unsigned rcpN = (unsigned)((((uint64_t)1 << 32) + (unsigned)N - 1) / (unsigned)N);
    if (hg)
      hipLaunchKernelGGL((scatter_rows_fm1_kernel<T, LOG2L, true>), dim3(grid), dim3(256), 0, s, (const uint4*)g_rows,
                         (const uint4*)tg, (const T*)fs.g1, (const uint4*)table, row_start, perm, (int)V, (unsigned)N,
                         rcpN, (int)padding_row, (uint4*)grad, long_rows, sink);
    else
      hipLaunchKernelGGL((scatter_rows_fm1_kernel<T, LOG2L, false>), dim3(grid), dim3(256), 0, s, (const uint4*)g_rows,
                         (const uint4*)tg, (const T*)fs.g1, (const uint4*)table, row_start, perm, (int)V, (unsigned)N,
                         rcpN, (int)padding_row, (uint4*)grad, long_rows, sink);
    if (hg) TRS_SC_TAIL(true, true, true);
    else TRS_SC_TAIL(false, true, true);
  } else if (scal) {
    if (hg) TRS_SCS(true, 2);
    else TRS_SCS(false, 2);
  } else if (hg && hf) TRS_SC(true, true);
  else if (hg) TRS_SC(true, false);
  else TRS_SC(false, true);
#undef TRS_SC
#undef TRS_SCS
#undef TRS_SC_TAIL
  if (g_first != nullptr)
    hipLaunchKernelGGL((scatter_first_long_kernel<T>), dim3(64), dim3(256), 0, s, (const T*)g_first, row_start, perm,
                       long_rows, (T*)grad_first);
}

template <typename T>
static int scatter_launch(const void* g_rows, const void* g_fm, const float* fm_sum, const void* table,
                          const int32_t* row_start, const int32_t* perm, int64_t V, int E, int N, int64_t gbs,
                          int64_t padding_row, void* grad, int32_t* long_rows, void* tg, float* scratch, int64_t B,
                          RowSink sink, hipStream_t s, int gcols, const void* g_first = nullptr,
                          void* grad_first = nullptr) {
  const int lg = log2_lanes_sc(E * (int)sizeof(T));
  const bool scal = gcols == 1 && E > 1 && fm_sum != nullptr;      // g_fm read as scalars: no alignment requirement
  const bool al = aligned16(g_rows) && (scal || aligned16(g_fm)) && aligned16(table) && aligned16(grad) && aligned16(fm_sum);
  if (lg >= 0 && al) {
    switch (lg) {
      case 0: scatter_group_launch<T, 0>(g_rows, g_fm, fm_sum, table, row_start, perm, V, N, gbs, padding_row, grad, long_rows, tg, scratch, B, sink, s, gcols, g_first, grad_first); break;
      case 1: scatter_group_launch<T, 1>(g_rows, g_fm, fm_sum, table, row_start, perm, V, N, gbs, padding_row, grad, long_rows, tg, scratch, B, sink, s, gcols, g_first, grad_first); break;
      case 2: scatter_group_launch<T, 2>(g_rows, g_fm, fm_sum, table, row_start, perm, V, N, gbs, padding_row, grad, long_rows, tg, scratch, B, sink, s, gcols, g_first, grad_first); break;
      case 3: scatter_group_launch<T, 3>(g_rows, g_fm, fm_sum, table, row_start, perm, V, N, gbs, padding_row, grad, long_rows, tg, scratch, B, sink, s, gcols, g_first, grad_first); break;
      case 4: scatter_group_launch<T, 4>(g_rows, g_fm, fm_sum, table, row_start, perm, V, N, gbs, padding_row, grad, long_rows, tg, scratch, B, sink, s, gcols, g_first, grad_first); break;
      case 5: scatter_group_launch<T, 5>(g_rows, g_fm, fm_sum, table, row_start, perm, V, N, gbs, padding_row, grad, long_rows, tg, scratch, B, sink, s, gcols, g_first, grad_first); break;
      default: scatter_group_launch<T, 6>(g_rows, g_fm, fm_sum, table, row_start, perm, V, N, gbs, padding_row, grad, long_rows, tg, scratch, B, sink, s, gcols, g_first, grad_first); break;
    }
  } else {
    if (g_first != nullptr) return 1;   // the companion table rides only in the 16-byte-vector walk
    // queue of the split rows: behind the plain queue of long rows (long_row_queue_bytes); its counter was zeroed with
    // the other one (scatter_rows_impl)
    int32_t* split_rows = long_rows + 1 + (B * N / LONG_ROW_ELEM + 2);
    hipLaunchKernelGGL((scatter_rows_elem_kernel<T>), dim3(stream_grid(V * E, 256, 256 * 32)), dim3(256), 0, s,
                       (const T*)g_rows, (const T*)g_fm, fm_sum, (const T*)table, row_start, perm, V, E, N, gbs,
                       padding_row, (T*)grad, long_rows, sink, gcols, split_rows, scratch);
    hipLaunchKernelGGL((scatter_long_rows_elem_kernel<T>), dim3(2048), dim3(256), 0, s, (const T*)g_rows,
                       (const T*)g_fm, fm_sum, (const T*)table, row_start, perm, E, N, gbs, (T*)grad, long_rows, sink,
                       gcols);
    hipLaunchKernelGGL((scatter_split_rows_elem_kernel<T>), dim3(512), dim3(256), 0, s, (const T*)g_rows,
                       (const T*)g_fm, fm_sum, row_start, perm, E, N, gbs, split_rows, scratch, gcols);
    hipLaunchKernelGGL((scatter_split_rows_finish_elem_kernel<T>), dim3(16), dim3(256), 0, s, (const T*)g_fm, fm_sum,
                       (const T*)table, E, (T*)grad, split_rows, scratch, sink);
  }
  return check_launch("scatter_rows");
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Zero-fill by a kernel, not hipMemsetAsync: a memset captured into a hipGraph becomes a memset NODE, and on this runtime
// the global-atomic bucket build replayed from a graph faulted (counters incremented on top of the previous replay's
// prefix sums -> positions past the end of perm) until the fills in front of atomically-updated counters were kernels.
__global__ __launch_bounds__(256) void zero_i32_kernel(int32_t* __restrict__ p, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0;
}
__global__ void zero_two_i32_kernel(int32_t* __restrict__ a, int32_t* __restrict__ b) {
  if (threadIdx.x == 0) { *a = 0; *b = 0; }
}
static int zero_i32(int32_t* p, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(zero_i32_kernel, dim3(stream_grid(n, 256, 1024)), dim3(256), 0, s, p, n);
  return 0;
}
__global__ __launch_bounds__(256) void zero2_i32_kernel(int32_t* __restrict__ p, int64_t n, int32_t* __restrict__ p2,
                                                        int64_t n2) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n + n2; i += stride) {
    if (i < n) p[i] = 0;
    else p2[i - n] = 0;
  }
}

// partitioned build: the LDS-counter count pass writes EVERY entry of row_start[0, V) itself, so the 4 MB zero fill of
// the counters (54 us inside the DeepFM step, squeezed in beside the fused MLP backward) is only needed when the build
// falls back to global atomics: a few words here (fall-back flags, the scan's status words, row_start[V]) ...
__global__ __launch_bounds__(256) void csr_zero_small_kernel(int32_t* __restrict__ flags, int32_t* __restrict__ status,
                                                            int nstatus, int32_t* __restrict__ last) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 64 + nstatus + 1; i += gridDim.x * blockDim.x) {
    if (i < 64) flags[i] = 0;
    else if (i < 64 + nstatus) status[i - 64] = 0;
    else *last = 0;
  }
}
// ... and the whole fill only behind a raised fall-back flag
__global__ __launch_bounds__(256) void zero_gated_i32_kernel(int32_t* __restrict__ p, int64_t n,
                                                            const int32_t* __restrict__ gate) {
  if (*gate == 0) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0;
}

}  // namespace trs

using namespace trs;

// workspace layout for csr_build: [slot: BN int32][tile_sums: ntiles int32][rowT: BN int32][flags: 256 B]
extern "C" size_t trs_csr_workspace_bytes(int64_t V, int64_t BN) {
  const size_t ntiles = (size_t)((V + 1 + SCAN_TILE - 1) / SCAN_TILE);
  return 2 * align_up((size_t)BN * 4, 256) + align_up((ntiles + 1) * 4, 256) + 512;   // + the scan's ticket word
}

extern "C" int trs_csr_build(const void* idx, int32_t idx_dtype, const int64_t* offsets, int64_t B, int32_t N,
                             int64_t V, int32_t* row_start, int32_t* perm, void* workspace, size_t ws_bytes,
                             int32_t* err_flag, trs_stream_t stream) {
  TRS_REQUIRE(row_start && workspace && (B == 0 || (idx && perm)), TRS_EINVAL, "csr_build: NULL pointer");
  TRS_REQUIRE(V > 0 && B >= 0 && N > 0, TRS_EINVAL, "csr_build: bad size");
  TRS_REQUIRE(idx_dtype == TRS_I64 || idx_dtype == TRS_I32, TRS_EDTYPE, "csr_build: idx dtype %d", idx_dtype);
  const int64_t BN = B * N;
  TRS_REQUIRE(BN < (int64_t)0x7fffffff && V < (int64_t)0x7ffffffe, TRS_ESHAPE,
              "csr_build: B*N and V must fit int32 (B*N=%lld V=%lld)", (long long)BN, (long long)V);
  TRS_REQUIRE(ws_bytes >= trs_csr_workspace_bytes(V, BN), TRS_EWORKSPACE, "csr_build: workspace %zu < %zu", ws_bytes,
              trs_csr_workspace_bytes(V, BN));
  hipStream_t s = (hipStream_t)stream;
  char* wsp = (char*)workspace;
  int32_t* slot = (int32_t*)wsp;
  wsp += align_up((size_t)BN * 4, 256);
  const int64_t n = V + 1;
  const int ntiles = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
  int32_t* tile_sums = (int32_t*)wsp;
  wsp += align_up((size_t)(ntiles + 1) * 4, 256);
  int32_t* rowT = (int32_t*)wsp;
  wsp += align_up((size_t)BN * 4, 256);
  int32_t* flags = (int32_t*)wsp;
  // partitioned (LDS-counter) build when the per-field ranges are few chunks each; otherwise global atomics
  // chunk size: enough (field, chunk) workgroups to cover the chip (each rescans its field's column of the batch,
  // so no more than ~16 per field on average), at most CSR2_CHUNK counters
  static const int64_t target_env = getenv("TRS_CSR_TARGET") ? atoll(getenv("TRS_CSR_TARGET")) : 0;      // (tuning: workgroups aimed at)
  const int64_t target = target_env > N ? target_env : std::max<int64_t>(256, 4 * (int64_t)N);
  int64_t chunk = (V + (target - N) - 1) / std::max<int64_t>(1, target - N);
  chunk = std::min<int64_t>(CSR2_CHUNK, std::max<int64_t>(1024, (chunk + 255) / 256 * 256));
  const int64_t max_items = (int64_t)N + (V + chunk - 1) / chunk;
  const bool part = offsets != nullptr && N <= CSR2_MAX_FIELDS && B >= 2048 && max_items <= 16 * (int64_t)N + 256 &&
                    max_items <= 16384;
  // one-pass scan: the status words hold 30-bit sums (B*N lookups in total) and are zeroed with the counters
  const bool onepass = (n + SCAN_TILE - 1) / SCAN_TILE <= 2048 && BN < ((int64_t)1 << 30);
  // TRS_CSR_LAZY_ZERO=1 (off by default): measured alternately on one box, DeepFM step 1.152-1.158 ms with the fill,
  // 1.159-1.166 without (profiles/r06_logs/ab_csr_lazy_zero.txt): the fill's 54 us inside the step were time spent WAITING
  // for wave slots beside the fused MLP backward, not work -- what replaces it on the side stream waits just the same
  static const bool lazy_zero = getenv("TRS_CSR_LAZY_ZERO") && getenv("TRS_CSR_LAZY_ZERO")[0] == '1';
  if (part && lazy_zero)
    hipLaunchKernelGGL(csr_zero_small_kernel, dim3(8), dim3(256), 0, s, flags, tile_sums, onepass ? ntiles + 1 : 0,
                       row_start + V);
  else
    hipLaunchKernelGGL(zero2_i32_kernel, dim3(stream_grid(n, 256, 1024)), dim3(256), 0, s, row_start, n, tile_sums,
                       (int64_t)(onepass ? ntiles + 1 : 0));
  const int32_t* gate = nullptr;
  if (part) {
    if (!lazy_zero) zero_i32(flags, 64, s);
    const int tiles = (int)((B + CSR2_TB - 1) / CSR2_TB);
    const size_t lds = (size_t)N * (CSR2_TB + 1) * 4;
    if (idx_dtype == TRS_I64)
      hipLaunchKernelGGL((csr2_rowid_kernel<int64_t>), dim3(tiles), dim3(256), lds, s, (const int64_t*)idx, offsets, B,
                         N, V, rowT, flags, err_flag, (int)max_items, (int)chunk, lazy_zero ? 1 : 0);
    else
      hipLaunchKernelGGL((csr2_rowid_kernel<int32_t>), dim3(tiles), dim3(256), lds, s, (const int32_t*)idx, offsets, B,
                         N, V, rowT, flags, err_flag, (int)max_items, (int)chunk, lazy_zero ? 1 : 0);
    hipLaunchKernelGGL((csr2_pass_kernel<false>), dim3((int)max_items), dim3(CSR2_THREADS), 0, s, rowT, offsets, B, N, V,
                       row_start, perm, flags, (int)chunk, 0);
    gate = flags;
    if (lazy_zero) hipLaunchKernelGGL(zero_gated_i32_kernel, dim3(256), dim3(256), 0, s, row_start, n, gate);
  }
  if (BN > 0) {
    // behind the partitioned build these kernels normally exit at once: a small grid keeps them off the CUs (the
    // grid-stride loops still cover every lookup when the fall-back flag is set)
    const int grid = part ? 256 : stream_grid(BN, 256, 256 * 16);
    if (idx_dtype == TRS_I64)
      hipLaunchKernelGGL((csr_count_kernel<int64_t>), dim3(grid), dim3(256), 0, s, (const int64_t*)idx, offsets, BN, N,
                         V, row_start, slot, err_flag, gate);
    else
      hipLaunchKernelGGL((csr_count_kernel<int32_t>), dim3(grid), dim3(256), 0, s, (const int32_t*)idx, offsets, BN, N,
                         V, row_start, slot, err_flag, gate);
  }
  if (onepass) {             // every tile's workgroup resident at once: one pass
    hipLaunchKernelGGL(scan_onepass_kernel, dim3(ntiles), dim3(SCAN_THREADS), 0, s, row_start, n, (unsigned*)tile_sums,
                       ntiles);
  } else {
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(ntiles), dim3(SCAN_THREADS), 0, s, row_start, n, tile_sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, tile_sums, ntiles);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(ntiles), dim3(SCAN_THREADS), 0, s, row_start, n, tile_sums);
  }
  if (part) {
    // stage for a chunk's piece of perm: what the LDS holds behind the 60 KB of counters (24 576 positions = 96 KB)
    static const int stage_cap = [] {
      const char* e = getenv("TRS_CSR_STAGE");
      if (e && e[0] == '0') return 0;
      return hipFuncSetAttribute((const void*)csr2_pass_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 CSR2_STAGE * 4) == hipSuccess ? CSR2_STAGE : 0;
    }();
    hipLaunchKernelGGL((csr2_pass_kernel<true>), dim3((int)max_items), dim3(CSR2_THREADS), (size_t)stage_cap * 4, s, rowT,
                       offsets, B, N, V, row_start, perm, flags, (int)chunk, stage_cap);
  }
  if (BN > 0) {
    // behind the partitioned build these kernels normally exit at once: a small grid keeps them off the CUs (the
    // grid-stride loops still cover every lookup when the fall-back flag is set)
    const int grid = part ? 256 : stream_grid(BN, 256, 256 * 16);
    if (idx_dtype == TRS_I64)
      hipLaunchKernelGGL((csr_fill_kernel<int64_t>), dim3(grid), dim3(256), 0, s, (const int64_t*)idx, offsets, BN, N,
                         row_start, slot, perm, gate);
    else
      hipLaunchKernelGGL((csr_fill_kernel<int32_t>), dim3(grid), dim3(256), 0, s, (const int32_t*)idx, offsets, BN, N,
                         row_start, slot, perm, gate);
  }
  return check_launch("csr_build");
}

static size_t long_row_entries(int64_t BN) { return (size_t)(BN / LONG_ROW + BN / LONG_CHUNK + 2); }
static size_t long_row_queue_bytes(int64_t BN) {
  // (row, chunk) pairs on the vector path, single row ids on the element path: room for the larger of the two
  // (+ on the element path the queue of (row, chunk) pairs of the rows split over several waves, behind the plain one)
  return align_up(std::max(long_row_entries(BN) * 8 + 8,
                           (size_t)(BN / LONG_ROW_ELEM + 3) * 4 + (size_t)(BN / ELEM_SPLIT + BN / LONG_ROW_ELEM / 64 + 2) * 8 + 8),
                  256);
}

extern "C" size_t trs_scatter_workspace_bytes(int64_t BN, int32_t N, int32_t E, int32_t dtype) {
  // [queue of hot rows + the counter][TG: (BN/N) x 2E values][chunk partials of very hot rows: entries x 2E fp32]
  const int64_t B = N > 0 ? (BN + N - 1) / N : 0;
  return long_row_queue_bytes(BN) + align_up((size_t)B * 2 * E * dtype_size(dtype), 256) +
         align_up(long_row_entries(BN) * 2 * E * 4, 256);
}

static int scatter_rows_impl(RowSink sink, const void* g_rows, int64_t g_rows_batch_stride, const void* g_fm,
                                int32_t g_fm_cols, const float* fm_sum, const void* table, const int32_t* row_start,
                                const int32_t* perm, int64_t BN, int64_t V, int32_t E, int32_t N, int32_t dtype,
                                int64_t padding_row, void* grad_table, void* workspace, size_t ws_bytes,
                                trs_stream_t stream, const void* g_first = nullptr, void* grad_first = nullptr) {
  TRS_REQUIRE(row_start && perm && grad_table && workspace, TRS_EINVAL, "scatter_rows: NULL pointer");
  TRS_REQUIRE(g_rows || g_fm, TRS_EINVAL, "scatter_rows: need g_rows and/or g_fm");
  TRS_REQUIRE((fm_sum == nullptr) || (g_fm && table), TRS_EINVAL, "scatter_rows: fm_sum needs g_fm and table");
  TRS_REQUIRE(g_fm == nullptr || g_fm_cols == E || (g_fm_cols == 1 && fm_sum != nullptr), TRS_EINVAL,
              "scatter_rows: g_fm_cols %d (E = %d full rows, or 1 = constant along E with fm_sum)", g_fm_cols, E);
  const int64_t gbs = g_rows_batch_stride > 0 ? g_rows_batch_stride : N;
  TRS_REQUIRE(gbs >= N, TRS_EINVAL, "scatter_rows: g_rows_batch_stride %lld < N", (long long)gbs);
  TRS_REQUIRE(V > 0 && E > 0 && N > 0 && BN >= 0, TRS_EINVAL, "scatter_rows: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "scatter_rows: dtype %d", dtype);
  TRS_REQUIRE(ws_bytes >= trs_scatter_workspace_bytes(BN, N, E, dtype), TRS_EWORKSPACE,
              "scatter_rows: workspace %zu < %zu", ws_bytes, trs_scatter_workspace_bytes(BN, N, E, dtype));
  TRS_REQUIRE(BN % N == 0, TRS_EINVAL, "scatter_rows: B*N=%lld not a multiple of N=%d", (long long)BN, N);
  void* tg = (char*)workspace + long_row_queue_bytes(BN);
  const int64_t B = BN / N;
  float* scratch = (float*)((char*)tg + align_up((size_t)B * 2 * E * dtype_size(dtype), 256));
  hipStream_t s = (hipStream_t)stream;
  int32_t* long_rows = (int32_t*)workspace;
  // the counters of the hot-row queue and (element path) of the split-row queue behind it
  hipLaunchKernelGGL(zero_two_i32_kernel, dim3(1), dim3(64), 0, s, long_rows, long_rows + 1 + (BN / LONG_ROW_ELEM + 2));
  int rc;
  if (dtype == TRS_F32)
    rc = scatter_launch<float>(g_rows, g_fm, fm_sum, table, row_start, perm, V, E, N, gbs, padding_row, grad_table,
                               long_rows, tg, scratch, B, sink, s, g_fm_cols, g_first, grad_first);
  else
    rc = scatter_launch<bf16_t>(g_rows, g_fm, fm_sum, table, row_start, perm, V, E, N, gbs, padding_row, grad_table,
                                long_rows, tg, scratch, B, sink, s, g_fm_cols, g_first, grad_first);
  if (rc == 1) return fail(TRS_ESHAPE, "scatter_rows_first: rows must be whole 16-byte vectors (E*sizeof %% 16 == 0)");
  return rc;
}

/* see include/trs_abi.h: the dense gradients of an embedding table AND of its first-order (E = 1) companion in one walk */
extern "C" int trs_scatter_rows_first(const void* g_rows, int64_t g_rows_batch_stride, const void* g_fm, int32_t g_fm_cols,
                                      const float* fm_sum, const void* table, const int32_t* row_start,
                                      const int32_t* perm, int64_t BN, int64_t V, int32_t E, int32_t N, int32_t dtype,
                                      int64_t padding_row, void* grad_table, const void* g_first, void* grad_first,
                                      void* workspace, size_t ws_bytes, trs_stream_t stream) {
  TRS_REQUIRE(g_first && grad_first, TRS_EINVAL, "scatter_rows_first: NULL first-order pointer");
  return scatter_rows_impl(RowSink{0, 0.f, 0.f, nullptr}, g_rows, g_rows_batch_stride, g_fm, g_fm_cols, fm_sum, table, row_start,
                           perm, BN, V, E, N, dtype, padding_row, grad_table, workspace, ws_bytes, stream, g_first,
                           grad_first);
}

extern "C" int trs_scatter_rows(const void* g_rows, int64_t g_rows_batch_stride, const void* g_fm, int32_t g_fm_cols,
                                const float* fm_sum,
                                const void* table, const int32_t* row_start, const int32_t* perm, int64_t BN, int64_t V,
                                int32_t E, int32_t N, int32_t dtype, int64_t padding_row, void* grad_table,
                                void* workspace, size_t ws_bytes, trs_stream_t stream) {
  return scatter_rows_impl(RowSink{0, 0.f, 0.f, nullptr}, g_rows, g_rows_batch_stride, g_fm, g_fm_cols, fm_sum, table, row_start,
                           perm, BN, V, E, N, dtype, padding_row, grad_table, workspace, ws_bytes, stream);
}

extern "C" int trs_scatter_rows_update(const void* g_rows, int64_t g_rows_batch_stride, const void* g_fm, int32_t g_fm_cols,
                                       const float* fm_sum, void* table, const int32_t* row_start, const int32_t* perm,
                                       int64_t BN, int64_t V, int32_t E, int32_t N, int32_t dtype, int64_t padding_row,
                                       int32_t optimizer, float lr, float eps, float* state, void* workspace,
                                       size_t ws_bytes, trs_stream_t stream) {
  TRS_REQUIRE(table, TRS_EINVAL, "scatter_rows_update: NULL table");
  TRS_REQUIRE(optimizer == 1 || optimizer == 2, TRS_EINVAL, "scatter_rows_update: optimizer %d (1 = SGD, 2 = Adagrad)",
              optimizer);
  TRS_REQUIRE(optimizer == 1 || state != nullptr, TRS_EINVAL, "scatter_rows_update: Adagrad needs the state buffer");
  return scatter_rows_impl(RowSink{optimizer, lr, eps, state}, g_rows, g_rows_batch_stride, g_fm, g_fm_cols, fm_sum, table,
                           row_start, perm, BN, V, E, N, dtype, padding_row, table, workspace, ws_bytes, stream);
}

extern "C" int trs_scatter_rows_update_adam(const void* g_rows, int64_t g_rows_batch_stride, const void* g_fm, int32_t g_fm_cols,
                                            const float* fm_sum, void* table, const int32_t* row_start,
                                            const int32_t* perm, int64_t BN, int64_t V, int32_t E, int32_t N,
                                            int32_t dtype, int64_t padding_row, float step_size, float beta1,
                                            float beta2, float eps, float* exp_avg, float* exp_avg_sq, void* workspace,
                                            size_t ws_bytes, trs_stream_t stream) {
  TRS_REQUIRE(table && exp_avg && exp_avg_sq, TRS_EINVAL, "scatter_rows_update_adam: NULL table / moment buffer");
  TRS_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, TRS_EINVAL,
              "scatter_rows_update_adam: betas (%g, %g) must be in [0, 1)", (double)beta1, (double)beta2);
  RowSink sink{3, step_size, eps, exp_avg};
  sink.beta1 = beta1;
  sink.beta2 = beta2;
  sink.state2 = exp_avg_sq;
  return scatter_rows_impl(sink, g_rows, g_rows_batch_stride, g_fm, g_fm_cols, fm_sum, table, row_start, perm, BN, V, E, N, dtype,
                           padding_row, table, workspace, ws_bytes, stream);
}

/* see include/trs_abi.h: the fused sparse optimizer step on a COMPACT list of distinct rows */
extern "C" int trs_scatter_rows_update_mapped(const void* g_rows, void* table, const int32_t* row_map,
                                              const int32_t* row_start, const int32_t* perm, int64_t K, int64_t U,
                                              int64_t V, int32_t E, int32_t dtype, int32_t optimizer, float lr, float eps,
                                              float beta1, float beta2, float* state, float* state2, void* workspace,
                                              size_t ws_bytes, trs_stream_t stream) {
  TRS_REQUIRE(g_rows && table && row_map, TRS_EINVAL, "scatter_rows_update_mapped: NULL pointer");
  TRS_REQUIRE(optimizer >= 1 && optimizer <= 3, TRS_EINVAL, "scatter_rows_update_mapped: optimizer %d", optimizer);
  TRS_REQUIRE(optimizer == 1 || state != nullptr, TRS_EINVAL, "scatter_rows_update_mapped: missing optimizer state");
  TRS_REQUIRE(optimizer != 3 || state2 != nullptr, TRS_EINVAL, "scatter_rows_update_mapped: Adam needs both moments");
  TRS_REQUIRE(U >= 0 && V > 0 && K >= 0, TRS_EINVAL, "scatter_rows_update_mapped: bad row counts");
  if (U == 0 || K == 0) return TRS_OK;      // a rank that received no lookups this step: nothing to update
  RowSink sink{optimizer, lr, eps, state};
  sink.map_rows = V;
  sink.beta1 = beta1;
  sink.beta2 = beta2;
  sink.state2 = state2;
  sink.row_map = row_map;
  return scatter_rows_impl(sink, g_rows, 0, nullptr, 0, nullptr, table, row_start, perm, K, U, E, 1, dtype, -1, table,
                           workspace, ws_bytes, stream);
}
