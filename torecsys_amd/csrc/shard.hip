// Row-sharded tables (multi-GPU lookup): bucket the B*N global row ids of the local batch by owner rank.
// counting sort with W buckets: per-workgroup LDS histograms, one global atomic per (workgroup, owner).
#include "trs_common.hpp"

namespace trs {

constexpr int MAX_WORLD = 256;
constexpr int CHUNK = 4096;  // positions per workgroup in the fill pass

// owner of global row r: floor(r / per) clamped into [0, W) without a 64-bit division (~80 emulated instructions per
// lookup: it was most of this pass's 23 + 30 us): a float-reciprocal estimate, exact after one step either way
// (r < 2^40 and W <= 256 keep the estimate within one of the quotient)
__device__ __forceinline__ int owner_of(int64_t r, int64_t per, float inv_per, int W) {
  if (r <= 0) return 0;
  int w = (int)((float)r * inv_per);
  w = w >= W ? W - 1 : w;
  if (r < (int64_t)w * per) --w;
  else if (w + 1 < W && r >= (int64_t)(w + 1) * per) ++w;
  return w;
}
// field of flat lookup p = p % N, same trick (exact for p < 2^24, the plain remainder beyond)
__device__ __forceinline__ int field_of(int64_t p, int N, float inv_n) {
  if (p >= (1 << 24)) return (int)(p % N);
  const int q = (int)((float)(int)p * inv_n);
  int rem = (int)p - q * N;
  if (rem < 0) rem += N;
  else if (rem >= N) rem -= N;
  return rem;
}

// hist[w] += 1 for every active lane, returning the value each lane's own atomic would have returned -- with ONE LDS
// atomic per distinct owner in the wave instead of one per lane: the lanes of a wave hit at most `world` different
// counters (a single one on a one-rank group), and an LDS atomic on one address retires about one lane per clock.  The
// loop runs once per distinct owner among the wave's lanes; every ballot / shuffle is executed by the whole wave.
__device__ __forceinline__ int wave_agg_inc(int* hist, int w, bool active) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(active);
  int res = 0;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int w0 = __shfl(w, leader, 64);
    const bool mine = active && w == w0;
    const unsigned long long m = __ballot(mine);
    int base = 0;
    if (lane == leader) base = atomicAdd(&hist[w0], __popcll(m));
    base = __shfl(base, leader, 64);
    if (mine) res = base + __popcll(m & ((1ull << lane) - 1ull));
    todo &= ~m;
  }
  return res;
}

template <typename IdxT>
__global__ __launch_bounds__(256) void bucket_count_kernel(const IdxT* __restrict__ idx,
                                                           const int64_t* __restrict__ offsets, int64_t BN, int N,
                                                           int64_t per, int W, unsigned long long* __restrict__ counts) {
  __shared__ int hist[MAX_WORLD];
  const float inv_per = 1.0f / (float)per, inv_n = 1.0f / (float)N;
  for (int w = threadIdx.x; w < W; w += blockDim.x) hist[w] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x; p0 < BN; p0 += stride) {      // (uniform trip count per wave)
    const int64_t p = p0 + threadIdx.x;
    const bool active = p < BN;
    int w = 0;
    if (active) {
      const int64_t r = load_row_id(idx, offsets, p, field_of(p, N, inv_n));
      w = owner_of(r, per, inv_per, W);
    }
    wave_agg_inc(hist, w, active);
  }
  __syncthreads();
  for (int w = threadIdx.x; w < W; w += blockDim.x)
    if (hist[w]) atomicAdd(&counts[w], (unsigned long long)hist[w]);
}

template <typename IdxT>
__global__ __launch_bounds__(256) void bucket_fill_kernel(const IdxT* __restrict__ idx,
                                                          const int64_t* __restrict__ offsets, int64_t BN, int N,
                                                          int64_t per, int W, const unsigned long long* __restrict__ counts,
                                                          unsigned long long* __restrict__ cursor,
                                                          int32_t* __restrict__ send_ids, int32_t* __restrict__ send_pos,
                                                          int32_t* __restrict__ inv_pos) {
  __shared__ int hist[MAX_WORLD];
  __shared__ long long base[MAX_WORLD];
  const float inv_per = 1.0f / (float)per, inv_n = 1.0f / (float)N;
  const int64_t p0 = (int64_t)blockIdx.x * CHUNK;
  const int64_t p1 = p0 + CHUNK < BN ? p0 + CHUNK : BN;
  for (int w = threadIdx.x; w < W; w += blockDim.x) hist[w] = 0;
  __syncthreads();
  for (int64_t q0 = p0; q0 < p1; q0 += blockDim.x) {
    const int64_t p = q0 + threadIdx.x;
    const bool active = p < p1;
    int w = 0;
    if (active) {
      const int64_t r = load_row_id(idx, offsets, p, field_of(p, N, inv_n));
      w = owner_of(r, per, inv_per, W);
    }
    wave_agg_inc(hist, w, active);
  }
  __syncthreads();
  if (threadIdx.x < W) {
    const int w = threadIdx.x;
    long long pre = 0;
    for (int k = 0; k < w; ++k) pre += (long long)counts[k];
    base[w] = pre + (hist[w] ? (long long)atomicAdd(&cursor[w], (unsigned long long)hist[w]) : 0);
  }
  __syncthreads();
  for (int w = threadIdx.x; w < W; w += blockDim.x) hist[w] = 0;
  __syncthreads();
  for (int64_t q0 = p0; q0 < p1; q0 += blockDim.x) {
    const int64_t p = q0 + threadIdx.x;
    const bool active = p < p1;
    int w = 0;
    int64_t r = 0;
    if (active) {
      r = load_row_id(idx, offsets, p, field_of(p, N, inv_n));
      w = owner_of(r, per, inv_per, W);
    }
    const int k = wave_agg_inc(hist, w, active);
    if (active) {
      const long long slot = base[w] + k;
      send_ids[slot] = (int32_t)(r - (int64_t)w * per);
      send_pos[slot] = (int32_t)p;
      if (inv_pos) inv_pos[p] = (int32_t)slot;
    }
  }
}

// counts / cursor start at zero.  A kernel, not hipMemsetAsync: inside a captured hipGraph the 8-byte memset nodes were
// seen racing with the kernels behind them (the fill pass then computes slots from garbage and writes out of bounds)
__global__ void bucket_zero_kernel(unsigned long long* __restrict__ a, unsigned long long* __restrict__ b, int W) {
  for (int w = threadIdx.x; w < W; w += blockDim.x) { a[w] = 0ull; b[w] = 0ull; }
}

static size_t align_up_s(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- un-permute of the sharded lookup with the LOCALLY-OWNED lookups read straight from this rank's shard ------------
// The requester's lookups are numbered in exchange order (slot s = inv_pos[p], grouped by owner).  The slots
// [self_lo, self_lo + self_n) are the ones this rank owns itself: their rows never travel (no owner-side gather into a send
// buffer, no all-to-all copy, no second read out of the receive buffer) -- the row is local[send_ids[s]].  Every other slot
// reads the received rows, which are stored WITHOUT the self segment: back[s - (s >= self_lo + self_n ? self_n : 0)].
// A local id outside [0, n_valid) (a global id past the table, the short last shard) reads as a zero row and raises
// err_flag, exactly as the owner-side gather does for ids that arrive over the wire.
// Same lane layout as embed_fm_group_kernel (fm.hip): L = E*s/16 lanes own a sample, four rows in flight, running sum and
// sum of squares of the lane's own columns in registers; writes the (B,N,E) block, FM second order and the fp32 field sum.
template <typename T, int LOG2L, bool STREAM>
__global__ __launch_bounds__(256) void embed_fm_sharded_group_kernel(
    const uint4* __restrict__ back, const uint4* __restrict__ local, const int32_t* __restrict__ inv_pos,
    const int32_t* __restrict__ send_ids, int32_t self_lo, int32_t self_n, int64_t n_valid, int64_t B, int N,
    uint4* __restrict__ emb, uint4* __restrict__ fm, float* __restrict__ fm_sum, int32_t* __restrict__ err_flag) {
  constexpr int L = 1 << LOG2L;
  constexpr int VE = Vec16<T>::VE;
  constexpr int CH = 4;
  const int lane_v = threadIdx.x & (L - 1);
  const int32_t self_hi = self_lo + self_n;
  const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> LOG2L;
  for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> LOG2L; b < B; b += groups) {
    float sm[VE], q[VE];
#pragma unroll
    for (int k = 0; k < VE; ++k) { sm[k] = 0.f; q[k] = 0.f; }
    for (int n0 = 0; n0 < N; n0 += CH) {
      int32_t sl[CH];
      int64_t r[CH];       // >= 0: row of `back`;  <= -2: local row -2 - r;  -1: nothing to read (zero row)
      uint4 v[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) sl[c] = n0 + c < N ? inv_pos[b * N + n0 + c] : -1;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        r[c] = -1;
        if (n0 + c < N && sl[c] >= 0) {
          if (sl[c] >= self_lo && sl[c] < self_hi) {
            const int64_t id = send_ids[sl[c]];
            if (id >= 0 && id < n_valid) r[c] = -2 - id;
            else if (err_flag != nullptr) *err_flag = 1;
          } else {
            r[c] = sl[c] - (sl[c] >= self_hi ? self_n : 0);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        v[c] = make_uint4(0, 0, 0, 0);
        if (r[c] >= 0) v[c] = back[r[c] * L + lane_v];
        else if (r[c] <= -2) v[c] = STREAM ? load_stream(&local[(-2 - r[c]) * L + lane_v]) : local[(-2 - r[c]) * L + lane_v];
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int n = n0 + c;
        if (n < N) {
          float x[VE];
          Vec16<T>::unpack(v[c], x);
#pragma unroll
          for (int k = 0; k < VE; ++k) { sm[k] += x[k]; q[k] = fmaf(x[k], x[k], q[k]); }
          if (emb != nullptr) store_stream(&emb[(b * N + n) * L + lane_v], v[c]);
        }
      }
    }
    if (fm != nullptr) {
      float o[VE];
#pragma unroll
      for (int k = 0; k < VE; ++k) o[k] = 0.5f * (sm[k] * sm[k] - q[k]);
      fm[b * L + lane_v] = Vec16<T>::pack(o);
    }
    if (fm_sum != nullptr) {
      float4* dst = reinterpret_cast<float4*>(fm_sum + (b * L + lane_v) * VE);
#pragma unroll
      for (int k = 0; k < VE; k += 4) dst[k / 4] = make_float4(sm[k], sm[k + 1], sm[k + 2], sm[k + 3]);
    }
  }
}

// any E (the E = 1 first-order table): one thread per element of the block; FM outputs (rare at such widths) by a
// per-(b, e) loop over the fields
template <typename T>
__global__ __launch_bounds__(256) void gather_sharded_elem_kernel(
    const T* __restrict__ back, const T* __restrict__ local, const int32_t* __restrict__ inv_pos,
    const int32_t* __restrict__ send_ids, int32_t self_lo, int32_t self_n, int64_t n_valid, int64_t BN, int E,
    T* __restrict__ emb, int32_t* __restrict__ err_flag) {
  const int64_t total = BN * E, stride = (int64_t)gridDim.x * blockDim.x;
  const int32_t self_hi = self_lo + self_n;
  const bool f32 = total < ((int64_t)1 << 32);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t p = udiv_fast(t, E, f32);
    const int e = (int)(t - p * E);
    const int32_t sl = inv_pos[p];
    T val = T{};
    if (sl >= 0) {
      if (sl >= self_lo && sl < self_hi) {
        const int64_t id = send_ids[sl];
        if (id >= 0 && id < n_valid) val = local[id * E + e];
        else if (err_flag != nullptr) *err_flag = 1;
      } else {
        val = back[(int64_t)(sl - (sl >= self_hi ? self_n : 0)) * E + e];
      }
    }
    emb[t] = val;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void fm_of_block_elem_kernel(const T* __restrict__ x, int64_t B, int N, int E,
                                                               T* __restrict__ fm, float* __restrict__ fm_sum) {
  const int64_t total = B * E, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / E;
    const int e = (int)(t - b * E);
    float s = 0.f, q = 0.f;
    for (int n = 0; n < N; ++n) {
      const float v = to_f32(x[(b * N + n) * E + e]);
      s += v;
      q = fmaf(v, v, q);
    }
    if (fm != nullptr) fm[t] = from_f32<T>(0.5f * (s * s - q));
    if (fm_sum != nullptr) fm_sum[t] = s;
  }
}

template <typename T>
static int embed_fm_sharded_launch(const void* back, const void* local, const int32_t* inv_pos, const int32_t* send_ids,
                                   int32_t self_lo, int32_t self_n, int64_t n_valid, int64_t local_rows, int64_t B, int N,
                                   int E, void* emb, void* fm, float* fm_sum, int32_t* err_flag, hipStream_t s) {
  const int rb = E * (int)sizeof(T);
  int lg = -1;
  if (rb % 16 == 0 && is_pow2(rb / 16) && rb / 16 <= 64) {
    lg = 0;
    while ((1 << lg) < rb / 16) ++lg;
  }
  const bool al = aligned16(back) && aligned16(local) && aligned16(emb) && aligned16(fm) && aligned16(fm_sum);
  if (lg >= 0 && al) {
    const int L = 1 << lg;
    const int grid = stream_grid(B * L, 256, 256 * 16);
    const bool stream = (size_t)local_rows * rb > ((size_t)512 << 20);      // as embed_fm: a table far beyond the caches
#define TRS_SH2(LG, ST)                                                                                               \
  hipLaunchKernelGGL((embed_fm_sharded_group_kernel<T, LG, ST>), dim3(grid), dim3(256), 0, s, (const uint4*)back,     \
                     (const uint4*)local, inv_pos, send_ids, self_lo, self_n, n_valid, B, N, (uint4*)emb, (uint4*)fm, \
                     fm_sum, err_flag)
#define TRS_SH(LG)      \
  if (stream) {         \
    TRS_SH2(LG, true);  \
  } else {              \
    TRS_SH2(LG, false); \
  }
    switch (lg) {
      case 0: TRS_SH(0); break;
      case 1: TRS_SH(1); break;
      case 2: TRS_SH(2); break;
      case 3: TRS_SH(3); break;
      case 4: TRS_SH(4); break;
      case 5: TRS_SH(5); break;
      default: TRS_SH(6); break;
    }
#undef TRS_SH
#undef TRS_SH2
  } else {
    if (emb == nullptr) return fail(TRS_EINVAL, "embed_fm_sharded: rows that are not whole 16-byte vectors need the block");
    hipLaunchKernelGGL((gather_sharded_elem_kernel<T>), dim3(stream_grid(B * N * E, 256, 256 * 32)), dim3(256), 0, s,
                       (const T*)back, (const T*)local, inv_pos, send_ids, self_lo, self_n, n_valid, B * N, E, (T*)emb,
                       err_flag);
    if (fm != nullptr || fm_sum != nullptr)
      hipLaunchKernelGGL((fm_of_block_elem_kernel<T>), dim3(stream_grid(B * E, 256, 256 * 16)), dim3(256), 0, s,
                         (const T*)emb, B, N, E, (T*)fm, fm_sum);
  }
  return check_launch("embed_fm_sharded");
}

}  // namespace trs

using namespace trs;

extern "C" size_t trs_bucket_workspace_bytes(int64_t BN, int32_t world) {
  (void)BN;
  return align_up_s((size_t)world * 8, 256);
}

extern "C" int trs_bucket_by_owner(const void* idx, int32_t idx_dtype, const int64_t* offsets, int64_t B, int32_t N,
                                   int64_t rows_per_rank, int32_t world, int64_t* counts, int32_t* send_ids,
                                   int32_t* send_pos, int32_t* inv_pos, void* workspace, size_t ws_bytes,
                                   trs_stream_t stream) {
  TRS_REQUIRE(counts && workspace, TRS_EINVAL, "bucket_by_owner: NULL pointer");
  TRS_REQUIRE(B >= 0 && N > 0 && rows_per_rank > 0 && world > 0, TRS_EINVAL, "bucket_by_owner: bad size");
  TRS_REQUIRE(world <= MAX_WORLD, TRS_ESHAPE, "bucket_by_owner: world %d > %d", world, MAX_WORLD);
  TRS_REQUIRE(rows_per_rank < (int64_t)0x7fffffff, TRS_ESHAPE, "bucket_by_owner: rows_per_rank must fit int32");
  TRS_REQUIRE(idx_dtype == TRS_I64 || idx_dtype == TRS_I32, TRS_EDTYPE, "bucket_by_owner: idx dtype %d", idx_dtype);
  TRS_REQUIRE(ws_bytes >= trs_bucket_workspace_bytes(B * N, world), TRS_EWORKSPACE, "bucket_by_owner: workspace");
  hipStream_t s = (hipStream_t)stream;
  const int64_t BN = B * N;
  TRS_REQUIRE(BN < (int64_t)0x7fffffff, TRS_ESHAPE, "bucket_by_owner: B*N must fit int32");
  hipLaunchKernelGGL(bucket_zero_kernel, dim3(1), dim3(64), 0, s, (unsigned long long*)counts,
                     (unsigned long long*)workspace, world);
  if (BN == 0) return check_launch("bucket_by_owner");
  TRS_REQUIRE(idx && send_ids && send_pos, TRS_EINVAL, "bucket_by_owner: NULL pointer");
  unsigned long long* cnt = (unsigned long long*)counts;
  unsigned long long* cursor = (unsigned long long*)workspace;
  const int g1 = stream_grid(BN, 256, 1024);
  const int g2 = (int)((BN + CHUNK - 1) / CHUNK);
  if (idx_dtype == TRS_I64) {
    hipLaunchKernelGGL((bucket_count_kernel<int64_t>), dim3(g1), dim3(256), 0, s, (const int64_t*)idx, offsets, BN, N,
                       rows_per_rank, world, cnt);
    hipLaunchKernelGGL((bucket_fill_kernel<int64_t>), dim3(g2), dim3(256), 0, s, (const int64_t*)idx, offsets, BN, N,
                       rows_per_rank, world, cnt, cursor, send_ids, send_pos, inv_pos);
  } else {
    hipLaunchKernelGGL((bucket_count_kernel<int32_t>), dim3(g1), dim3(256), 0, s, (const int32_t*)idx, offsets, BN, N,
                       rows_per_rank, world, cnt);
    hipLaunchKernelGGL((bucket_fill_kernel<int32_t>), dim3(g2), dim3(256), 0, s, (const int32_t*)idx, offsets, BN, N,
                       rows_per_rank, world, cnt, cursor, send_ids, send_pos, inv_pos);
  }
  return check_launch("bucket_by_owner");
}

/* see include/trs_abi.h */
extern "C" int trs_embed_fm_sharded(const void* back, int64_t back_rows, const void* local, int64_t local_rows,
                                    int64_t n_valid, int32_t E, int32_t dtype, const int32_t* inv_pos,
                                    const int32_t* send_ids, int32_t self_lo, int32_t self_n, int64_t B, int32_t N,
                                    void* emb, void* fm, float* fm_sum, int32_t* err_flag, trs_stream_t stream) {
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(inv_pos && (emb || fm || fm_sum), TRS_EINVAL, "embed_fm_sharded: NULL pointer");
  TRS_REQUIRE(self_n == 0 || (local && send_ids), TRS_EINVAL, "embed_fm_sharded: a self segment needs the shard and send_ids");
  TRS_REQUIRE(back_rows == 0 || back, TRS_EINVAL, "embed_fm_sharded: NULL receive buffer");
  TRS_REQUIRE(E > 0 && N > 0 && B > 0 && self_lo >= 0 && self_n >= 0 && back_rows >= 0 && local_rows >= 0 &&
                  n_valid >= 0 && n_valid <= local_rows,
              TRS_EINVAL, "embed_fm_sharded: bad size");
  TRS_REQUIRE(B * (int64_t)N < (int64_t)0x7fffffff, TRS_ESHAPE, "embed_fm_sharded: B*N must fit int32");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "embed_fm_sharded: dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TRS_F32)
    return embed_fm_sharded_launch<float>(back, local, inv_pos, send_ids, self_lo, self_n, n_valid, local_rows, B, N, E, emb,
                                          fm, fm_sum, err_flag, s);
  return embed_fm_sharded_launch<bf16_t>(back, local, inv_pos, send_ids, self_lo, self_n, n_valid, local_rows, B, N, E, emb,
                                         fm, fm_sum, err_flag, s);
}
