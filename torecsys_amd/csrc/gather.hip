// K1: embedding row gather (bit-exact copy), field-aware gather (I3), permute helpers.
// HBM-bound: every lane moves one 16-byte vector of a table row; a 128-byte bf16 row (E=64) is
// fetched by 8 adjacent lanes = one full cache line per row, one wave instruction = 8 rows.
#include <algorithm>

#include "trs_common.hpp"

namespace trs {

static thread_local char g_err[512];
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TRS_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return TRS_OK;
}

// Zero-fill by a KERNEL (never hipMemsetAsync): captured into a hipGraph a memset becomes a memset NODE, and on this
// runtime such nodes did not reliably re-zero their target in replays (scatter.hip, shard.hip found it the hard way).
// Every fill the library enqueues goes through here.  ``bytes`` is a multiple of 2; 16-byte stores where p allows.
__global__ __launch_bounds__(256) void zero_bytes_kernel(char* __restrict__ p, size_t head, size_t nvec, size_t tail) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4* v = reinterpret_cast<uint4*>(p + head);
  for (size_t i = t0; i < nvec; i += stride) v[i] = make_uint4(0, 0, 0, 0);
  for (size_t i = t0; i < head; i += stride) p[i] = 0;
  for (size_t i = t0; i < tail; i += stride) p[head + nvec * 16 + i] = 0;
}
int zero_bytes(void* p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return TRS_OK;
  const size_t mis = (size_t)(reinterpret_cast<uintptr_t>(p) & 15u);
  const size_t head = std::min(bytes, mis ? 16 - mis : (size_t)0);
  const size_t nvec = (bytes - head) / 16, tail = bytes - head - nvec * 16;
  hipLaunchKernelGGL(zero_bytes_kernel, dim3(stream_grid((int64_t)std::max<size_t>(nvec, 16), 256, 2048)), dim3(256), 0, s,
                     (char*)p, head, nvec, tail);
  return check_launch("zero_bytes");
}

// rows = B*N flat positions; vpr = 16-byte vectors per row (E*sizeof(T)/16).
// SHIFT >= 0: vpr == 1<<SHIFT (no integer division); SHIFT < 0: generic.
// ``stream`` (uniform): the table is far larger than the caches (> 512 MiB, e.g. a 16 GB shard of a row-sharded table):
// rows are fetched with streaming loads, as embed_fm does (fm.hip)
template <typename IdxT, int SHIFT, int UNROLL>
__global__ __launch_bounds__(256) void gather_rows_vec_kernel(
    const uint4* __restrict__ table, const IdxT* __restrict__ idx, const int64_t* __restrict__ offsets,
    uint4* __restrict__ out, int64_t total_vecs, int vpr, int N, int64_t V, int32_t* __restrict__ err_flag, bool stream) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; t0 < total_vecs; t0 += stride * UNROLL) {
    uint4 v[UNROLL];
    bool ok[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t t = t0 + u * stride;
      ok[u] = t < total_vecs;
      v[u] = make_uint4(0, 0, 0, 0);
      if (ok[u]) {
        int64_t p;
        int lane_v;
        if (SHIFT >= 0) {
          p = t >> SHIFT;
          lane_v = (int)(t & ((1 << SHIFT) - 1));
        } else {
          p = t / vpr;
          lane_v = (int)(t - p * vpr);
        }
        // flat positions fit 32 bits (checked by trs_gather_rows): a 32-bit remainder instead of the emulated 64-bit one
        const int n = (int)((unsigned)p % (unsigned)N);
        const int64_t r = load_row_id(idx, offsets, p, n);
        if (err_flag != nullptr && (r < 0 || r >= V)) {
          *err_flag = 1;
        } else {
          v[u] = stream ? load_stream(&table[r * vpr + lane_v]) : table[r * vpr + lane_v];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t t = t0 + u * stride;
      if (ok[u]) {
        // streaming store: the gathered block is consumed by a later kernel, keep L2 for the table rows
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        const u32x4 w = {v[u].x, v[u].y, v[u].z, v[u].w};
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(&out[t]));
      }
    }
  }
}

// generic element-wise path (row bytes not a multiple of 16, e.g. the E=1 first-order table)
template <typename T, typename IdxT>
__global__ __launch_bounds__(256) void gather_rows_elem_kernel(
    const T* __restrict__ table, const IdxT* __restrict__ idx, const int64_t* __restrict__ offsets,
    T* __restrict__ out, int64_t total, int E, int N, int64_t V, int32_t* __restrict__ err_flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool f32 = total < ((int64_t)1 << 32);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t p = udiv_fast(t, E, f32);
    const int e = (int)(t - p * E);
    const int n = (int)(p - udiv_fast(p, N, f32) * N);
    const int64_t r = load_row_id(idx, offsets, p, n);
    if (err_flag != nullptr && (r < 0 || r >= V)) {
      *err_flag = 1;
      out[t] = T{};
    } else {
      out[t] = table[r * E + e];
    }
  }
}

template <typename IdxT>
static int launch_gather_vec(const void* table, const IdxT* idx, const int64_t* offsets, void* out,
                             int64_t rows, int vpr, int N, int64_t V, int32_t* err_flag, hipStream_t s) {
  const int64_t total = rows * vpr;
  constexpr int UNROLL = 4;
  const int grid = stream_grid((total + UNROLL - 1) / UNROLL, 256, 256 * 32);
#define TRS_GV(SH)                                                                                  \
  hipLaunchKernelGGL((gather_rows_vec_kernel<IdxT, SH, UNROLL>), dim3(grid), dim3(256), 0, s,       \
                     (const uint4*)table, idx, offsets, (uint4*)out, total, vpr, N, V, err_flag,  \
                     (size_t)V * vpr * 16 > ((size_t)512 << 20))
  switch (vpr) {
    case 1: TRS_GV(0); break;
    case 2: TRS_GV(1); break;
    case 4: TRS_GV(2); break;
    case 8: TRS_GV(3); break;
    case 16: TRS_GV(4); break;
    case 32: TRS_GV(5); break;
    case 64: TRS_GV(6); break;
    default: TRS_GV(-1); break;
  }
#undef TRS_GV
  return check_launch("gather_rows");
}

template <typename T, typename IdxT>
static int launch_gather_elem(const void* table, const IdxT* idx, const int64_t* offsets, void* out,
                              int64_t rows, int E, int N, int64_t V, int32_t* err_flag, hipStream_t s) {
  const int64_t total = rows * E;
  const int grid = stream_grid(total, 256, 256 * 32);
  hipLaunchKernelGGL((gather_rows_elem_kernel<T, IdxT>), dim3(grid), dim3(256), 0, s, (const T*)table, idx,
                     offsets, (T*)out, total, E, N, V, err_flag);
  return check_launch("gather_rows(elem)");
}

template <typename IdxT>
static int gather_dispatch(const void* table, int64_t V, int E, int dtype, const IdxT* idx,
                           const int64_t* offsets, int64_t rows, int N, void* out, int32_t* err_flag,
                           hipStream_t s) {
  const int row_bytes = E * dtype_size(dtype);
  if (row_bytes % 16 == 0 && aligned16(table) && aligned16(out)) {
    return launch_gather_vec<IdxT>(table, idx, offsets, out, rows, row_bytes / 16, N, V, err_flag, s);
  }
  if (dtype == TRS_F32) return launch_gather_elem<float, IdxT>(table, idx, offsets, out, rows, E, N, V, err_flag, s);
  return launch_gather_elem<bf16_t, IdxT>(table, idx, offsets, out, rows, E, N, V, err_flag, s);
}

// ---- field-aware gather: out[b, i*N + j, :] = tables[i][g(b,j), :]
template <typename IdxT>
__global__ __launch_bounds__(256) void fa_gather_vec_kernel(
    const uint4* const* __restrict__ tables, const IdxT* __restrict__ idx, const int64_t* __restrict__ offsets,
    uint4* __restrict__ out, int64_t B, int N, int vpr, int64_t V, int32_t* __restrict__ err_flag) {
  // one item = (b, i, j, vec); consecutive threads walk vec, then j, then i: writes are contiguous.
  const int per_b = N * N * vpr;
  const int64_t total = B * per_b;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool f32 = total < ((int64_t)1 << 32);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = udiv_fast(t, per_b, f32);
    unsigned rem = (unsigned)(t - b * per_b);
    const unsigned i = rem / (unsigned)(N * vpr);
    rem -= i * N * vpr;
    const unsigned j = rem / (unsigned)vpr;
    const int lv = (int)(rem - j * vpr);
    const int64_t r = load_row_id(idx, offsets, b * N + j, (int)j);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (err_flag != nullptr && (r < 0 || r >= V)) {
      *err_flag = 1;
    } else {
      v = tables[i][r * vpr + lv];
    }
    store_stream(&out[t], v);      // 12.8 GB at B = 65 536, N = 39: never re-read by this kernel
  }
}

// The same for rows of a power of two of 16-byte units, round 5: a workgroup takes a sample at a time (its N row ids once
// into LDS), lane groups of one row walk the N x N (table, field) items four in flight; rows of tables larger than the
// caches (N tables of V rows: 5 GB at the BASELINE shape) come in by streaming loads.
template <typename IdxT, int LOG2U, bool STREAM>
__global__ __launch_bounds__(256) void fa_gather_rows_kernel(const uint4* const* __restrict__ tables,
                                                             const IdxT* __restrict__ idx,
                                                             const int64_t* __restrict__ offsets, uint4* __restrict__ out,
                                                             int64_t B, int N, int64_t V, int32_t* __restrict__ err_flag) {
  constexpr int U = 1 << LOG2U, G = 256 >> LOG2U, PIPE = 4;
  extern __shared__ __attribute__((aligned(16))) char fa_lds[];
  int64_t* rid = reinterpret_cast<int64_t*>(fa_lds);
  const int lv = threadIdx.x & (U - 1), grp = threadIdx.x >> LOG2U;
  const int NN = N * N;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
      int64_t r = load_row_id(idx, offsets, b * N + n, n);
      if (r < 0 || r >= V) {
        if (err_flag != nullptr) *err_flag = 1;
        r = -1;
      }
      rid[n] = r;
    }
    __syncthreads();
    uint4* ob = out + b * NN * U;
    for (int t0 = grp; t0 < NN; t0 += G * PIPE) {
      uint4 v[PIPE];
#pragma unroll
      for (int k = 0; k < PIPE; ++k) {
        const int t = t0 + k * G;
        v[k] = make_uint4(0, 0, 0, 0);
        if (t < NN) {
          const int i = t / N, j = t - i * N;
          const int64_t r = rid[j];
          if (r >= 0) v[k] = STREAM ? load_stream(&tables[i][r * U + lv]) : tables[i][r * U + lv];
        }
      }
#pragma unroll
      for (int k = 0; k < PIPE; ++k) {
        const int t = t0 + k * G;
        if (t < NN) store_stream(&ob[t * U + lv], v[k]);
      }
    }
  }
}

template <typename T, typename IdxT>
__global__ __launch_bounds__(256) void fa_gather_elem_kernel(
    const T* const* __restrict__ tables, const IdxT* __restrict__ idx, const int64_t* __restrict__ offsets,
    T* __restrict__ out, int64_t B, int N, int E, int64_t V, int32_t* __restrict__ err_flag) {
  const int64_t per_b = (int64_t)N * N * E;
  const int64_t total = B * per_b;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / per_b;
    int rem = (int)(t - b * per_b);
    const int i = rem / (N * E);
    rem -= i * N * E;
    const int j = rem / E;
    const int e = rem - j * E;
    const int64_t r = load_row_id(idx, offsets, b * N + j, j);
    if (err_flag != nullptr && (r < 0 || r >= V)) {
      *err_flag = 1;
      out[t] = T{};
    } else {
      out[t] = tables[i][r * E + e];
    }
  }
}

// ---- permutation helpers for the sharded lookup
template <bool SCATTER>
__global__ __launch_bounds__(256) void permute_rows_vec_kernel(const uint4* __restrict__ rows,
                                                               const int32_t* __restrict__ pos,
                                                               uint4* __restrict__ out, int64_t K, int vpr) {
  const int64_t total = K * vpr;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool f32 = total < ((int64_t)1 << 32);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t k = udiv_fast(t, vpr, f32);
    const int lv = (int)(t - k * vpr);
    const int64_t p = pos[k];
    if (SCATTER) store_stream(&out[p * vpr + lv], rows[t]);
    else store_stream(&out[t], rows[p * vpr + lv]);
  }
}
template <typename T, bool SCATTER>
__global__ __launch_bounds__(256) void permute_rows_elem_kernel(const T* __restrict__ rows,
                                                                const int32_t* __restrict__ pos,
                                                                T* __restrict__ out, int64_t K, int E) {
  const int64_t total = K * E;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t k = t / E;
    const int e = (int)(t - k * E);
    const int64_t p = pos[k];
    if (SCATTER) out[p * E + e] = rows[t];
    else out[t] = rows[p * E + e];
  }
}

template <bool SCATTER>
static int permute_rows(const void* rows, const int32_t* pos, int64_t K, int E, int dtype, void* out,
                        hipStream_t s) {
  TRS_REQUIRE(rows && pos && out, TRS_EINVAL, "permute_rows: NULL pointer");
  TRS_REQUIRE(K >= 0 && E > 0, TRS_EINVAL, "permute_rows: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "permute_rows: dtype %d", dtype);
  if (K == 0) return TRS_OK;
  const int row_bytes = E * dtype_size(dtype);
  if (row_bytes % 16 == 0 && aligned16(rows) && aligned16(out)) {
    const int vpr = row_bytes / 16;
    hipLaunchKernelGGL((permute_rows_vec_kernel<SCATTER>), dim3(stream_grid(K * vpr, 256, 8192)), dim3(256), 0,
                       s, (const uint4*)rows, pos, (uint4*)out, K, vpr);
  } else if (dtype == TRS_F32) {
    hipLaunchKernelGGL((permute_rows_elem_kernel<float, SCATTER>), dim3(stream_grid(K * E, 256, 8192)),
                       dim3(256), 0, s, (const float*)rows, pos, (float*)out, K, E);
  } else {
    hipLaunchKernelGGL((permute_rows_elem_kernel<bf16_t, SCATTER>), dim3(stream_grid(K * E, 256, 8192)),
                       dim3(256), 0, s, (const bf16_t*)rows, pos, (bf16_t*)out, K, E);
  }
  return check_launch("permute_rows");
}

// ---- index staging: per-field columns -> (B, W) index matrix ---------------------------------------
constexpr int PACK_MAX_W = 256;
constexpr int PACK_TB = 64;          // samples per tile
struct PackArgs {
  const void* col[PACK_MAX_W];       // base pointer of output column c inside its source
  int32_t stride[PACK_MAX_W];        // elements between consecutive samples in that source
};

template <typename SrcT, typename DstT>
__global__ __launch_bounds__(256) void pack_columns_kernel(PackArgs a, int W, int64_t B, DstT* __restrict__ out) {
  extern __shared__ char pack_smem[];
  DstT* tile = reinterpret_cast<DstT*>(pack_smem);      // [PACK_TB][W + 1]
  const int64_t b0 = (int64_t)blockIdx.x * PACK_TB;
  const int nb = (int)((B - b0) < PACK_TB ? (B - b0) : PACK_TB);
  // read: consecutive lanes walk the samples of one column (contiguous for 1-D sources)
  for (int e = threadIdx.x; e < W * PACK_TB; e += 256) {
    const int c = e / PACK_TB, bl = e - c * PACK_TB;
    if (bl < nb) tile[bl * (W + 1) + c] = (DstT) reinterpret_cast<const SrcT*>(a.col[c])[(b0 + bl) * a.stride[c]];
  }
  __syncthreads();
  // write: rows of the index matrix are contiguous
  const int total = nb * W;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int bl = e / W, c = e - bl * W;
    out[b0 * W + e] = tile[bl * (W + 1) + c];
  }
}

__global__ void mark_timestamp_kernel(unsigned long long* ring, int capacity) {
  if (threadIdx.x == 0) {
    const unsigned long long i = ring[0];
    ring[0] = i + 1;
    ring[1 + (i % (unsigned long long)capacity)] = wall_clock64();
  }
}

}  // namespace trs

using namespace trs;

extern "C" int trs_pack_columns(const void* const* srcs, const int32_t* widths, int32_t nsrc, int32_t src_dtype,
                                int64_t B, void* out, int32_t out_dtype, trs_stream_t stream) {
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(srcs && widths && out, TRS_EINVAL, "pack_columns: NULL pointer");
  TRS_REQUIRE(nsrc > 0 && B > 0, TRS_EINVAL, "pack_columns: bad size");
  TRS_REQUIRE((src_dtype == TRS_I64 || src_dtype == TRS_I32) && (out_dtype == TRS_I64 || out_dtype == TRS_I32),
              TRS_EDTYPE, "pack_columns: dtypes %d -> %d", src_dtype, out_dtype);
  PackArgs a;
  int W = 0;
  const int esz = src_dtype == TRS_I64 ? 8 : 4;
  for (int j = 0; j < nsrc; ++j) {
    TRS_REQUIRE(srcs[j] != nullptr && widths[j] > 0, TRS_EINVAL, "pack_columns: source %d is NULL or empty", j);
    TRS_REQUIRE(W + widths[j] <= PACK_MAX_W, TRS_ESHAPE, "pack_columns: more than %d columns", PACK_MAX_W);
    for (int t = 0; t < widths[j]; ++t) {
      a.col[W] = (const char*)srcs[j] + (size_t)t * esz;
      a.stride[W] = widths[j];
      ++W;
    }
  }
  const int grid = (int)((B + PACK_TB - 1) / PACK_TB);
  const size_t lds = (size_t)PACK_TB * (W + 1) * (out_dtype == TRS_I64 ? 8 : 4);
  TRS_REQUIRE(lds <= 64 * 1024, TRS_ESHAPE, "pack_columns: %d columns of %d-byte indices exceed the 64 KB tile", W,
              out_dtype == TRS_I64 ? 8 : 4);
  hipStream_t s = (hipStream_t)stream;
  if (src_dtype == TRS_I64 && out_dtype == TRS_I64)
    hipLaunchKernelGGL((pack_columns_kernel<int64_t, int64_t>), dim3(grid), dim3(256), lds, s, a, W, B, (int64_t*)out);
  else if (src_dtype == TRS_I64)
    hipLaunchKernelGGL((pack_columns_kernel<int64_t, int32_t>), dim3(grid), dim3(256), lds, s, a, W, B, (int32_t*)out);
  else if (out_dtype == TRS_I64)
    hipLaunchKernelGGL((pack_columns_kernel<int32_t, int64_t>), dim3(grid), dim3(256), lds, s, a, W, B, (int64_t*)out);
  else
    hipLaunchKernelGGL((pack_columns_kernel<int32_t, int32_t>), dim3(grid), dim3(256), lds, s, a, W, B, (int32_t*)out);
  return check_launch("pack_columns");
}

extern "C" int trs_mark_timestamp(uint64_t* ring, int32_t capacity, trs_stream_t stream) {
  TRS_REQUIRE(ring != nullptr && capacity > 0, TRS_EINVAL, "mark_timestamp: NULL ring or capacity <= 0");
  hipLaunchKernelGGL(mark_timestamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)ring,
                     (int)capacity);
  return check_launch("mark_timestamp");
}

extern "C" int64_t trs_wall_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
  return khz;
}

extern "C" int trs_version(void) { return TRS_ABI_VERSION; }
extern "C" const char* trs_last_error_string(void) { return trs::err_buf(); }

// ---- N separate tables, one column of the index block each (a StackedInput of SingleIndexEmbeddings, reference
// inputs/base/stacked_inp.py:94-134: N nn.Embedding lookups + a cat): out[b, n, :] = tables[n][idx[b, n], :].  The tables
// stay the caller's separate allocations (N parameters); the kernel takes their base pointers and row counts.  Same lane
// layout as gather_rows_vec_kernel: one lane = one 16-byte vector of a row, UNROLL independent rows in flight.
template <typename IdxT, int UNROLL>
__global__ __launch_bounds__(256) void gather_tables_vec_kernel(const uint4* const* __restrict__ tables,
                                                                const int64_t* __restrict__ table_rows,
                                                                const IdxT* __restrict__ idx, uint4* __restrict__ out,
                                                                int64_t total_vecs, int vpr, int N,
                                                                int32_t* __restrict__ err_flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; t0 < total_vecs; t0 += stride * UNROLL) {
    uint4 v[UNROLL];
    bool ok[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t t = t0 + u * stride;
      ok[u] = t < total_vecs;
      v[u] = make_uint4(0, 0, 0, 0);
      if (ok[u]) {
        const unsigned p = (unsigned)(t / vpr);      // flat positions fit 32 bits (checked by the entry point)
        const int lane_v = (int)(t - (int64_t)p * vpr);
        const int n = (int)(p % (unsigned)N);
        const int64_t r = (int64_t)idx[p];
        if (r < 0 || r >= table_rows[n]) {
          if (err_flag != nullptr) *err_flag = 1;      // reads as a zero row
        } else {
          v[u] = tables[n][r * vpr + lane_v];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t t = t0 + u * stride;
      if (ok[u]) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        const u32x4 w = {v[u].x, v[u].y, v[u].z, v[u].w};
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(&out[t]));
      }
    }
  }
}

template <typename T, typename IdxT>
__global__ __launch_bounds__(256) void gather_tables_elem_kernel(const T* const* __restrict__ tables,
                                                                 const int64_t* __restrict__ table_rows,
                                                                 const IdxT* __restrict__ idx, T* __restrict__ out,
                                                                 int64_t total, int E, int N, int32_t* __restrict__ err_flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t p = t / E;
    const int e = (int)(t - p * E);
    const int n = (int)(p % N);
    const int64_t r = (int64_t)idx[p];
    if (r < 0 || r >= table_rows[n]) {
      if (err_flag != nullptr) *err_flag = 1;
      out[t] = T{};
    } else {
      out[t] = tables[n][r * E + e];
    }
  }
}

template <typename IdxT>
static int gather_tables_dispatch(const void* const* tables, const int64_t* table_rows, int E, int dtype, const IdxT* idx,
                                  int64_t rows, int N, void* out, int32_t* err_flag, hipStream_t s) {
  const int row_bytes = E * dtype_size(dtype);
  if (row_bytes % 16 == 0 && aligned16(out)) {      // (the tables' own alignment: the host checks every base pointer)
    const int vpr = row_bytes / 16;
    const int64_t total = rows * vpr;
    constexpr int UNROLL = 4;
    hipLaunchKernelGGL((gather_tables_vec_kernel<IdxT, UNROLL>), dim3(stream_grid((total + UNROLL - 1) / UNROLL, 256, 256 * 32)),
                       dim3(256), 0, s, (const uint4* const*)tables, table_rows, idx, (uint4*)out, total, vpr, N, err_flag);
  } else if (dtype == TRS_F32) {
    hipLaunchKernelGGL((gather_tables_elem_kernel<float, IdxT>), dim3(stream_grid(rows * E, 256, 256 * 32)), dim3(256), 0, s,
                       (const float* const*)tables, table_rows, idx, (float*)out, rows * E, E, N, err_flag);
  } else {
    hipLaunchKernelGGL((gather_tables_elem_kernel<bf16_t, IdxT>), dim3(stream_grid(rows * E, 256, 256 * 32)), dim3(256), 0, s,
                       (const bf16_t* const*)tables, table_rows, idx, (bf16_t*)out, rows * E, E, N, err_flag);
  }
  return check_launch("gather_rows_tables");
}

extern "C" int trs_gather_rows_tables(const void* const* tables, const int64_t* table_rows, int32_t E, int32_t dtype,
                                      const void* idx, int32_t idx_dtype, int64_t B, int32_t N, void* out,
                                      int32_t* err_flag, trs_stream_t stream) {
  if (B == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  TRS_REQUIRE(tables && table_rows && idx && out, TRS_EINVAL, "gather_rows_tables: NULL pointer");
  TRS_REQUIRE(E > 0 && B >= 0 && N > 0, TRS_EINVAL, "gather_rows_tables: bad size E=%d B=%lld N=%d", E, (long long)B, N);
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "gather_rows_tables: dtype %d", dtype);
  TRS_REQUIRE(idx_dtype == TRS_I64 || idx_dtype == TRS_I32, TRS_EDTYPE, "gather_rows_tables: idx dtype %d", idx_dtype);
  TRS_REQUIRE(B * (int64_t)N < ((int64_t)1 << 32), TRS_ESHAPE, "gather_rows_tables: B*N = %lld lookups must fit 32 bits",
              (long long)(B * (int64_t)N));
  hipStream_t s = (hipStream_t)stream;
  if (idx_dtype == TRS_I64)
    return gather_tables_dispatch<int64_t>(tables, table_rows, E, dtype, (const int64_t*)idx, B * N, N, out, err_flag, s);
  return gather_tables_dispatch<int32_t>(tables, table_rows, E, dtype, (const int32_t*)idx, B * N, N, out, err_flag, s);
}

extern "C" int trs_gather_rows(const void* table, int64_t V, int32_t E, int32_t dtype, const void* idx,
                               int32_t idx_dtype, const int64_t* offsets, int64_t B, int32_t N, void* out,
                               int32_t* err_flag, trs_stream_t stream) {
  if (B == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  TRS_REQUIRE(table && idx && out, TRS_EINVAL, "gather_rows: NULL pointer");
  TRS_REQUIRE(V > 0 && E > 0 && B >= 0 && N > 0, TRS_EINVAL, "gather_rows: bad size V=%lld E=%d B=%lld N=%d",
              (long long)V, E, (long long)B, N);
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "gather_rows: dtype %d", dtype);
  TRS_REQUIRE(idx_dtype == TRS_I64 || idx_dtype == TRS_I32, TRS_EDTYPE, "gather_rows: idx dtype %d", idx_dtype);
  TRS_REQUIRE(B * (int64_t)N < ((int64_t)1 << 32), TRS_ESHAPE, "gather_rows: B*N = %lld lookups must fit 32 bits",
              (long long)(B * (int64_t)N));
  hipStream_t s = (hipStream_t)stream;
  if (idx_dtype == TRS_I64)
    return gather_dispatch<int64_t>(table, V, E, dtype, (const int64_t*)idx, offsets, B * N, N, out, err_flag, s);
  return gather_dispatch<int32_t>(table, V, E, dtype, (const int32_t*)idx, offsets, B * N, N, out, err_flag, s);
}

template <typename IdxT>
static int fa_dispatch(const void* const* tables, int64_t V, int E, int dtype, const IdxT* idx,
                       const int64_t* offsets, int64_t B, int N, void* out, int32_t* err_flag, hipStream_t s) {
  const int row_bytes = E * dtype_size(dtype);
  if (row_bytes % 16 == 0 && aligned16(out)) {
    const int vpr = row_bytes / 16;
    const int64_t total = B * N * N * vpr;
    if (is_pow2(vpr) && vpr <= 64 && N <= 4096) {
      int lg = 0;
      while ((1 << lg) < vpr) ++lg;
      const bool stream = (size_t)N * V * row_bytes > ((size_t)512 << 20);
      const int grid = (int)std::min<int64_t>(B, 256 * 8);
      const size_t lds = (size_t)N * 8;
#define TRS_FAG(LG)                                                                                                     \
  if (stream)                                                                                                           \
    hipLaunchKernelGGL((fa_gather_rows_kernel<IdxT, LG, true>), dim3(grid), dim3(256), lds, s,                           \
                       (const uint4* const*)tables, idx, offsets, (uint4*)out, B, N, V, err_flag);                      \
  else                                                                                                                  \
    hipLaunchKernelGGL((fa_gather_rows_kernel<IdxT, LG, false>), dim3(grid), dim3(256), lds, s,                          \
                       (const uint4* const*)tables, idx, offsets, (uint4*)out, B, N, V, err_flag)
      switch (lg) {
        case 0: TRS_FAG(0); break;
        case 1: TRS_FAG(1); break;
        case 2: TRS_FAG(2); break;
        case 3: TRS_FAG(3); break;
        case 4: TRS_FAG(4); break;
        case 5: TRS_FAG(5); break;
        default: TRS_FAG(6); break;
      }
#undef TRS_FAG
      return check_launch("fa_gather_rows");
    }
    hipLaunchKernelGGL((fa_gather_vec_kernel<IdxT>), dim3(stream_grid(total, 256, 256 * 32)), dim3(256), 0, s,
                       (const uint4* const*)tables, idx, offsets, (uint4*)out, B, N, vpr, V, err_flag);
  } else if (dtype == TRS_F32) {
    const int64_t total = B * N * N * E;
    hipLaunchKernelGGL((fa_gather_elem_kernel<float, IdxT>), dim3(stream_grid(total, 256, 256 * 32)), dim3(256),
                       0, s, (const float* const*)tables, idx, offsets, (float*)out, B, N, E, V, err_flag);
  } else {
    const int64_t total = B * N * N * E;
    hipLaunchKernelGGL((fa_gather_elem_kernel<bf16_t, IdxT>), dim3(stream_grid(total, 256, 256 * 32)), dim3(256),
                       0, s, (const bf16_t* const*)tables, idx, offsets, (bf16_t*)out, B, N, E, V, err_flag);
  }
  return check_launch("fa_gather_rows");
}

extern "C" int trs_fa_gather_rows(const void* const* tables, int64_t V, int32_t E, int32_t dtype,
                                  const void* idx, int32_t idx_dtype, const int64_t* offsets, int64_t B,
                                  int32_t N, void* out, int32_t* err_flag, trs_stream_t stream) {
  if (B == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  TRS_REQUIRE(tables && idx && out, TRS_EINVAL, "fa_gather_rows: NULL pointer");
  TRS_REQUIRE(V > 0 && E > 0 && B >= 0 && N > 0, TRS_EINVAL, "fa_gather_rows: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "fa_gather_rows: dtype %d", dtype);
  TRS_REQUIRE(idx_dtype == TRS_I64 || idx_dtype == TRS_I32, TRS_EDTYPE, "fa_gather_rows: idx dtype %d", idx_dtype);
  if (B == 0) return TRS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (idx_dtype == TRS_I64)
    return fa_dispatch<int64_t>(tables, V, E, dtype, (const int64_t*)idx, offsets, B, N, out, err_flag, s);
  return fa_dispatch<int32_t>(tables, V, E, dtype, (const int32_t*)idx, offsets, B, N, out, err_flag, s);
}

extern "C" int trs_scatter_by_pos(const void* rows, const int32_t* pos, int64_t K, int32_t E, int32_t dtype,
                                  void* out, trs_stream_t stream) {
  if (K == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  return permute_rows<true>(rows, pos, K, E, dtype, out, (hipStream_t)stream);
}
extern "C" int trs_gather_by_pos(const void* rows, const int32_t* pos, int64_t K, int32_t E, int32_t dtype,
                                 void* out, trs_stream_t stream) {
  if (K == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  return permute_rows<false>(rows, pos, K, E, dtype, out, (hipStream_t)stream);
}
