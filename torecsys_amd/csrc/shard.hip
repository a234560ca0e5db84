// Row-sharded tables (multi-GPU lookup): bucket the B*N global row ids of the local batch by owner rank.
// counting sort with W buckets: per-workgroup LDS histograms, one global atomic per (workgroup, owner).
#include "trs_common.hpp"

namespace trs {

constexpr int MAX_WORLD = 256;
constexpr int CHUNK = 4096;  // positions per workgroup in the fill pass

template <typename IdxT>
__global__ __launch_bounds__(256) void bucket_count_kernel(const IdxT* __restrict__ idx,
                                                           const int64_t* __restrict__ offsets, int64_t BN, int N,
                                                           int64_t per, int W, unsigned long long* __restrict__ counts) {
  __shared__ int hist[MAX_WORLD];
  for (int w = threadIdx.x; w < W; w += blockDim.x) hist[w] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < BN; p += stride) {
    const int64_t r = load_row_id(idx, offsets, p, (int)((uint64_t)p < ((uint64_t)1 << 32) ? (unsigned)p % (unsigned)N : p % N));
    int w = (int)(r / per);
    w = w < 0 ? 0 : (w >= W ? W - 1 : w);
    atomicAdd(&hist[w], 1);
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
  const int64_t p0 = (int64_t)blockIdx.x * CHUNK;
  const int64_t p1 = p0 + CHUNK < BN ? p0 + CHUNK : BN;
  for (int w = threadIdx.x; w < W; w += blockDim.x) hist[w] = 0;
  __syncthreads();
  for (int64_t p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    const int64_t r = load_row_id(idx, offsets, p, (int)((uint64_t)p < ((uint64_t)1 << 32) ? (unsigned)p % (unsigned)N : p % N));
    int w = (int)(r / per);
    w = w < 0 ? 0 : (w >= W ? W - 1 : w);
    atomicAdd(&hist[w], 1);
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
  for (int64_t p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    const int64_t r = load_row_id(idx, offsets, p, (int)((uint64_t)p < ((uint64_t)1 << 32) ? (unsigned)p % (unsigned)N : p % N));
    int w = (int)(r / per);
    w = w < 0 ? 0 : (w >= W ? W - 1 : w);
    const long long slot = base[w] + atomicAdd(&hist[w], 1);
    send_ids[slot] = (int32_t)(r - (int64_t)w * per);
    send_pos[slot] = (int32_t)p;
    if (inv_pos) inv_pos[p] = (int32_t)slot;
  }
}

// counts / cursor start at zero.  A kernel, not hipMemsetAsync: inside a captured hipGraph the 8-byte memset nodes were
// seen racing with the kernels behind them (the fill pass then computes slots from garbage and writes out of bounds)
__global__ void bucket_zero_kernel(unsigned long long* __restrict__ a, unsigned long long* __restrict__ b, int W) {
  for (int w = threadIdx.x; w < W; w += blockDim.x) { a[w] = 0ull; b[w] = 0ull; }
}

static size_t align_up_s(size_t x, size_t a) { return (x + a - 1) / a * a; }

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
