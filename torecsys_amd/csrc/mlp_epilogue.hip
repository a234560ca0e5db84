// Element-wise epilogues of the MLP backward that PyTorch runs as separate passes (the GEMMs themselves stay on
// hipBLASLt): for y = relu(x W^T + b), given gy and y,
//     gz[r,c] = gy[r,c] * (y[r,c] > 0)          gb[c] = sum_r gz[r,c]
// in ONE pass (ATen: threshold_backward, then a column-sum that re-reads gz).  HBM-bound: 2 reads + 1 write of
// rows x C elements.  Each workgroup owns a band of rows, every thread keeps the column sums of its 16-byte
// column vector in registers, bands are combined through a [bands][C] fp32 partial buffer by a second kernel.
#include "trs_common.hpp"

namespace trs {

constexpr int RB_THREADS = 256;

template <typename T>
__global__ __launch_bounds__(RB_THREADS) void relu_bwd_bias_kernel(const uint4* __restrict__ gy, const uint4* __restrict__ y,
                                                                   uint4* __restrict__ gz, float* __restrict__ part,
                                                                   int64_t rows, int vpr /* 16-B vectors per row */,
                                                                   int rows_per_band) {
  constexpr int VE = Vec16<T>::VE;
  __shared__ float red[RB_THREADS][VE];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_band;
  const int64_t r1 = r0 + rows_per_band < rows ? r0 + rows_per_band : rows;
  // thread -> (column vector cv, row lane rl): consecutive threads walk the vectors of one row (coalesced)
  const int cv = threadIdx.x % vpr, rl = threadIdx.x / vpr, rstep = RB_THREADS / vpr;
  float acc[VE];
#pragma unroll
  for (int k = 0; k < VE; ++k) acc[k] = 0.f;
  if (rl < rstep) {
    constexpr int U = 4;   // independent row loads in flight per thread
    for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)rstep * U) {
      uint4 gv[U], yv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = rb + (int64_t)u * rstep;
        gv[u] = make_uint4(0, 0, 0, 0);
        yv[u] = make_uint4(0, 0, 0, 0);
        if (r < r1) {
          gv[u] = gy[r * vpr + cv];
          yv[u] = y[r * vpr + cv];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = rb + (int64_t)u * rstep;
        if (r < r1) {
          float g[VE], a[VE];
          Vec16<T>::unpack(gv[u], g);
          Vec16<T>::unpack(yv[u], a);
#pragma unroll
          for (int k = 0; k < VE; ++k) {
            g[k] = a[k] > 0.f ? g[k] : 0.f;
            acc[k] += g[k];
          }
          gz[r * vpr + cv] = Vec16<T>::pack(g);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < VE; ++k) red[threadIdx.x][k] = acc[k];
  __syncthreads();
  if (rl == 0) {   // one thread per column vector folds the row lanes and writes the band's partial sums
#pragma unroll
    for (int k = 0; k < VE; ++k) {
      float s = 0.f;
      for (int j = 0; j < rstep; ++j) s += red[cv + j * vpr][k];
      part[(size_t)blockIdx.x * vpr * VE + cv * VE + k] = s;
    }
  }
}

// out[c] = sum over bands of part[band][c]: 16 band lanes x 16 columns per workgroup, 8 loads in flight per thread
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ part, int nbands, int C,
                                                              float* __restrict__ out) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < C) {
    for (int b0 = bl; b0 < nbands; b0 += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + 16 * u;
        v[u] = b < nbands ? part[(size_t)b * C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
  }
  red[bl][cl] = s;
  __syncthreads();
  if (bl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][cl];
    out[c] = t;
  }
}

}  // namespace trs

using namespace trs;

extern "C" size_t trs_relu_bwd_bias_workspace_bytes(int64_t rows, int32_t C) {
  (void)rows;
  return (size_t)2048 * C * 4 + 256;
}

extern "C" int trs_relu_bwd_bias(const void* gy, const void* y, int64_t rows, int32_t C, int32_t dtype, void* gz,
                                 float* gb, void* workspace, size_t ws_bytes, trs_stream_t stream) {
  TRS_REQUIRE(rows >= 0 && C > 0, TRS_EINVAL, "relu_bwd_bias: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "relu_bwd_bias: dtype %d", dtype);
  TRS_REQUIRE(gb && workspace, TRS_EINVAL, "relu_bwd_bias: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  const int row_bytes = C * dtype_size(dtype);
  TRS_REQUIRE(row_bytes % 16 == 0 && row_bytes / 16 <= RB_THREADS, TRS_ESHAPE,
              "relu_bwd_bias: row bytes %d must be a multiple of 16 and at most %d", row_bytes, RB_THREADS * 16);
  if (rows == 0) {
    if (hipMemsetAsync(gb, 0, (size_t)C * 4, s) != hipSuccess) return check_launch("relu_bwd_bias(memset)");
    return TRS_OK;
  }
  TRS_REQUIRE(gy && y && gz, TRS_EINVAL, "relu_bwd_bias: NULL pointer");
  TRS_REQUIRE(aligned16(gy) && aligned16(y) && aligned16(gz), TRS_EALIGN, "relu_bwd_bias: 16-byte alignment");
  const int vpr = row_bytes / 16;
  int nbands = (int)std::min<int64_t>(2048, (rows + 31) / 32);
  const int rows_per_band = (int)((rows + nbands - 1) / nbands);
  nbands = (int)((rows + rows_per_band - 1) / rows_per_band);
  TRS_REQUIRE(ws_bytes >= (size_t)nbands * C * 4, TRS_EWORKSPACE, "relu_bwd_bias: workspace too small");
  float* part = (float*)workspace;
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((relu_bwd_bias_kernel<float>), dim3(nbands), dim3(RB_THREADS), 0, s, (const uint4*)gy,
                       (const uint4*)y, (uint4*)gz, part, rows, vpr, rows_per_band);
  else
    hipLaunchKernelGGL((relu_bwd_bias_kernel<bf16_t>), dim3(nbands), dim3(RB_THREADS), 0, s, (const uint4*)gy,
                       (const uint4*)y, (uint4*)gz, part, rows, vpr, rows_per_band);
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((C + 15) / 16), dim3(256), 0, s, part, nbands, C, gb);
  return check_launch("relu_bwd_bias");
}
