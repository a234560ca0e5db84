// Element-wise epilogues of the MLP backward that PyTorch runs as separate passes (the GEMMs themselves stay on
// hipBLASLt): for y = relu(x W^T + b), given gy and y,
//     gz[r,c] = gy[r,c] * (y[r,c] > 0)          gb[c] = sum_r gz[r,c]
// in ONE pass (ATen: threshold_backward, then a column-sum that re-reads gz).  HBM-bound: 2 reads + 1 write of
// rows x C elements.  Each workgroup owns a band of rows, every thread keeps the column sums of its 16-byte
// column vector in registers, bands are combined through a [bands][C] fp32 partial buffer by a second kernel.
#include <algorithm>

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

// ---------------------------------------------------------------------------------------------
// Linear with ONE output unit (the logit layer of the DeepFM / xDeepFM MLP): out[r] = h[r,:] . w + b.
// As a GEMM it is a 1-column problem (hipBLASLt: 20 us forward, 28 us input gradient, 86 us weight gradient at
// 65 536 x 512); it is a row-wise dot product and its transpose -- HBM-bound passes over h.
// Lanes along the columns (one 16-byte vector per lane), LPR = C/VE lanes per row, 256/LPR rows per workgroup pass.
template <typename T>
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const uint4* __restrict__ h, const uint4* __restrict__ w,
                                                         const T* __restrict__ bias, int64_t rows, int lpr,
                                                         T* __restrict__ out) {
  constexpr int VE = Vec16<T>::VE;
  const int v = threadIdx.x % lpr, rr = threadIdx.x / lpr, rpp = 256 / lpr;
  float wv[VE];
  Vec16<T>::unpack(w[v], wv);
  const float b0 = bias != nullptr ? to_f32(bias[0]) : 0.f;
  constexpr int U = 4;          // independent row loads in flight per thread (one load per iteration is latency-bound)
  const int64_t step = (int64_t)gridDim.x * rpp;
  for (int64_t r0 = (int64_t)blockIdx.x * rpp + rr; r0 < rows; r0 += step * U) {
    uint4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * step;
      raw[u] = make_uint4(0, 0, 0, 0);
      if (r < rows) raw[u] = h[r * lpr + v];
    }
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float x[VE];
      Vec16<T>::unpack(raw[u], x);
      acc[u] = 0.f;
#pragma unroll
      for (int k = 0; k < VE; ++k) acc[u] = fmaf(x[k], wv[k], acc[u]);
    }
    if (lpr >= 4) {
      // reduce the U = 4 row sums together: two exchange steps leave each lane with ONE row's partial (row index =
      // two bits of v), the remaining log2(lpr) - 2 steps run on a single value: lpr-lane sums of 4 rows in
      // log2(lpr) + 1 shuffles instead of 4 log2(lpr)
      const int h1 = lpr >> 1, h2 = lpr >> 2;
      const bool up1 = (v & h1) != 0, up2 = (v & h2) != 0;
      // step 1 (distance lpr/2): lower half keeps rows 0,1, upper half rows 2,3
      const float s0 = up1 ? acc[0] : acc[2], s1 = up1 ? acc[1] : acc[3];
      float k0 = (up1 ? acc[2] : acc[0]) + __shfl_xor(s0, h1, 64);
      float k1 = (up1 ? acc[3] : acc[1]) + __shfl_xor(s1, h1, 64);
      // step 2 (distance lpr/4): keep one of the two
      const float s2 = up2 ? k0 : k1;
      float k = (up2 ? k1 : k0) + __shfl_xor(s2, h2, 64);
      for (int o = lpr >> 3; o > 0; o >>= 1) k += __shfl_xor(k, o, 64);
      // the lanes with the low log2(lpr)-2 bits of v clear hold row u = 2*up1 + up2
      if ((v & (h2 - 1)) == 0) {
        const int u = (up1 ? 2 : 0) + (up2 ? 1 : 0);
        const int64_t r = r0 + u * step;
        if (r < rows) out[r] = from_f32<T>(k + b0);
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * step;
        float a = acc[u];
        for (int o = lpr >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (v == 0 && r < rows) out[r] = from_f32<T>(a + b0);
      }
    }
  }
}

// gh[r,:] = g[r] * w;  partial[blk][c] = sum over the workgroup's rows of g[r] * h[r,c];  partial[blk][C] = sum g[r]
template <typename T>
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const T* __restrict__ g, const uint4* __restrict__ h,
                                                         const uint4* __restrict__ w, int64_t rows, int lpr,
                                                         uint4* __restrict__ gh, float* __restrict__ partial) {
  constexpr int VE = Vec16<T>::VE;
  __shared__ float red[256][VE + 1];
  const int v = threadIdx.x % lpr, rr = threadIdx.x / lpr, rpp = 256 / lpr;
  const int C = lpr * VE;
  float wv[VE], gw[VE], gb = 0.f;
  Vec16<T>::unpack(w[v], wv);
#pragma unroll
  for (int k = 0; k < VE; ++k) gw[k] = 0.f;
  constexpr int U = 4;
  const int64_t step = (int64_t)gridDim.x * rpp;
  for (int64_t r0 = (int64_t)blockIdx.x * rpp + rr; r0 < rows; r0 += step * U) {
    uint4 raw[U];
    float gr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * step;
      raw[u] = make_uint4(0, 0, 0, 0);
      gr[u] = 0.f;
      if (r < rows) {
        gr[u] = to_f32(g[r]);
        if (partial != nullptr) raw[u] = h[r * lpr + v];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * step;
      if (r >= rows) continue;
      if (partial != nullptr) {
        float x[VE];
        Vec16<T>::unpack(raw[u], x);
#pragma unroll
        for (int k = 0; k < VE; ++k) gw[k] = fmaf(gr[u], x[k], gw[k]);
        if (v == 0) gb += gr[u];
      }
      if (gh != nullptr) {
        float o[VE];
#pragma unroll
        for (int k = 0; k < VE; ++k) o[k] = gr[u] * wv[k];
        gh[r * lpr + v] = Vec16<T>::pack(o);
      }
    }
  }
  if (partial == nullptr) return;
#pragma unroll
  for (int k = 0; k < VE; ++k) red[threadIdx.x][k] = gw[k];
  red[threadIdx.x][VE] = gb;
  __syncthreads();
  if (rr == 0) {
    float* mine = partial + (size_t)blockIdx.x * (C + 1);
#pragma unroll
    for (int k = 0; k < VE; ++k) {
      float t = 0.f;
      for (int q = 0; q < rpp; ++q) t += red[q * lpr + v][k];
      mine[v * VE + k] = t;
    }
    if (v == 0) {
      float t = 0.f;
      for (int q = 0; q < rpp; ++q) t += red[q * lpr][VE];
      mine[C] = t;
    }
  }
}

__global__ void rowdot_finish_kernel(const float* __restrict__ tmp, int C, float* __restrict__ gw,
                                         float* __restrict__ gb) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) gw[c] = tmp[c];
  else if (c == C) gb[0] = tmp[C];
}

static int rowdot_fwd_blocks(int64_t rows, int lpr) {      // no partial buffer to bound the forward's grid
  const int rpp = 256 / lpr;
  return (int)std::min<int64_t>((rows + (int64_t)rpp * 4 - 1) / ((int64_t)rpp * 4), 8192);
}
static int rowdot_blocks(int64_t rows, int lpr) {
  const int rpp = 256 / lpr;
  return (int)std::min<int64_t>((rows + rpp - 1) / rpp, 1023);
}
static bool rowdot_shape_ok(int C, int dtype) {
  const int ve = dtype == TRS_F32 ? 4 : 8;
  const int lpr = C % ve == 0 ? C / ve : 0;
  return lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0;
}

// Finish of the split-K weight gradient: gw[r,c] = sum_s part[s,r,c] for the un-padded (out_rows x out_cols) corner of
// the (S, R, Cc) fp32 partial products, cast to the parameter dtype; blockIdx.y == gridDim.y-1 also casts the fp32
// bias gradient.  One launch instead of ATen's sum(0) + strided slice copy + cast (+ slice + cast for the bias).
template <typename T>
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ part, int S, int R, int Cc,
                                                          int out_rows, int out_cols, T* __restrict__ gw,
                                                          const float* __restrict__ gbf, T* __restrict__ gb) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if ((int)blockIdx.y == out_rows) {           // the extra row of workgroups: bias gradient
    if (gb != nullptr && c < out_rows) gb[c] = from_f32<T>(gbf[c]);
    return;
  }
  if (c >= out_cols) return;
  const int r = blockIdx.y;
  const float* p = part + (size_t)r * Cc + c;
  const size_t stride = (size_t)R * Cc;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = 0;
  for (; s + 4 <= S; s += 4) {
    a0 += p[(size_t)s * stride];
    a1 += p[(size_t)(s + 1) * stride];
    a2 += p[(size_t)(s + 2) * stride];
    a3 += p[(size_t)(s + 3) * stride];
  }
  for (; s < S; ++s) a0 += p[(size_t)s * stride];
  gw[(size_t)r * out_cols + c] = from_f32<T>((a0 + a1) + (a2 + a3));
}

// The finish of a layer with FEW output rows and MANY splits (the 400 -> 1 head at 65 536 rows: 256 partial products of
// 8 x 400): one thread per column would walk the 256 partials in 64 dependent rounds inside two workgroups (20.8 us
// for 3.3 MB).  Here a column belongs to 16 threads, each adds every 16th partial, the 16 sums meet in LDS.  Bias row as
// above.  The sum of a column is taken in another order than wgrad_finish_kernel's (fp32, then one rounding to T).
constexpr int WF_SL = 16;            // threads per column
constexpr int WF_COLS = 256 / WF_SL; // columns per workgroup
template <typename T>
__global__ __launch_bounds__(256) void wgrad_finish_few_rows_kernel(const float* __restrict__ part, int S, int R, int Cc,
                                                                   int out_rows, int out_cols, T* __restrict__ gw,
                                                                   const float* __restrict__ gbf, T* __restrict__ gb) {
  __shared__ float red[WF_SL][WF_COLS + 1];
  if ((int)blockIdx.y == out_rows) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (gb != nullptr && c < out_rows) gb[c] = from_f32<T>(gbf[c]);
    return;
  }
  const int cl = threadIdx.x % WF_COLS, sl = threadIdx.x / WF_COLS;
  const int c = blockIdx.x * WF_COLS + cl;
  const int r = blockIdx.y;
  const size_t stride = (size_t)R * Cc;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < out_cols) {
    const float* p = part + (size_t)r * Cc + c;
    int s = sl;
    for (; s + 3 * WF_SL < S; s += 4 * WF_SL) {
      a0 += p[(size_t)s * stride];
      a1 += p[(size_t)(s + WF_SL) * stride];
      a2 += p[(size_t)(s + 2 * WF_SL) * stride];
      a3 += p[(size_t)(s + 3 * WF_SL) * stride];
    }
    for (; s < S; s += WF_SL) a0 += p[(size_t)s * stride];
  }
  red[sl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && c < out_cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < WF_SL; ++k) t += red[k][cl];
    gw[(size_t)r * out_cols + c] = from_f32<T>(t);
  }
}

// The same finish for partial products stored TRANSPOSED, part (S, Cc, R) = slices of x^T g: for a wide input layer
// (2496 x 512) hipBLASLt runs that orientation 1.5x faster than g^T x.  32 x 32 tiles through LDS so that both the
// reads (along r) and the writes (along c) are coalesced.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_finish_t_kernel(const float* __restrict__ part, int S, int Cc, int R,
                                                            int out_rows, int out_cols, T* __restrict__ gw,
                                                            const float* __restrict__ gbf, T* __restrict__ gb) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  if ((int)blockIdx.y == (out_cols + 31) / 32) {                // the extra row of workgroups: bias gradient
    const int r = blockIdx.x * 32 + tx;
    if (gb != nullptr && ty == 0 && r < out_rows) gb[r] = from_f32<T>(gbf[r]);
    return;
  }
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const size_t stride = (size_t)Cc * R;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < out_cols && r < out_rows) {
      const float* p = part + (size_t)c * R + r;
      float a0 = 0.f, a1 = 0.f;
      int sidx = 0;
      for (; sidx + 2 <= S; sidx += 2) {
        a0 += p[(size_t)sidx * stride];
        a1 += p[(size_t)(sidx + 1) * stride];
      }
      if (sidx < S) a0 += p[(size_t)sidx * stride];
      acc[i] = a0 + a1;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = acc[i];     // tile[c - c0][r - r0]
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < out_rows && c < out_cols) gw[(size_t)r * out_cols + c] = from_f32<T>(tile[tx][ty + 8 * i]);
  }
}

}  // namespace trs

using namespace trs;

/* Linear with one output unit: out[r] = h[r,:] . w + bias[0]        h (rows, C), w (C), C * sizeof(T) / 16 a power of
 * two <= 64 (C = 512 bf16, 256 fp32, ...).  bwd: gh (rows, C) = g[r] * w (may be NULL); gw (C) and gb (1) fp32 =
 * sum_r g[r] h[r,:] and sum_r g[r], ACCUMULATED into (may both be NULL); workspace: partial sums per workgroup.    */
extern "C" int trs_rowdot_fwd(const void* h, const void* w, const void* bias, int64_t rows, int32_t C, int32_t dtype,
                              void* out, trs_stream_t stream) {
  TRS_REQUIRE(rows >= 0 && C > 0, TRS_EINVAL, "rowdot_fwd: bad size");
  if (rows == 0) return TRS_OK;
  TRS_REQUIRE(h && w && out, TRS_EINVAL, "rowdot_fwd: NULL pointer");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "rowdot_fwd: dtype %d", dtype);
  TRS_REQUIRE(rowdot_shape_ok(C, dtype), TRS_ESHAPE, "rowdot_fwd: C = %d (C*elem/16 must be a power of two <= 64)", C);
  TRS_REQUIRE(aligned16(h) && aligned16(w), TRS_EALIGN, "rowdot_fwd: 16-byte alignment");
  const int lpr = C / (dtype == TRS_F32 ? 4 : 8);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((rowdot_fwd_kernel<float>), dim3(rowdot_fwd_blocks(rows, lpr)), dim3(256), 0, s, (const uint4*)h,
                       (const uint4*)w, (const float*)bias, rows, lpr, (float*)out);
  else
    hipLaunchKernelGGL((rowdot_fwd_kernel<bf16_t>), dim3(rowdot_fwd_blocks(rows, lpr)), dim3(256), 0, s, (const uint4*)h,
                       (const uint4*)w, (const bf16_t*)bias, rows, lpr, (bf16_t*)out);
  return check_launch("rowdot_fwd");
}

extern "C" size_t trs_rowdot_bwd_workspace_bytes(int64_t rows, int32_t C) {
  return (size_t)1024 * (size_t)(C + 1) * 4 + 256;
}

extern "C" int trs_rowdot_bwd(const void* g, const void* h, const void* w, int64_t rows, int32_t C, int32_t dtype,
                              void* gh, float* gw, float* gb, void* workspace, size_t ws_bytes, trs_stream_t stream) {
  TRS_REQUIRE(rows >= 0 && C > 0, TRS_EINVAL, "rowdot_bwd: bad size");
  if (rows == 0) return TRS_OK;
  TRS_REQUIRE(g && h && w, TRS_EINVAL, "rowdot_bwd: NULL pointer");
  TRS_REQUIRE((gw == nullptr) == (gb == nullptr), TRS_EINVAL, "rowdot_bwd: gw and gb go together");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "rowdot_bwd: dtype %d", dtype);
  TRS_REQUIRE(rowdot_shape_ok(C, dtype), TRS_ESHAPE, "rowdot_bwd: C = %d (C*elem/16 must be a power of two <= 64)", C);
  TRS_REQUIRE(aligned16(h) && aligned16(w) && aligned16(gh), TRS_EALIGN, "rowdot_bwd: 16-byte alignment");
  TRS_REQUIRE(gw == nullptr || (workspace != nullptr && ws_bytes >= trs_rowdot_bwd_workspace_bytes(rows, C)),
              TRS_EWORKSPACE, "rowdot_bwd: workspace too small");
  const int lpr = C / (dtype == TRS_F32 ? 4 : 8);
  const int grid = rowdot_blocks(rows, lpr);
  hipStream_t s = (hipStream_t)stream;
  float* part = gw != nullptr ? (float*)workspace : nullptr;
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((rowdot_bwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)g, (const uint4*)h,
                       (const uint4*)w, rows, lpr, (uint4*)gh, part);
  else
    hipLaunchKernelGGL((rowdot_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)g, (const uint4*)h,
                       (const uint4*)w, rows, lpr, (uint4*)gh, part);
  if (gw != nullptr) {
    // column sums of the (grid, C+1) partials; the last column is sum g
    float* tmp = part + (size_t)grid * (C + 1);          // C+1 floats behind the partials (inside the 1024-block budget)
    if (grid >= 1024) return fail(TRS_EWORKSPACE, "rowdot_bwd: internal workspace layout");
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((C + 1 + 15) / 16), dim3(256), 0, s, part, grid, C + 1, tmp);
    hipLaunchKernelGGL(rowdot_finish_kernel, dim3((C + 1 + 255) / 256), dim3(256), 0, s, tmp, C, gw, gb);
  }
  return check_launch("rowdot_bwd");
}

// ------------------------------------------------------------------------------------------------
// One-output Linear over the concatenation of two blocks, without the concatenation (the head of
// deep_and_cross_network.py:82-92: torch.cat([cross_out, deep_out], dim='O') -> flatten(('N','O')) -> nn.Linear(cat, 1)).
// A row of the virtual concatenation is VA + VD 16-byte vectors (a's N*Ea values, then d's N*Eb); lane l of a wave owns
// the vectors l, l + 64, ... of every row its wave walks, so the weights it needs (w is field-major [a_f | d_f]) are
// per-lane constants and stay in registers, packed.  One wave per row: 10 loads of 16 B in flight per lane at
// 39 x (64 + 64) bf16.  HBM-bound: 2 x 327 MB read (forward), the same read plus 2 x 327 MB written (backward) at
// 65 536 rows -- against cat (654 MB read + written), a 4992-wide GEMV, its two backward GEMMs (654 MB written, 654 MB
// read), two slice copies and the add of the block gradient in the composition.
namespace trs {
struct CatHeadShape {
  int VA, VD;            // vectors per row of a / d
  int va_pf, vd_pf;      // vectors per field
  int wstride, Ea;       // elements: Ea + Eb, Ea
};
// element offset into w of virtual vector j (j < VA + VD)
__device__ __forceinline__ int cat_head_woff(const CatHeadShape& sh, int j, int VE) {
  if (j < sh.VA) {
    const int f = j / sh.va_pf;
    return f * sh.wstride + (j - f * sh.va_pf) * VE;
  }
  const int jj = j - sh.VA;
  const int f = jj / sh.vd_pf;
  return f * sh.wstride + sh.Ea + (jj - f * sh.vd_pf) * VE;
}

template <typename T, int ITERS>
__global__ __launch_bounds__(256) void cat_head_fwd_kernel(const uint4* __restrict__ a, const uint4* __restrict__ d,
                                                           const T* __restrict__ w, const T* __restrict__ bias,
                                                           int64_t rows, CatHeadShape sh, T* __restrict__ out) {
  constexpr int VE = Vec16<T>::VE;
  const int lane = threadIdx.x & 63;
  const int VT = sh.VA + sh.VD;
  uint4 wp[ITERS];
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int j = lane + 64 * i;
    wp[i] = j < VT ? *reinterpret_cast<const uint4*>(w + cat_head_woff(sh, j, VE)) : make_uint4(0, 0, 0, 0);
  }
  const float b0 = bias != nullptr ? to_f32(bias[0]) : 0.f;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nw) {
    uint4 raw[ITERS];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int j = lane + 64 * i;
      raw[i] = make_uint4(0, 0, 0, 0);
      if (j < sh.VA) raw[i] = load_stream(&a[r * sh.VA + j]);
      else if (j < VT) raw[i] = load_stream(&d[r * sh.VD + (j - sh.VA)]);
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      float x[VE], wv[VE];
      Vec16<T>::unpack(raw[i], x);
      Vec16<T>::unpack(wp[i], wv);
#pragma unroll
      for (int k = 0; k < VE; ++k) acc = fmaf(x[k], wv[k], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) out[r] = from_f32<T>(acc + b0);
  }
}

// ga[r,:] = g[r] * w_a, gd[r,:] = g[r] * w_d;  partial[blk][c] = sum over the workgroup's rows of g[r] * cat[r,c] in w's
// order, partial[blk][C] = sum g[r]
template <typename T, int ITERS>
__global__ __launch_bounds__(256) void cat_head_bwd_kernel(const T* __restrict__ g, const uint4* __restrict__ a,
                                                           const uint4* __restrict__ d, const T* __restrict__ w,
                                                           int64_t rows, CatHeadShape sh, uint4* __restrict__ ga,
                                                           uint4* __restrict__ gd, float* __restrict__ partial) {
  constexpr int VE = Vec16<T>::VE;
  __shared__ float red[ITERS * 64 * VE + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int VT = sh.VA + sh.VD;
  const int C = VT * VE;
  uint4 wp[ITERS];
  float acc[ITERS][VE];
  float gsum = 0.f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int j = lane + 64 * i;
    wp[i] = j < VT ? *reinterpret_cast<const uint4*>(w + cat_head_woff(sh, j, VE)) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < VE; ++k) acc[i][k] = 0.f;
  }
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += nw) {
    const float gr = to_f32(g[r]);
    gsum += gr;
    uint4 raw[ITERS];
    if (partial != nullptr) {
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
        const int j = lane + 64 * i;
        raw[i] = make_uint4(0, 0, 0, 0);
        if (j < sh.VA) raw[i] = load_stream(&a[r * sh.VA + j]);
        else if (j < VT) raw[i] = load_stream(&d[r * sh.VD + (j - sh.VA)]);
      }
    }
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int j = lane + 64 * i;
      float wv[VE], o[VE];
      Vec16<T>::unpack(wp[i], wv);
#pragma unroll
      for (int k = 0; k < VE; ++k) o[k] = gr * wv[k];
      if (j < sh.VA) {
        if (ga != nullptr) ga[r * sh.VA + j] = Vec16<T>::pack(o);
      } else if (j < VT) {
        if (gd != nullptr) gd[r * sh.VD + (j - sh.VA)] = Vec16<T>::pack(o);
      }
      if (partial != nullptr) {
        float x[VE];
        Vec16<T>::unpack(raw[i], x);
#pragma unroll
        for (int k = 0; k < VE; ++k) acc[i][k] = fmaf(gr, x[k], acc[i][k]);
      }
    }
  }
  if (partial == nullptr) return;
  // the four waves add their sums into LDS one after the other (20 KB instead of four images)
  for (int wv_ = 0; wv_ < 4; ++wv_) {
    if (wave == wv_) {
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < VE; ++k) {
          float* p = &red[((i * 64 + lane) * VE) + k];
          *p = wv_ == 0 ? acc[i][k] : *p + acc[i][k];
        }
      }
      if (lane == 0) red[ITERS * 64 * VE] = wv_ == 0 ? gsum : red[ITERS * 64 * VE] + gsum;
    }
    __syncthreads();
  }
  float* mine = partial + (size_t)blockIdx.x * (C + 1);
  for (int e = threadIdx.x; e < C; e += 256) {
    const int j = e / VE, k = e - j * VE;
    mine[cat_head_woff(sh, j, VE) + k] = red[e];        // red is indexed by (virtual vector j, k): j = lane + 64 i
  }
  if (threadIdx.x == 0) mine[C] = red[ITERS * 64 * VE];
}

static bool cat_head_shape(int N, int Ea, int Eb, int dtype, CatHeadShape* sh) {
  const int ve = dtype == TRS_F32 ? 4 : 8;
  if (N <= 0 || Ea <= 0 || Eb <= 0 || Ea % ve != 0 || Eb % ve != 0) return false;
  sh->va_pf = Ea / ve;
  sh->vd_pf = Eb / ve;
  sh->VA = N * sh->va_pf;
  sh->VD = N * sh->vd_pf;
  sh->wstride = Ea + Eb;
  sh->Ea = Ea;
  return sh->VA + sh->VD <= 64 * 16;
}
static int cat_head_blocks(int64_t rows) { return (int)std::min<int64_t>((rows + 3) / 4, 512); }

template <typename T, int ITERS>
static void cat_head_launch_fwd(const void* a, const void* d, const void* w, const void* bias, int64_t rows,
                                const CatHeadShape& sh, void* out, hipStream_t s) {
  hipLaunchKernelGGL((cat_head_fwd_kernel<T, ITERS>), dim3((int)std::min<int64_t>((rows + 3) / 4, 4096)), dim3(256), 0, s,
                     (const uint4*)a, (const uint4*)d, (const T*)w, (const T*)bias, rows, sh, (T*)out);
}
template <typename T, int ITERS>
static void cat_head_launch_bwd(const void* g, const void* a, const void* d, const void* w, int64_t rows,
                                const CatHeadShape& sh, void* ga, void* gd, float* part, hipStream_t s) {
  hipLaunchKernelGGL((cat_head_bwd_kernel<T, ITERS>), dim3(cat_head_blocks(rows)), dim3(256), 0, s, (const T*)g,
                     (const uint4*)a, (const uint4*)d, (const T*)w, rows, sh, (uint4*)ga, (uint4*)gd, part);
}
}  // namespace trs

// ITERS = vectors per lane: 2 / 4 / 6 / 8 / 10 / 12 / 16 cover up to 1024 vectors per row (10: 39 x (64 + 64) bf16)
#define TRS_CAT_HEAD_DISPATCH(T, FN, ...)                                            \
  do {                                                                               \
    const int it = (sh.VA + sh.VD + 63) / 64;                                        \
    if (it <= 2) FN<T, 2>(__VA_ARGS__);                                              \
    else if (it <= 4) FN<T, 4>(__VA_ARGS__);                                         \
    else if (it <= 6) FN<T, 6>(__VA_ARGS__);                                         \
    else if (it <= 8) FN<T, 8>(__VA_ARGS__);                                         \
    else if (it <= 10) FN<T, 10>(__VA_ARGS__);                                       \
    else if (it <= 12) FN<T, 12>(__VA_ARGS__);                                       \
    else FN<T, 16>(__VA_ARGS__);                                                     \
  } while (0)

extern "C" int trs_cat_head_fwd(const void* a, const void* d, const void* w, const void* bias, int64_t rows, int32_t N,
                                int32_t Ea, int32_t Eb, int32_t dtype, void* out, trs_stream_t stream) {
  TRS_REQUIRE(rows >= 0, TRS_EINVAL, "cat_head_fwd: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "cat_head_fwd: dtype %d", dtype);
  CatHeadShape sh;
  TRS_REQUIRE(cat_head_shape(N, Ea, Eb, dtype, &sh), TRS_ESHAPE,
              "cat_head_fwd: N = %d, Ea = %d, Eb = %d (rows of whole 16-byte vectors, at most 1024 per sample)", N, Ea, Eb);
  if (rows == 0) return TRS_OK;
  TRS_REQUIRE(a && d && w && out, TRS_EINVAL, "cat_head_fwd: NULL pointer");
  TRS_REQUIRE(aligned16(a) && aligned16(d) && aligned16(w), TRS_EALIGN, "cat_head_fwd: 16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TRS_F32) TRS_CAT_HEAD_DISPATCH(float, cat_head_launch_fwd, a, d, w, bias, rows, sh, out, s);
  else TRS_CAT_HEAD_DISPATCH(bf16_t, cat_head_launch_fwd, a, d, w, bias, rows, sh, out, s);
  return check_launch("cat_head_fwd");
}

extern "C" size_t trs_cat_head_bwd_workspace_bytes(int64_t rows, int32_t N, int32_t Ea, int32_t Eb) {
  (void)rows;
  return (size_t)513 * ((size_t)N * (Ea + Eb) + 1) * 4 + 256;
}

extern "C" int trs_cat_head_bwd(const void* g, const void* a, const void* d, const void* w, int64_t rows, int32_t N,
                                int32_t Ea, int32_t Eb, int32_t dtype, void* ga, void* gd, float* gw, float* gb,
                                void* workspace, size_t ws_bytes, trs_stream_t stream) {
  TRS_REQUIRE(rows >= 0, TRS_EINVAL, "cat_head_bwd: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "cat_head_bwd: dtype %d", dtype);
  TRS_REQUIRE((gw == nullptr) == (gb == nullptr), TRS_EINVAL, "cat_head_bwd: gw and gb go together");
  CatHeadShape sh;
  TRS_REQUIRE(cat_head_shape(N, Ea, Eb, dtype, &sh), TRS_ESHAPE,
              "cat_head_bwd: N = %d, Ea = %d, Eb = %d (rows of whole 16-byte vectors, at most 1024 per sample)", N, Ea, Eb);
  const int C = N * (Ea + Eb);
  hipStream_t s = (hipStream_t)stream;
  if (rows == 0) {
    if (gw != nullptr) {
      if (int rc = zero_bytes(gw, (size_t)C * 4, s)) return rc;
      return zero_bytes(gb, 4, s);
    }
    return TRS_OK;
  }
  TRS_REQUIRE(g && w && (gw == nullptr || (a && d)), TRS_EINVAL, "cat_head_bwd: NULL pointer");
  TRS_REQUIRE(aligned16(a) && aligned16(d) && aligned16(w) && aligned16(ga) && aligned16(gd), TRS_EALIGN,
              "cat_head_bwd: 16-byte alignment");
  TRS_REQUIRE(gw == nullptr || (workspace != nullptr && ws_bytes >= trs_cat_head_bwd_workspace_bytes(rows, N, Ea, Eb)),
              TRS_EWORKSPACE, "cat_head_bwd: workspace too small");
  float* part = gw != nullptr ? (float*)workspace : nullptr;
  if (dtype == TRS_F32) TRS_CAT_HEAD_DISPATCH(float, cat_head_launch_bwd, g, a, d, w, rows, sh, ga, gd, part, s);
  else TRS_CAT_HEAD_DISPATCH(bf16_t, cat_head_launch_bwd, g, a, d, w, rows, sh, ga, gd, part, s);
  if (gw != nullptr) {
    const int grid = cat_head_blocks(rows);
    float* tmp = part + (size_t)grid * (C + 1);
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((C + 1 + 15) / 16), dim3(256), 0, s, part, grid, C + 1, tmp);
    hipLaunchKernelGGL(rowdot_finish_kernel, dim3((C + 1 + 255) / 256), dim3(256), 0, s, tmp, C, gw, gb);
  }
  return check_launch("cat_head_bwd");
}

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
    return zero_bytes(gb, (size_t)C * 4, s);
  }
  TRS_REQUIRE(gy && y && gz, TRS_EINVAL, "relu_bwd_bias: NULL pointer");
  TRS_REQUIRE(aligned16(gy) && aligned16(y) && aligned16(gz), TRS_EALIGN, "relu_bwd_bias: 16-byte alignment");
  const int vpr = row_bytes / 16;
  int nbands = (int)std::min<int64_t>(1024, (rows + 31) / 32);
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

extern "C" int trs_wgrad_finish(const float* part, int32_t S, int32_t R, int32_t Cc, int32_t out_rows, int32_t out_cols,
                                int32_t dtype, void* gw, const float* gb_f32, void* gb, trs_stream_t stream) {
  TRS_REQUIRE(S > 0 && R > 0 && Cc > 0 && out_rows > 0 && out_cols > 0 && out_rows <= R && out_cols <= Cc, TRS_EINVAL,
              "wgrad_finish: bad size");
  TRS_REQUIRE(part && gw, TRS_EINVAL, "wgrad_finish: NULL pointer");
  TRS_REQUIRE((gb == nullptr) == (gb_f32 == nullptr), TRS_EINVAL, "wgrad_finish: gb and gb_f32 go together");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "wgrad_finish: dtype %d", dtype);
  TRS_REQUIRE(gb == nullptr || out_rows <= ((out_cols + 255) / 256) * 256, TRS_ESHAPE,
              "wgrad_finish: out_rows %d exceeds the bias row of workgroups", out_rows);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((out_cols + 255) / 256, out_rows + 1);
  if (S >= 4 * WF_SL && (int64_t)grid.x * out_rows < 64) {      // few rows, many splits: 16 threads per column
    dim3 gridf((out_cols + WF_COLS - 1) / WF_COLS, out_rows + 1);
    if (dtype == TRS_F32)
      hipLaunchKernelGGL((wgrad_finish_few_rows_kernel<float>), gridf, dim3(256), 0, s, part, S, R, Cc, out_rows, out_cols,
                         (float*)gw, gb_f32, (float*)gb);
    else
      hipLaunchKernelGGL((wgrad_finish_few_rows_kernel<bf16_t>), gridf, dim3(256), 0, s, part, S, R, Cc, out_rows, out_cols,
                         (bf16_t*)gw, gb_f32, (bf16_t*)gb);
    return check_launch("wgrad_finish");
  }
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((wgrad_finish_kernel<float>), grid, dim3(256), 0, s, part, S, R, Cc, out_rows, out_cols,
                       (float*)gw, gb_f32, (float*)gb);
  else
    hipLaunchKernelGGL((wgrad_finish_kernel<bf16_t>), grid, dim3(256), 0, s, part, S, R, Cc, out_rows, out_cols,
                       (bf16_t*)gw, gb_f32, (bf16_t*)gb);
  return check_launch("wgrad_finish");
}

extern "C" int trs_wgrad_finish_t(const float* part, int32_t S, int32_t Cc, int32_t R, int32_t out_rows, int32_t out_cols,
                                  int32_t dtype, void* gw, const float* gb_f32, void* gb, trs_stream_t stream) {
  TRS_REQUIRE(S > 0 && R > 0 && Cc > 0 && out_rows > 0 && out_cols > 0 && out_rows <= R && out_cols <= Cc, TRS_EINVAL,
              "wgrad_finish_t: bad size");
  TRS_REQUIRE(part && gw, TRS_EINVAL, "wgrad_finish_t: NULL pointer");
  TRS_REQUIRE((gb == nullptr) == (gb_f32 == nullptr), TRS_EINVAL, "wgrad_finish_t: gb and gb_f32 go together");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "wgrad_finish_t: dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((out_rows + 31) / 32, (out_cols + 31) / 32 + 1);
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((wgrad_finish_t_kernel<float>), grid, dim3(256), 0, s, part, S, Cc, R, out_rows, out_cols,
                       (float*)gw, gb_f32, (float*)gb);
  else
    hipLaunchKernelGGL((wgrad_finish_t_kernel<bf16_t>), grid, dim3(256), 0, s, part, S, Cc, R, out_rows, out_cols,
                       (bf16_t*)gw, gb_f32, (bf16_t*)gb);
  return check_launch("wgrad_finish_t");
}

// ------------------------------------------------------------------------------------------------
// several strided 2-D copies in ONE launch: desc[k] = {src, dst, rows, cols, src_ld, dst_ld} (addresses as int64,
// sizes in elements).  The padded copies of an MLP stack's weights and biases (8 tensors of 1 - 2 000 000 elements)
// are refreshed before every training forward; as separate copies they were 8 launches of ~5 us each.
namespace trs {
__global__ __launch_bounds__(256) void copy_padded_many_kernel(const int64_t* __restrict__ desc, int elem_size) {
  const int64_t* d = desc + (size_t)blockIdx.y * 6;
  const char* src = reinterpret_cast<const char*>(d[0]);
  char* dst = reinterpret_cast<char*>(d[1]);
  const int64_t rows = d[2], cols = d[3], sld = d[4], dld = d[5];
  const int64_t rb = cols * elem_size;                     // bytes per row
  const bool vec = rb % 16 == 0 && (sld * elem_size) % 16 == 0 && (dld * elem_size) % 16 == 0 &&
                   (d[0] & 15) == 0 && (d[1] & 15) == 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (vec) {
    const int64_t vpr = rb / 16, total = rows * vpr;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
      const int64_t r = t / vpr, c = t - r * vpr;
      *reinterpret_cast<uint4*>(dst + r * dld * elem_size + c * 16) =
          *reinterpret_cast<const uint4*>(src + r * sld * elem_size + c * 16);
    }
  } else {
    const int64_t total = rows * rb;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
      const int64_t r = t / rb, c = t - r * rb;
      dst[r * dld * elem_size + c] = src[r * sld * elem_size + c];
    }
  }
}
}  // namespace trs

// dst (rows, c_out) = [src (rows, c_in) | 0]: the (B,1) logit gradient widened to the 8 columns the fused MLP tail's last
// layer runs at -- one launch instead of a fill and a strided copy (2-byte elements)
namespace trs {
__global__ __launch_bounds__(256) void pad_cols_kernel(const uint16_t* __restrict__ src, int c_in, uint16_t* __restrict__ dst,
                                                       int c_out, int64_t total) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t r = t / c_out;
    const int c = (int)(t - r * c_out);
    dst[t] = c < c_in ? src[r * c_in + c] : (uint16_t)0;
  }
}
}  // namespace trs

extern "C" int trs_pad_cols(const void* src, int32_t c_in, void* dst, int32_t c_out, int64_t rows, int32_t dtype,
                            trs_stream_t stream) {
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "pad_cols: bf16 only");
  TRS_REQUIRE(c_in > 0 && c_out >= c_in && rows >= 0, TRS_ESHAPE, "pad_cols: %d -> %d columns", c_in, c_out);
  if (rows == 0) return TRS_OK;
  TRS_REQUIRE(src && dst, TRS_EINVAL, "pad_cols: NULL pointer");
  const int64_t total = rows * c_out;
  hipLaunchKernelGGL(trs::pad_cols_kernel, dim3(trs::stream_grid(total, 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)src, c_in, (uint16_t*)dst, c_out, total);
  return trs::check_launch("pad_cols");
}

extern "C" int trs_copy_padded_many(const int64_t* desc, int32_t n, int32_t elem_size, int64_t max_elems,
                                    trs_stream_t stream) {
  if (n <= 0) return TRS_OK;
  TRS_REQUIRE(desc != nullptr, TRS_EINVAL, "copy_padded_many: NULL descriptor table");
  TRS_REQUIRE(elem_size == 2 || elem_size == 4, TRS_EDTYPE, "copy_padded_many: element size %d", elem_size);
  const int64_t work = std::max<int64_t>(1, max_elems * elem_size / 16);
  const int gx = (int)std::min<int64_t>((work + 255) / 256, 256);
  hipLaunchKernelGGL(trs::copy_padded_many_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, desc, elem_size);
  return trs::check_launch("copy_padded_many");
}
