// K4: cross network  x_{l+1} = x0 * (x_l W_l^T + b_l) + x0  over rows = B*N vectors of length E.
//
// Two implementations behind one entry point:
//  * generic (any E, any L, both dtypes): VALU, a tile of R rows per workgroup staged in LDS;
//  * MFMA   (bf16, E % 32 == 0, E <= 128): see cross_mfma.hip -- the whole L-layer chain of a
//    16-row tile stays in registers, W_l fragments come from LDS, so HBM sees x once in, once out.
// The backward reproduces the reference's cut gradient: x_0 enters layer 0's linear map detached
// (cross_network.py:65), so dx = sum_l g_{l+1}*(u_l+1) only (+ W_0^T du_0 iff detach_first == 0).
#include "trs_common.hpp"

namespace trs {

// cross_mfma.hip; both return 1 if the shape is not covered by the MFMA path
size_t cross_mfma_workspace_bytes(int E, int L);
int cross_mfma_fwd(const void* x, const void* W, const void* b, int64_t rows, int E, int L, void* out,
                   void* workspace, size_t ws_bytes, hipStream_t s);
int cross_mfma_bwd(const void* x, const void* W, const void* b, const void* g, int64_t rows, int E, int L,
                   void* dx, float* dW, float* db, int detach_first, void* workspace, size_t ws_bytes,
                   hipStream_t s);

constexpr int CR = 16;  // rows per tile (generic path)

template <typename T>
__global__ __launch_bounds__(256) void cross_fwd_generic(const T* __restrict__ x, const T* __restrict__ W,
                                                         const T* __restrict__ bias, int64_t rows, int E, int L,
                                                         T* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* x0 = reinterpret_cast<float*>(smem);  // [CR][E]
  float* xa = x0 + CR * E;
  float* xb = xa + CR * E;
  const int64_t ntiles = (rows + CR - 1) / CR;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * CR;
    __syncthreads();
    for (int i = threadIdx.x; i < CR * E; i += blockDim.x) {
      const int64_t row = r0 + i / E;
      const float v = row < rows ? to_f32(x[row * E + (i % E)]) : 0.f;
      x0[i] = v;
      xa[i] = v;
    }
    __syncthreads();
    for (int l = 0; l < L; ++l) {
      const T* Wl = W + (size_t)l * E * E;
      for (int i = threadIdx.x; i < CR * E; i += blockDim.x) {
        const int r = i / E, eo = i - r * E;
        float acc = to_f32(bias[l * E + eo]);
        const T* wrow = Wl + (size_t)eo * E;
        const float* xr = xa + r * E;
        for (int k = 0; k < E; ++k) acc = fmaf(to_f32(wrow[k]), xr[k], acc);
        xb[i] = fmaf(x0[i], acc, x0[i]);
      }
      __syncthreads();
      float* t = xa; xa = xb; xb = t;
    }
    for (int i = threadIdx.x; i < CR * E; i += blockDim.x) {
      const int64_t row = r0 + i / E;
      if (row < rows) out[row * E + (i % E)] = from_f32<T>(xa[i]);
    }
  }
}

// generic backward.  LDS: x0[R][E], g[R][E], gn[R][E], dx0[R][E], du[R][E], xs[L][R][E], us[L][R][E]
template <typename T>
__global__ __launch_bounds__(256) void cross_bwd_generic(const T* __restrict__ x, const T* __restrict__ W,
                                                         const T* __restrict__ bias, const T* __restrict__ gout,
                                                         int64_t rows, int E, int L, int R, T* __restrict__ dx,
                                                         float* __restrict__ dW, float* __restrict__ db,
                                                         int detach_first) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int RE = R * E;
  float* x0 = reinterpret_cast<float*>(smem);
  float* g = x0 + RE;
  float* gn = g + RE;
  float* dx0 = gn + RE;
  float* du = dx0 + RE;
  float* xs = du + RE;             // x_l, l = 0..L-1
  float* us = xs + (size_t)L * RE; // u_l
  const int64_t ntiles = (rows + R - 1) / R;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * R;
    __syncthreads();
    for (int i = threadIdx.x; i < RE; i += blockDim.x) {
      const int64_t row = r0 + i / E;
      const bool ok = row < rows;
      const float v = ok ? to_f32(x[row * E + (i % E)]) : 0.f;
      x0[i] = v;
      xs[i] = v;
      g[i] = ok ? to_f32(gout[row * E + (i % E)]) : 0.f;
      dx0[i] = 0.f;
    }
    __syncthreads();
    // forward recompute, keeping x_l and u_l
    for (int l = 0; l < L; ++l) {
      const T* Wl = W + (size_t)l * E * E;
      const float* xl = xs + (size_t)l * RE;
      for (int i = threadIdx.x; i < RE; i += blockDim.x) {
        const int r = i / E, eo = i - r * E;
        float acc = to_f32(bias[l * E + eo]);
        const T* wrow = Wl + (size_t)eo * E;
        const float* xr = xl + r * E;
        for (int k = 0; k < E; ++k) acc = fmaf(to_f32(wrow[k]), xr[k], acc);
        us[(size_t)l * RE + i] = acc;
        if (l + 1 < L) xs[(size_t)(l + 1) * RE + i] = fmaf(x0[i], acc, x0[i]);
      }
      __syncthreads();
    }
    for (int l = L - 1; l >= 0; --l) {
      const T* Wl = W + (size_t)l * E * E;
      const float* xl = xs + (size_t)l * RE;
      const float* ul = us + (size_t)l * RE;
      for (int i = threadIdx.x; i < RE; i += blockDim.x) {
        du[i] = g[i] * x0[i];
        dx0[i] = fmaf(g[i], ul[i] + 1.f, dx0[i]);
      }
      __syncthreads();
      // dW_l[eo][k] += sum_r du[r][eo] * x_l[r][k];  db_l[eo] += sum_r du[r][eo]
      for (int i = threadIdx.x; i < E * E; i += blockDim.x) {
        const int eo = i / E, k = i - eo * E;
        float s = 0.f;
        for (int r = 0; r < R; ++r) s = fmaf(du[r * E + eo], xl[r * E + k], s);
        atomicAdd(&dW[(size_t)l * E * E + i], s);
      }
      for (int eo = threadIdx.x; eo < E; eo += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += du[r * E + eo];
        atomicAdd(&db[l * E + eo], s);
      }
      if (l > 0 || !detach_first) {
        for (int i = threadIdx.x; i < RE; i += blockDim.x) {
          const int r = i / E, k = i - r * E;
          float s = 0.f;
          for (int eo = 0; eo < E; ++eo) s = fmaf(to_f32(Wl[(size_t)eo * E + k]), du[r * E + eo], s);
          gn[i] = s;
        }
      }
      __syncthreads();
      float* t = g; g = gn; gn = t;
    }
    for (int i = threadIdx.x; i < RE; i += blockDim.x) {
      const int64_t row = r0 + i / E;
      if (row < rows) dx[row * E + (i % E)] = from_f32<T>(dx0[i] + (detach_first ? 0.f : g[i]));
    }
  }
}

}  // namespace trs

using namespace trs;

extern "C" size_t trs_cross_workspace_bytes(int64_t rows, int32_t E, int32_t L, int32_t dtype) {
  (void)rows;
  if (dtype != TRS_BF16 || E <= 0 || L <= 0) return 0;
  return cross_mfma_workspace_bytes(E, L);
}

extern "C" int trs_cross_fwd(const void* x, const void* W, const void* b, int64_t rows, int32_t E, int32_t L,
                             int32_t dtype, void* out, void* workspace, size_t ws_bytes, trs_stream_t stream) {
  if (rows == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  TRS_REQUIRE(x && out && (L == 0 || (W && b)), TRS_EINVAL, "cross_fwd: NULL pointer");
  TRS_REQUIRE(rows >= 0 && E > 0 && L >= 0, TRS_EINVAL, "cross_fwd: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "cross_fwd: dtype %d", dtype);
  if (rows == 0) return TRS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TRS_BF16 && L > 0) {
    const int rc = cross_mfma_fwd(x, W, b, rows, E, L, out, workspace, ws_bytes, s);
    if (rc <= 0) return rc;
  }
  const size_t lds = (size_t)3 * CR * E * 4;
  TRS_REQUIRE(lds <= 64 * 1024, TRS_ESHAPE, "cross_fwd: E=%d too large", E);
  const int grid = (int)std::min<int64_t>((rows + CR - 1) / CR, 256 * 8);
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((cross_fwd_generic<float>), dim3(grid), dim3(256), lds, s, (const float*)x, (const float*)W,
                       (const float*)b, rows, E, L, (float*)out);
  else
    hipLaunchKernelGGL((cross_fwd_generic<bf16_t>), dim3(grid), dim3(256), lds, s, (const bf16_t*)x,
                       (const bf16_t*)W, (const bf16_t*)b, rows, E, L, (bf16_t*)out);
  return check_launch("cross_fwd");
}

extern "C" int trs_cross_bwd(const void* x, const void* W, const void* b, const void* g, int64_t rows, int32_t E,
                             int32_t L, int32_t dtype, int32_t detach_first, void* dx, float* dW, float* db,
                             void* workspace, size_t ws_bytes, trs_stream_t stream) {
  if (rows == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  TRS_REQUIRE(x && g && dx && (L == 0 || (W && b && dW && db)), TRS_EINVAL, "cross_bwd: NULL pointer");
  TRS_REQUIRE(rows >= 0 && E > 0 && L >= 0, TRS_EINVAL, "cross_bwd: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "cross_bwd: dtype %d", dtype);
  if (rows == 0) return TRS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TRS_BF16 && L > 0) {
    const int rc = cross_mfma_bwd(x, W, b, g, rows, E, L, dx, dW, db, detach_first, workspace, ws_bytes, s);
    if (rc <= 0) return rc;
  }
  int R = CR;
  auto lds_for = [&](int r) { return (size_t)(5 + 2 * (size_t)L) * r * E * 4; };
  while (R > 1 && lds_for(R) > 64 * 1024) R >>= 1;
  TRS_REQUIRE(lds_for(R) <= 64 * 1024, TRS_ESHAPE, "cross_bwd: E=%d L=%d too large", E, L);
  const int grid = (int)std::min<int64_t>((rows + R - 1) / R, 256 * 8);
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((cross_bwd_generic<float>), dim3(grid), dim3(256), lds_for(R), s, (const float*)x,
                       (const float*)W, (const float*)b, (const float*)g, rows, E, L, R, (float*)dx, dW, db,
                       detach_first);
  else
    hipLaunchKernelGGL((cross_bwd_generic<bf16_t>), dim3(grid), dim3(256), lds_for(R), s, (const bf16_t*)x,
                       (const bf16_t*)W, (const bf16_t*)b, (const bf16_t*)g, rows, E, L, R, (bf16_t*)dx, dW, db,
                       detach_first);
  return check_launch("cross_bwd");
}
