// K5/K6: compress-interaction-network layer: outer product over the field dims per (b,e), formed on the
// fly, contracted with the Conv1d(k=1) weight:
//     y[b,c,e] = bias[c] + sum_{n,h} Wc[c, n*H+h] * x0[b,n,e] * xk[b,h,e]
// The reference materialises Z = (B, N*H, E) (25-84 GB at the BASELINE shape); here Z never exists.
//
// Two implementations behind one entry point:
//  * generic (any shape, both dtypes): VALU, one sample per workgroup, x0/xk/gy staged in LDS;
//  * MFMA (bf16; see cin_mfma.hip): Z tiles are generated in registers as the A operand
//    (rows = (b,e) pairs, K = (n,h)) and contracted with Wc fragments on the matrix cores.
#include "trs_common.hpp"

namespace trs {

int cin_mfma_fwd(const void* x0, const void* xk, const void* Wc, const void* bias, int64_t B, int N, int H, int C,
                 int E, void* y, float* stats, hipStream_t s);  // returns 1 if the shape is not covered
int cin_mfma_bwd(const void* x0, const void* xk, const void* Wc, const void* gy, int64_t B, int N, int H, int C, int E,
                 float* dWc, void* dx0, void* dxk, int accumulate_dx0, hipStream_t s);

// LDS: x0s[N][E], xks[H][E] (fp32)
template <typename T>
__global__ __launch_bounds__(256) void cin_fwd_generic(const T* __restrict__ x0, const T* __restrict__ xk,
                                                       const T* __restrict__ Wc, const T* __restrict__ bias,
                                                       int64_t B, int N, int H, int C, int E, int EC,
                                                       T* __restrict__ y, float* __restrict__ stats) {
  // work item = (sample b, chunk of EC embedding columns); LDS rows are EC wide
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* x0s = reinterpret_cast<float*>(smem);
  float* xks = x0s + N * EC;
  const int nchunk = (E + EC - 1) / EC;
  for (int64_t item = blockIdx.x; item < B * nchunk; item += gridDim.x) {
    const int64_t b = item / nchunk;
    const int e0 = (int)(item - b * nchunk) * EC;
    const int ec = min(EC, E - e0);
    __syncthreads();
    for (int i = threadIdx.x; i < N * ec; i += blockDim.x)
      x0s[(i / ec) * EC + i % ec] = to_f32(x0[(b * N + i / ec) * E + e0 + i % ec]);
    for (int i = threadIdx.x; i < H * ec; i += blockDim.x)
      xks[(i / ec) * EC + i % ec] = to_f32(xk[(b * H + i / ec) * E + e0 + i % ec]);
    __syncthreads();
    for (int i = threadIdx.x; i < C * ec; i += blockDim.x) {
      const int c = i / ec, e = i - c * ec;
      const T* w = Wc + (size_t)c * N * H;
      float acc = bias ? to_f32(bias[c]) : 0.f;
      for (int n = 0; n < N; ++n) {
        float inner = 0.f;
        for (int h = 0; h < H; ++h) inner = fmaf(to_f32(w[n * H + h]), xks[h * EC + e], inner);
        acc = fmaf(x0s[n * EC + e], inner, acc);
      }
      const T yo = from_f32<T>(acc);
      y[(b * C + c) * E + e0 + e] = yo;
      if (stats) {
        const float v = to_f32(yo);
        atomicAdd(&stats[c], v);
        atomicAdd(&stats[C + c], v * v);
      }
    }
  }
}

// LDS: x0s[N][EC], xks[H][EC], gys[C][EC]; work item = (sample b, chunk of EC embedding columns)
template <typename T>
__global__ __launch_bounds__(256) void cin_bwd_generic(const T* __restrict__ x0, const T* __restrict__ xk,
                                                       const T* __restrict__ Wc, const T* __restrict__ gy, int64_t B,
                                                       int N, int H, int C, int E, int EC, float* __restrict__ dWc,
                                                       T* __restrict__ dx0, T* __restrict__ dxk, int accumulate_dx0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* x0s = reinterpret_cast<float*>(smem);
  float* xks = x0s + N * EC;
  float* gys = xks + H * EC;
  const int NH = N * H;
  const int nchunk = (E + EC - 1) / EC;
  for (int64_t item = blockIdx.x; item < B * nchunk; item += gridDim.x) {
    const int64_t b = item / nchunk;
    const int e0 = (int)(item - b * nchunk) * EC;
    const int ec = min(EC, E - e0);
    __syncthreads();
    for (int i = threadIdx.x; i < N * ec; i += blockDim.x)
      x0s[(i / ec) * EC + i % ec] = to_f32(x0[(b * N + i / ec) * E + e0 + i % ec]);
    for (int i = threadIdx.x; i < H * ec; i += blockDim.x)
      xks[(i / ec) * EC + i % ec] = to_f32(xk[(b * H + i / ec) * E + e0 + i % ec]);
    for (int i = threadIdx.x; i < C * ec; i += blockDim.x)
      gys[(i / ec) * EC + i % ec] = to_f32(gy[(b * C + i / ec) * E + e0 + i % ec]);
    __syncthreads();
    // dWc[c][n*H+h] += sum_e gy[c][e] * x0[n][e] * xk[h][e]
    if (dWc) {
      for (int i = threadIdx.x; i < C * NH; i += blockDim.x) {
        const int c = i / NH, nh = i - c * NH;
        const int n = nh / H, h = nh - n * H;
        float s = 0.f;
        for (int e = 0; e < ec; ++e) s = fmaf(gys[c * EC + e] * x0s[n * EC + e], xks[h * EC + e], s);
        atomicAdd(&dWc[i], s);
      }
    }
    // dx0[n][e] = sum_h (sum_c Wc[c][nH+h] gy[c][e]) * xk[h][e]
    if (dx0) {
      for (int i = threadIdx.x; i < N * ec; i += blockDim.x) {
        const int n = i / ec, e = i - n * ec;
        float acc = 0.f;
        for (int h = 0; h < H; ++h) {
          float m = 0.f;
          for (int c = 0; c < C; ++c) m = fmaf(to_f32(Wc[(size_t)c * NH + n * H + h]), gys[c * EC + e], m);
          acc = fmaf(m, xks[h * EC + e], acc);
        }
        const int64_t o = (b * N + n) * E + e0 + e;
        if (accumulate_dx0) acc += to_f32(dx0[o]);
        dx0[o] = from_f32<T>(acc);
      }
    }
    // dxk[h][e] = sum_n (sum_c Wc[c][nH+h] gy[c][e]) * x0[n][e]
    if (dxk) {
      for (int i = threadIdx.x; i < H * ec; i += blockDim.x) {
        const int h = i / ec, e = i - h * ec;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) {
          float m = 0.f;
          for (int c = 0; c < C; ++c) m = fmaf(to_f32(Wc[(size_t)c * NH + n * H + h]), gys[c * EC + e], m);
          acc = fmaf(m, x0s[n * EC + e], acc);
        }
        dxk[(b * H + h) * E + e0 + e] = from_f32<T>(acc);
      }
    }
  }
}

}  // namespace trs

using namespace trs;

#define TRS_CIN_CHECK(name)                                                                            \
  TRS_REQUIRE(B >= 0 && N > 0 && H > 0 && C > 0 && E > 0, TRS_EINVAL, name ": bad size");              \
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, name ": dtype %d", dtype)

extern "C" int trs_cin_fwd(const void* x0, const void* xk, const void* Wc, const void* bias, int64_t B, int32_t N,
                           int32_t H, int32_t C, int32_t E, int32_t dtype, void* y, float* stats,
                           trs_stream_t stream) {
  if (B == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  TRS_REQUIRE(x0 && xk && Wc && y, TRS_EINVAL, "cin_fwd: NULL pointer");
  TRS_CIN_CHECK("cin_fwd");
  if (B == 0) return TRS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TRS_BF16) {
    const int rc = cin_mfma_fwd(x0, xk, Wc, bias, B, N, H, C, E, y, stats, s);
    if (rc <= 0) return rc;
  }
  int EC = E;
  while (EC > 1 && (size_t)(N + H) * EC * 4 > 48 * 1024) EC = (EC + 1) / 2;
  const size_t lds = (size_t)(N + H) * EC * 4;
  TRS_REQUIRE(lds <= 64 * 1024, TRS_ESHAPE, "cin_fwd: N+H = %d too large for the generic path", N + H);
  const int grid = (int)std::min<int64_t>(B * ((E + EC - 1) / EC), 256 * 8);
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((cin_fwd_generic<float>), dim3(grid), dim3(256), lds, s, (const float*)x0, (const float*)xk,
                       (const float*)Wc, (const float*)bias, B, N, H, C, E, EC, (float*)y, stats);
  else
    hipLaunchKernelGGL((cin_fwd_generic<bf16_t>), dim3(grid), dim3(256), lds, s, (const bf16_t*)x0, (const bf16_t*)xk,
                       (const bf16_t*)Wc, (const bf16_t*)bias, B, N, H, C, E, EC, (bf16_t*)y, stats);
  return check_launch("cin_fwd");
}

extern "C" int trs_cin_bwd(const void* x0, const void* xk, const void* Wc, const void* gy, int64_t B, int32_t N,
                           int32_t H, int32_t C, int32_t E, int32_t dtype, float* dWc, void* dx0, void* dxk,
                           int32_t accumulate_dx0, trs_stream_t stream) {
  if (B == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  TRS_REQUIRE(x0 && xk && Wc && gy, TRS_EINVAL, "cin_bwd: NULL pointer");
  TRS_CIN_CHECK("cin_bwd");
  if (B == 0) return TRS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TRS_BF16) {
    const int rc = cin_mfma_bwd(x0, xk, Wc, gy, B, N, H, C, E, dWc, dx0, dxk, accumulate_dx0, s);
    if (rc <= 0) return rc;
  }
  int EC = E;
  while (EC > 1 && (size_t)(N + H + C) * EC * 4 > 48 * 1024) EC = (EC + 1) / 2;
  const size_t lds = (size_t)(N + H + C) * EC * 4;
  TRS_REQUIRE(lds <= 64 * 1024, TRS_ESHAPE, "cin_bwd: N+H+C = %d too large for the generic path", N + H + C);
  const int grid = (int)std::min<int64_t>(B * ((E + EC - 1) / EC), 256 * 8);
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((cin_bwd_generic<float>), dim3(grid), dim3(256), lds, s, (const float*)x0, (const float*)xk,
                       (const float*)Wc, (const float*)gy, B, N, H, C, E, EC, dWc, (float*)dx0, (float*)dxk,
                       accumulate_dx0);
  else
    hipLaunchKernelGGL((cin_bwd_generic<bf16_t>), dim3(grid), dim3(256), lds, s, (const bf16_t*)x0, (const bf16_t*)xk,
                       (const bf16_t*)Wc, (const bf16_t*)gy, B, N, H, C, E, EC, dWc, (bf16_t*)dx0, (bf16_t*)dxk,
                       accumulate_dx0);
  return check_launch("cin_bwd");
}
