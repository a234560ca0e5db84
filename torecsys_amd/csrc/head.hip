// The scalar head of the CTR models and its loss (callers M1 / M2 / M4 of section 8a):
//
//   logit[b] = sum_e fm[b,e] + sum_n feat[b,n] + sum_k extra_k[b] + bias          (B,1)
//       models/ctr/factorization_machine.py:55-66   fm_second.sum('O') + feat.sum('N') (+ bias)
//       models/ctr/deep_fm.py:75-104                 cat([fm_second, fm_first]).sum('O') + deep_out
//       models/ctr/xdeep_fm.py:117-121               feat.sum('N') + cin_out + deep_out + bias
//   loss = mean_b( max(x,0) - x*y + log1p(exp(-|x|)) ),  dL/dx = (sigmoid(x) - y) * g / B     (BCEWithLogitsLoss, mean)
//
// As ATen ops these are ~35 launches of 4-23 us on (B,<=64) tensors (two row sums, adds, casts, log-sigmoid chain and
// their backwards: ~0.16 ms of a 1.4 ms DeepFM step); here: one pass forward, none backward for the logit (its gradient is
// the incoming (B,1) column broadcast -- views), two small launches for the loss forward, one for its backward.
// HBM-bound, 13 MB at the BASELINE shape: a 16-lane group per sample, fp32 accumulation, one rounding on store.
#include <algorithm>

#include "trs_common.hpp"

namespace trs {

constexpr int HEAD_MAX_EXTRAS = 4;
struct HeadExtras {
  const void* p[HEAD_MAX_EXTRAS];
  int64_t stride[HEAD_MAX_EXTRAS];      // elements between consecutive samples
  int n;
};

template <typename T>
__global__ __launch_bounds__(256) void ctr_logit_kernel(const T* __restrict__ fm, int E, const T* __restrict__ feat, int N,
                                                        HeadExtras ex, const T* __restrict__ bias, int64_t B,
                                                        T* __restrict__ out) {
  const int lane = threadIdx.x & 15;
  const bool fm_vec = fm != nullptr && (E * (int)sizeof(T)) % 8 == 0 && (reinterpret_cast<uintptr_t>(fm) & 7u) == 0;
  const int64_t groups = (int64_t)gridDim.x * (blockDim.x >> 4);
  const float b0 = bias != nullptr ? to_f32(bias[0]) : 0.f;
  for (int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4); b < B; b += groups) {
    float acc = 0.f;
    if (fm != nullptr) {
      const T* r = fm + b * E;
      if (fm_vec) {      // 8 bytes per lane and load: a 64-wide bf16 row is one load per lane instead of four
        constexpr int VE = 8 / (int)sizeof(T);
        for (int e = lane * VE; e < E; e += 16 * VE) {
          const uint2 u = *reinterpret_cast<const uint2*>(r + e);
          if constexpr (sizeof(T) == 2) {
            acc += __uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xffff0000u) + __uint_as_float(u.y << 16) +
                   __uint_as_float(u.y & 0xffff0000u);
          } else {
            acc += __uint_as_float(u.x) + __uint_as_float(u.y);
          }
        }
      } else {
        for (int e = lane; e < E; e += 16) acc += to_f32(r[e]);
      }
    }
    if (feat != nullptr) {
      const T* r = feat + b * N;
      for (int n = lane; n < N; n += 16) acc += to_f32(r[n]);
    }
#pragma unroll
    for (int k = 0; k < HEAD_MAX_EXTRAS; ++k)      // (compile-time k: the pointers stay in scalar registers)
      if (k < ex.n && lane == k) acc += to_f32(static_cast<const T*>(ex.p[k])[b * ex.stride[k]]);
    acc += __shfl_xor(acc, 8, 16);
    acc += __shfl_xor(acc, 4, 16);
    acc += __shfl_xor(acc, 2, 16);
    acc += __shfl_xor(acc, 1, 16);
    if (lane == 0) out[b] = from_f32<T>(acc + b0);
  }
}

__device__ __forceinline__ float bce_term(float x, float y) {
  // max(x,0) - x*y + log1p(exp(-|x|)): the stable form ATen's binary_cross_entropy_with_logits uses
  return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}

constexpr int BCE_BLOCK = 256;
constexpr int BCE_MAX_BLOCKS = 256;

template <typename T, typename L>
__global__ __launch_bounds__(BCE_BLOCK) void bce_partial_kernel(const T* __restrict__ x, const L* __restrict__ y, int64_t B,
                                                                float* __restrict__ partial) {
  __shared__ float red[BCE_BLOCK / 64];
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * BCE_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BCE_BLOCK + threadIdx.x; i < B; i += stride)
    acc += bce_term(to_f32(x[i]), to_f32(y[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < BCE_BLOCK / 64; ++w) s += red[w];
    partial[blockIdx.x] = s;
  }
}

// one wave: the partials in a fixed order (reproducible), mean over B
__global__ __launch_bounds__(64) void bce_finish_kernel(const float* __restrict__ partial, int n, float inv_B,
                                                        float* __restrict__ loss) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) acc += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (threadIdx.x == 0) *loss = acc * inv_B;
}

// batches up to BCE_ONE_MAX logits: ONE workgroup of 1024 threads does the whole mean (8 per thread) -- one launch
// instead of two where a launch is ~5 us and the work under 1 us.  Fixed order: reproducible.  NOT beyond that: at
// 65 536 logits the one workgroup ran 63 us (64 dependent load + exp + log1p rounds per thread on one CU) against
// 5 + 5 us for partial + finish (profiles/r05_bench_deepfm_step_timeline.md of a8457d2 against the one of fdf8b4f).
constexpr int64_t BCE_ONE_MAX = 1 << 13;
template <typename T, typename L>
__global__ __launch_bounds__(1024) void bce_one_kernel(const T* __restrict__ x, const L* __restrict__ y, int64_t B,
                                                       float inv_B, float* __restrict__ loss) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += 1024) acc += bce_term(to_f32(x[i]), to_f32(y[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w];
    *loss = s * inv_B;
  }
}

template <typename T, typename L>
__global__ __launch_bounds__(256) void bce_bwd_kernel(const T* __restrict__ x, const L* __restrict__ y,
                                                      const float* __restrict__ gout, float inv_B, int64_t B,
                                                      T* __restrict__ gx) {
  const float g = (gout != nullptr ? *gout : 1.f) * inv_B;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += stride) {
    const float xv = to_f32(x[i]);
    const float sg = 1.f / (1.f + expf(-xv));
    gx[i] = from_f32<T>((sg - to_f32(y[i])) * g);
  }
}

template <typename T>
static int bce_fwd_launch(const void* x, const void* y, int label_dtype, int64_t B, float* loss, float* ws, hipStream_t s) {
  if (B <= BCE_ONE_MAX) {
    if (label_dtype == TRS_F32)
      hipLaunchKernelGGL((bce_one_kernel<T, float>), dim3(1), dim3(1024), 0, s, (const T*)x, (const float*)y, B, 1.f / (float)B, loss);
    else
      hipLaunchKernelGGL((bce_one_kernel<T, bf16_t>), dim3(1), dim3(1024), 0, s, (const T*)x, (const bf16_t*)y, B, 1.f / (float)B, loss);
    return check_launch("bce_logits_fwd");
  }
  const int blocks = (int)std::min<int64_t>(BCE_MAX_BLOCKS, (B + BCE_BLOCK * 4 - 1) / (BCE_BLOCK * 4));
  if (label_dtype == TRS_F32)
    hipLaunchKernelGGL((bce_partial_kernel<T, float>), dim3(blocks), dim3(BCE_BLOCK), 0, s, (const T*)x, (const float*)y, B, ws);
  else
    hipLaunchKernelGGL((bce_partial_kernel<T, bf16_t>), dim3(blocks), dim3(BCE_BLOCK), 0, s, (const T*)x, (const bf16_t*)y, B, ws);
  hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(64), 0, s, ws, blocks, 1.f / (float)B, loss);
  return check_launch("bce_logits_fwd");
}

template <typename T>
static int bce_bwd_launch(const void* x, const void* y, int label_dtype, const float* gout, int64_t B, void* gx,
                          hipStream_t s) {
  const int blocks = stream_grid(B, 256, 1024);
  if (label_dtype == TRS_F32)
    hipLaunchKernelGGL((bce_bwd_kernel<T, float>), dim3(blocks), dim3(256), 0, s, (const T*)x, (const float*)y, gout,
                       1.f / (float)B, B, (T*)gx);
  else
    hipLaunchKernelGGL((bce_bwd_kernel<T, bf16_t>), dim3(blocks), dim3(256), 0, s, (const T*)x, (const bf16_t*)y, gout,
                       1.f / (float)B, B, (T*)gx);
  return check_launch("bce_logits_bwd");
}

}  // namespace trs

using namespace trs;

extern "C" int trs_ctr_logit_fwd(const void* fm, int32_t E, const void* feat, int32_t N, const void* const* extras,
                                 const int64_t* extra_strides, int32_t n_extras, const void* bias, int64_t B,
                                 int32_t dtype, void* out, trs_stream_t stream) {
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "ctr_logit_fwd: dtype %d", dtype);
  TRS_REQUIRE(n_extras >= 0 && n_extras <= HEAD_MAX_EXTRAS, TRS_ESHAPE, "ctr_logit_fwd: at most %d extra columns", HEAD_MAX_EXTRAS);
  TRS_REQUIRE(B >= 0 && E >= 0 && N >= 0, TRS_ESHAPE, "ctr_logit_fwd: negative size");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(out != nullptr && (n_extras == 0 || (extras != nullptr && extra_strides != nullptr)), TRS_EINVAL,
              "ctr_logit_fwd: NULL pointer");
  HeadExtras ex;
  ex.n = n_extras;
  for (int k = 0; k < HEAD_MAX_EXTRAS; ++k) {
    ex.p[k] = k < n_extras ? extras[k] : nullptr;
    ex.stride[k] = k < n_extras ? extra_strides[k] : 0;
    TRS_REQUIRE(k >= n_extras || ex.p[k] != nullptr, TRS_EINVAL, "ctr_logit_fwd: NULL extra column %d", k);
  }
  hipStream_t s = (hipStream_t)stream;
  const int blocks = stream_grid(B, 16, 2048);
  if (dtype == TRS_F32)
    hipLaunchKernelGGL(ctr_logit_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)(E > 0 ? fm : nullptr), E,
                       (const float*)(N > 0 ? feat : nullptr), N, ex, (const float*)bias, B, (float*)out);
  else
    hipLaunchKernelGGL(ctr_logit_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)(E > 0 ? fm : nullptr), E,
                       (const bf16_t*)(N > 0 ? feat : nullptr), N, ex, (const bf16_t*)bias, B, (bf16_t*)out);
  return check_launch("ctr_logit_fwd");
}

extern "C" size_t trs_bce_logits_workspace_bytes(int64_t B) { return (size_t)BCE_MAX_BLOCKS * 4; }

extern "C" int trs_bce_logits_fwd(const void* logits, int32_t dtype, const void* labels, int32_t label_dtype, int64_t B,
                                  float* loss, void* workspace, size_t ws_bytes, trs_stream_t stream) {
  TRS_REQUIRE((dtype == TRS_F32 || dtype == TRS_BF16) && (label_dtype == TRS_F32 || label_dtype == TRS_BF16), TRS_EDTYPE,
              "bce_logits_fwd: dtype %d / %d", dtype, label_dtype);
  TRS_REQUIRE(B > 0, TRS_ESHAPE, "bce_logits_fwd: the mean over an empty batch is undefined");
  TRS_REQUIRE(logits && labels && loss && workspace, TRS_EINVAL, "bce_logits_fwd: NULL pointer");
  TRS_REQUIRE(ws_bytes >= trs_bce_logits_workspace_bytes(B), TRS_EWORKSPACE, "bce_logits_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  return dtype == TRS_F32 ? bce_fwd_launch<float>(logits, labels, label_dtype, B, loss, (float*)workspace, s)
                          : bce_fwd_launch<bf16_t>(logits, labels, label_dtype, B, loss, (float*)workspace, s);
}

extern "C" int trs_bce_logits_bwd(const void* logits, int32_t dtype, const void* labels, int32_t label_dtype,
                                  const float* gout, int64_t B, void* glogits, trs_stream_t stream) {
  TRS_REQUIRE((dtype == TRS_F32 || dtype == TRS_BF16) && (label_dtype == TRS_F32 || label_dtype == TRS_BF16), TRS_EDTYPE,
              "bce_logits_bwd: dtype %d / %d", dtype, label_dtype);
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(logits && labels && glogits, TRS_EINVAL, "bce_logits_bwd: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  return dtype == TRS_F32 ? bce_bwd_launch<float>(logits, labels, label_dtype, gout, B, glogits, s)
                          : bce_bwd_launch<bf16_t>(logits, labels, label_dtype, gout, B, glogits, s);
}
