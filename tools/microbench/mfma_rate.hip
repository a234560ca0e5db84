// Sustained rate of back-to-back v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x16_bf16 on every SIMD (4 waves per SIMD, 8
// independent accumulators per wave, no memory traffic): what the matrix pipe delivers at the clock the chip settles on.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  s16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int i = 0; i < 4; ++i) { a4[i] = (short)(threadIdx.x + i); b4[i] = (short)(i + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, out, iters);
      else hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mfmas = 1024.0 * 4 * iters * 8;
      const double flop = mfmas * (mode == 0 ? 16384.0 : 8192.0);
      printf("%s: %.3f ms, %.2f ns per MFMA and SIMD, %.0f TFLOP/s\n", mode == 0 ? "16x16x32 bf16" : "16x16x16 bf16", ms,
             ms * 1e6 / (mfmas / (256.0 * 4)), flop / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
