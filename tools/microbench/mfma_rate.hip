// Sustained rate of back-to-back v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x16_bf16 on every SIMD (4 waves per SIMD, 8
// independent accumulators per wave, no memory traffic): what the matrix pipe delivers at the clock the chip settles on.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// 32x32x16 bf16: 16 accumulator registers per MFMA, half the instructions per FLOP of 16x16x32
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a[4], b;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 8; ++i) a[j][i] = (__bf16)(float)((threadIdx.x + i + 3 * j) & 7);
  for (int i = 0; i < 8; ++i) b[i] = (__bf16)(float)(i & 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b));
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // every accumulator gets its OWN operands: with shared ones the eight chains are identical and hipcc folds them into
  // one dependent chain over rotated registers (the first version of this benchmark measured that chain, not the pipe)
  bf16x8 a[8], b;
  s16x4 a4[8], b4;
  for (int j = 0; j < 8; ++j) {
    for (int i = 0; i < 8; ++i) a[j][i] = (__bf16)(float)((threadIdx.x + i + 3 * j) & 7);
    for (int i = 0; i < 4; ++i) a4[j][i] = (short)(0x3f80 + ((threadIdx.x + i + j) & 3));
  }
  for (int i = 0; i < 8; ++i) b[i] = (__bf16)(float)(i & 1);
  for (int i = 0; i < 4; ++i) b4[i] = (short)0x3f80;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // in-place accumulators through asm: hipcc's own allocation rotates them (dst = src registers + 2)
      if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b));
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4[i], b4, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  const int iters = 20000;
  // waves per SIMD = grid / 256 (one 4-wave workgroup per CU and wave slot): 1024 = 4 per SIMD (default), 256 = 1
  const int grid = argc > 1 ? atoi(argv[1]) : 1024;
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters);
      else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mfmas = (double)grid * 4 * iters * 8;
      const double flop = mfmas * (mode == 0 ? 16384.0 : 8192.0);
      printf("%s: %.3f ms, %.2f ns per MFMA and SIMD, %.0f TFLOP/s\n", mode == 0 ? "16x16x32 bf16" : "16x16x16 bf16", ms,
             ms * 1e6 / (mfmas / (256.0 * 4)), flop / (ms * 1e-3) / 1e12);
    }
  }
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mfmas = (double)grid * 4 * iters * 4;
      printf("32x32x16 bf16: %.3f ms, %.2f ns per MFMA and SIMD, %.0f TFLOP/s\n", ms, ms * 1e6 / (mfmas / (256.0 * 4)),
             mfmas * 32768.0 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
