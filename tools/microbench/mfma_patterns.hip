#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// MODE 0: in place (VGPR).  1: ping-pong dst != srcC, disjoint VGPRs.  2: in place, AGPR accumulators.
// 3: builtin (compiler's choice).  4: in place VGPR with 2 independent VALU ops between MFMAs
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[8], acc2[8];
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i] = acc[i]; }
  bf16x8 a[8], b;
  for (int j = 0; j < 8; ++j)
    for (int i = 0; i < 8; ++i) a[j][i] = (__bf16)(float)((threadIdx.x + i + 3 * j) & 7);
  for (int i = 0; i < 8; ++i) b[i] = (__bf16)(float)(i & 1);
  float v0 = threadIdx.x, v1 = 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b));
      if (MODE == 1) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc2[i]) : "v"(a[i]), "v"(b), "v"(acc[i]));
      }
      if (MODE == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i]), "v"(b));
      if (MODE == 3) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i], 0, 0, 0);
      if (MODE == 4) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %1, %1, %0, %0" : "+v"(v0), "+v"(v1));
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc[i]) : "v"(a[i]), "v"(b), "v"(acc2[i]));
    }
  }
  float s = v0 + v1;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(float* out, int grid, int iters, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mfmas = (double)grid * 4 * iters * 8 * (MODE == 1 ? 2 : 1);
  printf("%-44s %.3f ms  %.0f TFLOP/s\n", name, best, mfmas * 16384.0 / (best * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
  float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
  const int grid = argc > 1 ? atoi(argv[1]) : 1024, iters = 10000;
  run<0>(out, grid, iters, "in place, VGPR accumulators");
  run<1>(out, grid, iters, "ping-pong dst != srcC (disjoint VGPRs)");
  run<2>(out, grid, iters, "in place, AGPR accumulators");
  run<3>(out, grid, iters, "builtin, hipcc's allocation");
  run<4>(out, grid, iters, "in place VGPR + 2 VALU per MFMA");
  return 0;
}
