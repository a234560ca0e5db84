// What the matrix pipe sustains as a function of the operand DATA (round 4).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
// v_mfma_f32_32x32x16_{bf16,f16}, 16 in-place accumulator tiles per wave, operands resident in registers (no memory
// traffic at all), one wave per SIMD (grid 256 x 256 threads) or two.  Operand contents: zeros, small integers (what
// mfma_rate.hip used), or random values ~N(0,1).  The clock the launch ran at = shader cycles (s_memtime) / wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ float gauss(unsigned seed) {          // sum of 4 uniforms, roughly N(0,1)
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += (hash32(seed * 4u + i) & 0xffffff) * (1.f / 16777216.f);
  return (s - 2.f) * 1.7320508f;
}

template <bool F16, int DATA>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  u32x4 a[8], b[2];
  for (int f = 0; f < 10; ++f) {
    unsigned w[4];
    for (int q = 0; q < 4; ++q) {
      float v0, v1;
      const unsigned seed = ((blockIdx.x * 256 + threadIdx.x) * 10 + f) * 4 + q;
      if (DATA == 0) { v0 = 0.f; v1 = 0.f; }
      else if (DATA == 1) { v0 = (float)((threadIdx.x + q + 3 * f) & 7); v1 = (float)(q & 1); }
      else { v0 = gauss(2 * seed); v1 = gauss(2 * seed + 1); }
      if (F16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        h2 h = {(_Float16)v0, (_Float16)v1};
        w[q] = __builtin_bit_cast(unsigned, h);
      } else {
        typedef __attribute__((ext_vector_type(2))) __bf16 b2;
        b2 h = {(__bf16)v0, (__bf16)v1};
        w[q] = __builtin_bit_cast(unsigned, h);
      }
    }
    if (f < 8) a[f] = u32x4{w[0], w[1], w[2], w[3]};
    else b[f - 8] = u32x4{w[0], w[1], w[2], w[3]};
  }
  const long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (F16) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i]), "v"(b[i & 1]));
      } else {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i]), "v"(b[i & 1]));
      }
    }
  }
  const long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = t1 - t0; }
}

template <bool F16, int DATA>
void run(float* out, long long* clk, int grid, int iters, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<F16, DATA>), dim3(grid), dim3(256), 0, 0, out, clk, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  long long h[2]; (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  int khz = 100000; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  const double mfmas = (double)grid * 4 * iters * 8;
  const double ghz = (double)h[0] / ((double)h[1] / (khz * 1e3)) / 1e9;
  printf("%-40s grid %4d  %.3f ms  %5.0f TFLOP/s  shader clock %.2f GHz  pipe busy %.0f %%\n", name, grid, best,
         mfmas * 32768.0 / (best * 1e-3) / 1e12, ghz, 100.0 * (double)iters * 8 * 32 * (grid > 256 ? 2 : 1) / (double)h[0]);
}
int main(int argc, char** argv) {
  float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
  long long* clk; (void)hipMalloc(&clk, 4096 * 16);
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  for (int grid : {256, 512}) {
    run<false, 0>(out, clk, grid, iters, "bf16 32x32x16, zeros");
    run<false, 1>(out, clk, grid, iters, "bf16 32x32x16, small integers");
    run<false, 2>(out, clk, grid, iters, "bf16 32x32x16, random N(0,1)");
    run<true, 0>(out, clk, grid, iters, "f16  32x32x16, zeros");
    run<true, 1>(out, clk, grid, iters, "f16  32x32x16, small integers");
    run<true, 2>(out, clk, grid, iters, "f16  32x32x16, random N(0,1)");
  }
  return 0;
}
