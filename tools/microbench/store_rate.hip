// store issue rate per CU: pattern 0 = 16 rows x 64 B per wave store (row stride 832 B), pattern 1 = 1 KB contiguous
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int PAT, int NT>
__global__ __launch_bounds__(512) void k(char* base, int iters, size_t region) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every workgroup owns a 128 x 832 B tile (106 KB), rewritten `iters` times
  char* tile = base + ((size_t)blockIdx.x * 128 * 832) % region;
  u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      size_t off;
      if (PAT == 0) {          // wave w: rows 16w..16w+15, piece (lane>>4) of k-slice j
        off = (size_t)(16 * wave + (lane & 15)) * 832 + j * 64 + (lane >> 4) * 16;
      } else {                 // contiguous: store index s = j*8 + wave, 1 KB each
        off = ((size_t)(j * 8 + wave) * 64 + lane) * 16;
      }
      if (NT == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(tile + off), "v"(v) : "memory");
      else if (NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(tile + off), "v"(v) : "memory");
      else if (NT == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(tile + off), "v"(v) : "memory");
      else if (NT == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(tile + off), "v"(v) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(tile + off), "v"(v) : "memory");
    }
    v.x += 1;
  }
}
int main() {
  const size_t region = (size_t)256 * 128 * 832;      // 27 MB: one tile per CU
  char* d;
  hipMalloc(&d, region + (1 << 20));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200;
  for (int pat = 0; pat < 2; ++pat)
    for (int nt = 0; nt < 5; ++nt)
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
#define LAUNCH(P, N) if (pat == P && nt == N) k<P, N><<<256, 512>>>(d, iters, region);
        LAUNCH(0, 0) LAUNCH(0, 1) LAUNCH(0, 2) LAUNCH(0, 3) LAUNCH(0, 4)
        LAUNCH(1, 0) LAUNCH(1, 1) LAUNCH(1, 2) LAUNCH(1, 3) LAUNCH(1, 4)
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 256.0 * iters * 13 * 8 * 1024;
        printf("pattern %d nt %d: %.3f ms  %.1f GB/s total  %.1f B/ns/CU\n", pat, nt, ms, bytes / ms / 1e6, bytes / 256 / (ms * 1e6));
      }
  return 0;
}
