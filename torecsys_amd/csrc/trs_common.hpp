// Shared device/host helpers for libtrs_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/trs_abi.h"

namespace trs {

// ---------------------------------------------------------------- error reporting (host)
char* err_buf();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);
// zero-fill by a kernel (gather.hip): hipMemsetAsync is not used anywhere in the library (memset nodes in hipGraph replays)
int zero_bytes(void* p, size_t bytes, hipStream_t s);

#define TRS_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) return ::trs::fail((code), __VA_ARGS__); \
  } while (0)

// ---------------------------------------------------------------- value types
struct bf16_t {
  uint16_t v;
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) { return __uint_as_float(bits16 << 16); }

// round-to-nearest-even (same rounding as at::BFloat16); lowers to the gfx950 hardware convert
// v_cvt_pk_bf16_f32 -- the bit-twiddling form costs ~5 VALU ops per element and made bf16 epilogues
// VALU-bound.
typedef __attribute__((ext_vector_type(2))) __bf16 trs_bf16x2;
__device__ __forceinline__ uint32_t f32x2_to_bf16x2_bits(float lo, float hi) {
  trs_bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)f);
}

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return bf16_bits_to_f32(x.v); }
template <typename T>
__device__ __forceinline__ T from_f32(float x);
template <>
__device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return bf16_t{(uint16_t)f32_to_bf16_bits(x)}; }

// A 16-byte vector of T: VE = 4 floats or 8 bf16.
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  static constexpr int VE = 4;
  static __device__ __forceinline__ void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
    f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <>
struct Vec16<bf16_t> {
  static constexpr int VE = 8;
  static __device__ __forceinline__ void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    uint4 u;
    u.x = f32x2_to_bf16x2_bits(f[0], f[1]);
    u.y = f32x2_to_bf16x2_bits(f[2], f[3]);
    u.z = f32x2_to_bf16x2_bits(f[4], f[5]);
    u.w = f32x2_to_bf16x2_bits(f[6], f[7]);
    return u;
  }
};

// ---------------------------------------------------------------- (i<j) field pairs
// pair index p = i*(2N-i-1)/2 + (j-i-1), lexicographic in (i, j) -- the order of the reference's row_idx / col_idx
// lists (layers/ctr/inner_product_network.py:43-52).
__device__ __forceinline__ void pair_ij(int p, int N, int* i_out, int* j_out) {
  const float d = (float)(2 * N - 1);
  int i = (int)((d - sqrtf(fmaxf(d * d - 8.f * (float)p, 0.f))) * 0.5f);
  if (i < 0) i = 0;
  if (i > N - 2) i = N - 2;
  while (i > 0 && i * (2 * N - i - 1) / 2 > p) --i;
  while (i < N - 2 && (i + 1) * (2 * N - i - 2) / 2 <= p) ++i;
  *i_out = i;
  *j_out = p - i * (2 * N - i - 1) / 2 + i + 1;
}
__host__ __device__ inline int pair_index_of(int i, int j, int N) { return i * (2 * N - i - 1) / 2 + j - i - 1; }

// Conflict-free pair schedule (round-robin tournament, circle method): the NC2 pairs are split into R rounds of at
// most H pairs that share no field, so the units of a workgroup can add into per-field LDS accumulators with plain
// read-modify-writes inside a round (ds_add_f32 on one address runs at ~1 lane per clock) and synchronise between
// rounds.  Entry (r, k) = (i << 16) | j, or -1 (the dummy player of an odd N).
__host__ __device__ inline int sched_rounds(int N) { return (N & 1) ? N : N - 1; }
__host__ __device__ inline int sched_width(int N) { return (N + 1) / 2; }
__device__ __forceinline__ int sched_entry(int r, int k, int N) {
  const int M = (N & 1) ? N + 1 : N;          // even number of players; player M-1 is a dummy when N is odd
  const int R = M - 1;
  int a, b;
  if (k == 0) {
    a = r; b = M - 1;
  } else {
    a = (r + k) % R; b = (r - k + R) % R;
  }
  if (a >= N || b >= N) return -1;
  return ((a < b ? a : b) << 16) | (a < b ? b : a);
}

// ---------------------------------------------------------------- index loads
template <typename IdxT>
__device__ __forceinline__ int64_t load_row_id(const IdxT* __restrict__ idx, const int64_t* __restrict__ offsets,
                                               int64_t p, int n) {
  int64_t r = (int64_t)idx[p];
  if (offsets) r += offsets[n];
  return r;
}

inline int ceil_div_i(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
inline bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
inline int dtype_size(int dtype) { return dtype == TRS_F32 ? 4 : 2; }
inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// grid for memory-bound grid-stride kernels: enough blocks to fill 256 CUs several times over
inline int stream_grid(int64_t work_items, int block, int max_blocks = 256 * 16) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

// Workgroups of `kernel` (threads per block, dynamic LDS bytes) that are resident on the device at once.  Kernels whose
// blocks walk equal shares of the work (a wave or block owns its share for the whole launch) size their grid from this:
// a grid a little above it runs as a full round plus a nearly empty one -- 1024 blocks on 768 slots take two rounds for
// 1.33 rounds of work.
inline int resident_blocks(const void* kernel, int threads, size_t dyn_lds) {
  int dev = 0, cus = 256, blocks = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, threads, dyn_lds) != hipSuccess || blocks < 1)
    blocks = 1;
  return cus * blocks;
}

// Flat element counters are int64 in the kernels' loops, but almost always fit 32 bits: a runtime-divisor 64-bit
// division is ~80 emulated instructions per element, the 32-bit one a float reciprocal + fix-up.  ``fits32`` is uniform
// over the launch (total element count < 2^32).
__device__ __forceinline__ int64_t udiv_fast(int64_t a, int d, bool fits32) {
  return fits32 ? (int64_t)((unsigned)a / (unsigned)d) : a / d;
}
// 16-byte streaming load: data that is read exactly once (the (B,N,E) block gradient in the bucket walk) should not
// displace what IS re-read -- the embedding table in the 256 MiB Infinity Cache: with plain loads of the 327 MB gradient the
// next step's lookup found the table evicted (134 us in the DeepFM step), with these it runs at 112-116 us
// (profiles/r05_logs/ab_scatter_nt_loads.txt)
__device__ __forceinline__ uint4 load_stream(const uint4* src) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
  return make_uint4(w.x, w.y, w.z, w.w);
}
// 16-byte streaming store: the destination is consumed by a later kernel, keep L2 for data that is re-read
__device__ __forceinline__ void store_stream(uint4* dst, const uint4& v) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const u32x4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(dst));
}

}  // namespace trs
