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

}  // namespace trs
