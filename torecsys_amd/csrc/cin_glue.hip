// CIN layer glue on the channels-last activations (B, E, C) bf16:  BatchNorm1d (batch or running statistics) +
// ReLU + the direct / hidden split + the sum over E of the direct half, layers/ctr/compress_interaction_network.py
// :137-181 (Batchnorm, Activation, chunk(2, dim=1), cat(direct).sum over E).
// The ATen sequence costs ~5.6 ms per layer at B = 65 536, E = 64, C = 256 (statistics, transform, relu, a copy of
// the hidden half, the pooled reduction and two casts: 7 passes over a 2.1 GB tensor) and ~6 ms in the backward;
// here the forward is two passes (column statistics; normalise + relu + write the hidden half + pool the direct
// half in registers -- the direct half is never written) and the backward two (reduce dgamma/dbeta; apply).
// Thread layout: 256 threads = (256 / vpr) rows x vpr 16-byte column vectors, vpr = C / 8; a workgroup walks one
// sample (E rows) at a time, so the pooled sums are plain register accumulations.
#include <algorithm>

#include "trs_common.hpp"

namespace trs {

constexpr int GLUE_THREADS = 256;

struct GlueIdx {
  int vpr, rpp, v, rr, c0;
};
__device__ __forceinline__ GlueIdx glue_idx(int C) {
  GlueIdx g;
  g.vpr = C / 8;
  g.rpp = GLUE_THREADS / g.vpr;
  g.v = threadIdx.x % g.vpr;
  g.rr = threadIdx.x / g.vpr;
  g.c0 = g.v * 8;
  return g;
}

// reduce the per-row-group partial vectors of a workgroup: part[k] (8 columns per thread) -> out[c] for rr == 0
__device__ __forceinline__ void glue_block_reduce(float* lds, const GlueIdx& g, int C, const float* part, float* out8) {
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) lds[g.rr * C + g.c0 + k] = part[k];
  __syncthreads();
  if (g.rr == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float s = 0.f;
      for (int r = 0; r < g.rpp; ++r) s += lds[r * C + g.c0 + k];
      out8[k] = s;
    }
  }
}

// pass 1: per-workgroup column sums / sums of squares  ->  partial[blk][2][C]
__global__ __launch_bounds__(GLUE_THREADS) void glue_stats_kernel(const bf16_t* __restrict__ y, int64_t B, int E, int C,
                                                                  float* __restrict__ partial) {
  extern __shared__ float lds[];
  const GlueIdx g = glue_idx(C);
  float s[8], ss[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s[k] = 0.f; ss[k] = 0.f; }
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint4* rows = reinterpret_cast<const uint4*>(y + b * E * (int64_t)C);
    for (int e = g.rr; e < E; e += g.rpp) {
      float x[8];
      Vec16<bf16_t>::unpack(rows[(int64_t)e * g.vpr + g.v], x);
#pragma unroll
      for (int k = 0; k < 8; ++k) { s[k] += x[k]; ss[k] = fmaf(x[k], x[k], ss[k]); }
    }
  }
  float o[8];
  glue_block_reduce(lds, g, C, s, o);
  if (g.rr == 0)
#pragma unroll
    for (int k = 0; k < 8; ++k) partial[((size_t)blockIdx.x * 2 + 0) * C + g.c0 + k] = o[k];
  glue_block_reduce(lds, g, C, ss, o);
  if (g.rr == 0)
#pragma unroll
    for (int k = 0; k < 8; ++k) partial[((size_t)blockIdx.x * 2 + 1) * C + g.c0 + k] = o[k];
}

// pass 2: z = relu(y * scale + shift);  hidden[b,e,c-Hs] = z for c >= Hs;  pooled[b,c] = sum_e z for c < D
__global__ __launch_bounds__(GLUE_THREADS) void glue_apply_fwd_kernel(const bf16_t* __restrict__ y,
                                                                      const float* __restrict__ scale,
                                                                      const float* __restrict__ shift, int64_t B, int E,
                                                                      int C, int D, int Hs, bf16_t* __restrict__ hidden,
                                                                      bf16_t* __restrict__ pooled) {
  extern __shared__ float lds[];
  const GlueIdx g = glue_idx(C);
  float a[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = scale[g.c0 + k]; sh[k] = shift[g.c0 + k]; }
  const bool is_hidden = g.c0 >= Hs, is_direct = g.c0 < D;
  const int HW = C - Hs;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint4* rows = reinterpret_cast<const uint4*>(y + b * E * (int64_t)C);
    float pool[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pool[k] = 0.f;
    for (int e = g.rr; e < E; e += g.rpp) {
      float x[8];
      Vec16<bf16_t>::unpack(rows[(int64_t)e * g.vpr + g.v], x);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        x[k] = fmaxf(fmaf(x[k], a[k], sh[k]), 0.f);
        pool[k] += x[k];
      }
      if (is_hidden)
        *reinterpret_cast<uint4*>(hidden + (b * E + e) * (int64_t)HW + (g.c0 - Hs)) = Vec16<bf16_t>::pack(x);
    }
    float o[8];
    glue_block_reduce(lds, g, C, pool, o);
    if (g.rr == 0 && is_direct) *reinterpret_cast<uint4*>(pooled + b * (int64_t)D + g.c0) = Vec16<bf16_t>::pack(o);
  }
}

// gradient wrt z before the relu mask: pooled gradient broadcast over e (c < D) + hidden gradient (c >= Hs)
__device__ __forceinline__ void glue_gz(const bf16_t* __restrict__ g_hidden, const float* gp8, bool is_hidden,
                                        bool is_direct, int64_t row, int HW, int hc, float* gz) {
#pragma unroll
  for (int k = 0; k < 8; ++k) gz[k] = is_direct ? gp8[k] : 0.f;
  if (is_hidden && g_hidden != nullptr) {
    float gh[8];
    Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(g_hidden + row * (int64_t)HW + hc), gh);
#pragma unroll
    for (int k = 0; k < 8; ++k) gz[k] += gh[k];
  }
}

// backward pass 1: dbeta[c] = sum gz*[z>0], dgamma[c] = sum gz*[z>0]*xhat  ->  partial[blk][2][C]
__global__ __launch_bounds__(GLUE_THREADS) void glue_bwd_reduce_kernel(
    const bf16_t* __restrict__ y, const bf16_t* __restrict__ g_hidden, const bf16_t* __restrict__ g_pooled,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, int64_t B, int E, int C, int D, int Hs, float* __restrict__ partial) {
  extern __shared__ float lds[];
  const GlueIdx g = glue_idx(C);
  float a[8], sh[8], mu[8], is[8], db[8], dg[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = scale[g.c0 + k]; sh[k] = shift[g.c0 + k]; mu[k] = mean[g.c0 + k]; is[k] = invstd[g.c0 + k];
    db[k] = 0.f; dg[k] = 0.f;
  }
  const bool is_hidden = g.c0 >= Hs, is_direct = g.c0 < D;
  const int HW = C - Hs;
  // channels that receive no gradient at all (neither pooled nor fed to a next layer: the last CIN layer's hidden half):
  // their sums are exact zeros, y is not read for them
  const bool dead = !(is_direct && g_pooled != nullptr) && !(is_hidden && g_hidden != nullptr);
  for (int64_t b = blockIdx.x; b < B && !dead; b += gridDim.x) {
    const uint4* rows = reinterpret_cast<const uint4*>(y + b * E * (int64_t)C);
    float gp8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) gp8[k] = 0.f;
    if (is_direct && g_pooled != nullptr)
      Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(g_pooled + b * (int64_t)D + g.c0), gp8);
    for (int e = g.rr; e < E; e += g.rpp) {
      float x[8], gz[8];
      Vec16<bf16_t>::unpack(rows[(int64_t)e * g.vpr + g.v], x);
      glue_gz(g_hidden, gp8, is_hidden, is_direct, b * E + e, HW, g.c0 - Hs, gz);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float m = fmaf(x[k], a[k], sh[k]) > 0.f ? gz[k] : 0.f;
        db[k] += m;
        dg[k] = fmaf(m, (x[k] - mu[k]) * is[k], dg[k]);
      }
    }
  }
  float o[8];
  glue_block_reduce(lds, g, C, db, o);
  if (g.rr == 0)
#pragma unroll
    for (int k = 0; k < 8; ++k) partial[((size_t)blockIdx.x * 2 + 0) * C + g.c0 + k] = o[k];
  glue_block_reduce(lds, g, C, dg, o);
  if (g.rr == 0)
#pragma unroll
    for (int k = 0; k < 8; ++k) partial[((size_t)blockIdx.x * 2 + 1) * C + g.c0 + k] = o[k];
}

// backward pass 2: gy = scale * (gz*[z>0] - c1 - xhat * c2)     c1 = dbeta / R, c2 = dgamma / R (0 with running stats)
__global__ __launch_bounds__(GLUE_THREADS) void glue_bwd_apply_kernel(
    const bf16_t* __restrict__ y, const bf16_t* __restrict__ g_hidden, const bf16_t* __restrict__ g_pooled,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ c1, const float* __restrict__ c2, int64_t B, int E, int C,
    int D, int Hs, bf16_t* __restrict__ gy) {
  const GlueIdx g = glue_idx(C);
  float a[8], sh[8], mu[8], is[8], k1[8], k2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = scale[g.c0 + k]; sh[k] = shift[g.c0 + k]; mu[k] = mean[g.c0 + k]; is[k] = invstd[g.c0 + k];
    k1[k] = c1[g.c0 + k]; k2[k] = c2[g.c0 + k];
  }
  const bool is_hidden = g.c0 >= Hs, is_direct = g.c0 < D;
  const int HW = C - Hs;
  const bool dead = !(is_direct && g_pooled != nullptr) && !(is_hidden && g_hidden != nullptr);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint4* rows = reinterpret_cast<const uint4*>(y + b * E * (int64_t)C);
    uint4* out = reinterpret_cast<uint4*>(gy + b * E * (int64_t)C);
    float gp8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) gp8[k] = 0.f;
    if (is_direct && g_pooled != nullptr)
      Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(g_pooled + b * (int64_t)D + g.c0), gp8);
    for (int e = g.rr; e < E; e += g.rpp) {
      float x[8], gz[8];
      if (dead) {          // no gradient reaches these channels (c1 = c2 = 0 for them too): zeros, y not read
        out[(int64_t)e * g.vpr + g.v] = make_uint4(0, 0, 0, 0);
        continue;
      }
      Vec16<bf16_t>::unpack(rows[(int64_t)e * g.vpr + g.v], x);
      glue_gz(g_hidden, gp8, is_hidden, is_direct, b * E + e, HW, g.c0 - Hs, gz);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float m = fmaf(x[k], a[k], sh[k]) > 0.f ? gz[k] : 0.f;
        gz[k] = a[k] * (m - k1[k] - (x[k] - mu[k]) * is[k] * k2[k]);
      }
      out[(int64_t)e * g.vpr + g.v] = Vec16<bf16_t>::pack(gz);
    }
  }
}

// ---- variants that ALSO emit a channels-first copy of their output (the weight-gradient kernel trs_cin_dw reads
// (B, channels, E) operands: without the copy PyTorch transposes the 1-2 GB tensors once per layer and step).
// Requires E == 8 * rpp (E = 64 with C = 256): a thread then owns 8 CONSECUTIVE e for its 8 channels, i.e. an 8 x 8
// register tile whose transpose is eight 16-byte vectors along e; they go through an XOR-swizzled LDS tile
// [channel][E] so that the global writes are whole rows.
__device__ __forceinline__ uint4 glue_pack_col(float (*t)[8], int k) {
  float col[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) col[j] = t[j][k];
  return Vec16<bf16_t>::pack(col);
}

__global__ __launch_bounds__(GLUE_THREADS) void glue_apply_fwd_cf_kernel(const bf16_t* __restrict__ y,
                                                                         const float* __restrict__ scale,
                                                                         const float* __restrict__ shift, int64_t B,
                                                                         int E, int C, int D, int Hs,
                                                                         bf16_t* __restrict__ hidden,
                                                                         bf16_t* __restrict__ hidden_cf,
                                                                         bf16_t* __restrict__ pooled) {
  extern __shared__ float lds[];
  const GlueIdx g = glue_idx(C);
  float a[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = scale[g.c0 + k]; sh[k] = shift[g.c0 + k]; }
  const bool is_hidden = g.c0 >= Hs, is_direct = g.c0 < D;
  const int HW = C - Hs;
  const int e0 = g.rr * 8;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint4* rows = reinterpret_cast<const uint4*>(y + b * E * (int64_t)C);
    uint4 raw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[j] = rows[(int64_t)(e0 + j) * g.vpr + g.v];
    float t[8][8], pool[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pool[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      Vec16<bf16_t>::unpack(raw[j], t[j]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        t[j][k] = fmaxf(fmaf(t[j][k], a[k], sh[k]), 0.f);
        pool[k] += t[j][k];
      }
      if (is_hidden)
        *reinterpret_cast<uint4*>(hidden + (b * E + e0 + j) * (int64_t)HW + (g.c0 - Hs)) = Vec16<bf16_t>::pack(t[j]);
    }
    // channels-first copy through an LDS tile [channel][E (+8 pad)] so that the global writes are whole 128-byte rows
    // (16-byte pieces written straight from the registers reach only ~1.9 TB/s)
    bf16_t* tile = reinterpret_cast<bf16_t*>(lds + GLUE_THREADS * 8);
    const int TS = E + 8;
    __syncthreads();
    if (is_hidden) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        *reinterpret_cast<uint4*>(tile + (g.c0 - Hs + k) * TS + (e0 ^ ((((g.c0 - Hs) >> 3) & (g.rpp - 1)) << 3))) = glue_pack_col(t, k);
    }
    __syncthreads();
    {
      const int vpe = E / 8;                       // 16-byte vectors per channel row
      uint4* dst = reinterpret_cast<uint4*>(hidden_cf + b * HW * (int64_t)E);
      for (int i = threadIdx.x; i < HW * vpe; i += GLUE_THREADS) {
        const int ch = i / vpe, ve = i - ch * vpe;
        dst[i] = *reinterpret_cast<const uint4*>(tile + ch * TS + ((ve ^ ((ch >> 3) & (vpe - 1))) << 3));
      }
    }
    float o[8];
    glue_block_reduce(lds, g, C, pool, o);
    if (g.rr == 0 && is_direct) *reinterpret_cast<uint4*>(pooled + b * (int64_t)D + g.c0) = Vec16<bf16_t>::pack(o);
  }
}

__global__ __launch_bounds__(GLUE_THREADS) void glue_bwd_apply_cf_kernel(
    const bf16_t* __restrict__ y, const bf16_t* __restrict__ g_hidden, const bf16_t* __restrict__ g_pooled,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ c1, const float* __restrict__ c2, int64_t B, int E, int C,
    int D, int Hs, bf16_t* __restrict__ gy, bf16_t* __restrict__ gy_cf, float* __restrict__ colsum_partial) {
  const GlueIdx g = glue_idx(C);
  float a[8], sh[8], mu[8], is[8], k1[8], k2[8];
  float cs[8];                       // column sums of gy (the conv-bias gradient), folded into this pass
#pragma unroll
  for (int k = 0; k < 8; ++k) cs[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = scale[g.c0 + k]; sh[k] = shift[g.c0 + k]; mu[k] = mean[g.c0 + k]; is[k] = invstd[g.c0 + k];
    k1[k] = c1[g.c0 + k]; k2[k] = c2[g.c0 + k];
  }
  const bool is_hidden = g.c0 >= Hs, is_direct = g.c0 < D;
  const int HW = C - Hs;
  const bool dead = !(is_direct && g_pooled != nullptr) && !(is_hidden && g_hidden != nullptr);      // see glue_bwd_reduce_kernel
  const int e0 = g.rr * 8;
  extern __shared__ float lds[];
  bf16_t* tile = reinterpret_cast<bf16_t*>(lds);       // [C][E + 8]: see glue_apply_fwd_cf_kernel
  const int TS = E + 8;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint4* rows = reinterpret_cast<const uint4*>(y + b * E * (int64_t)C);
    uint4* out = reinterpret_cast<uint4*>(gy + b * E * (int64_t)C);
    float gp8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) gp8[k] = 0.f;
    if (is_direct && g_pooled != nullptr)
      Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(g_pooled + b * (int64_t)D + g.c0), gp8);
    uint4 raw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[j] = dead ? make_uint4(0, 0, 0, 0) : rows[(int64_t)(e0 + j) * g.vpr + g.v];
    float t[8][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x[8];
      Vec16<bf16_t>::unpack(raw[j], x);
      glue_gz(g_hidden, gp8, is_hidden, is_direct, b * E + e0 + j, HW, g.c0 - Hs, t[j]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float m = fmaf(x[k], a[k], sh[k]) > 0.f ? t[j][k] : 0.f;
        t[j][k] = dead ? 0.f : a[k] * (m - k1[k] - (x[k] - mu[k]) * is[k] * k2[k]);
      }
      const uint4 pk = Vec16<bf16_t>::pack(t[j]);
      out[(int64_t)(e0 + j) * g.vpr + g.v] = pk;
      if (colsum_partial != nullptr) {        // sums of the values as stored (bf16), like a pass over gy would see them
        float rv[8];
        Vec16<bf16_t>::unpack(pk, rv);
#pragma unroll
        for (int k = 0; k < 8; ++k) cs[k] += rv[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k)
      *reinterpret_cast<uint4*>(tile + (g.c0 + k) * TS + (e0 ^ (((g.c0 >> 3) & (g.rpp - 1)) << 3))) = glue_pack_col(t, k);
    __syncthreads();
    {
      const int vpe = E / 8;
      uint4* dst = reinterpret_cast<uint4*>(gy_cf + b * C * (int64_t)E);
      for (int i = threadIdx.x; i < C * vpe; i += GLUE_THREADS) {
        const int ch = i / vpe, ve = i - ch * vpe;
        dst[i] = *reinterpret_cast<const uint4*>(tile + ch * TS + ((ve ^ ((ch >> 3) & (vpe - 1))) << 3));
      }
    }
  }
  if (colsum_partial != nullptr) {
    float o[8];
    glue_block_reduce(lds, g, C, cs, o);     // begins with a barrier: every reader of the tile is done
    if (g.rr == 0)
#pragma unroll
      for (int k = 0; k < 8; ++k) colsum_partial[(size_t)blockIdx.x * C + g.c0 + k] = o[k];
  }
}

static int glue_grid(int64_t B) { return (int)std::min<int64_t>(B, 1024); }
static bool glue_cf_ok(int C, int E) { return C % 8 == 0 && C / 8 <= GLUE_THREADS && E == 8 * (GLUE_THREADS / (C / 8)); }
static bool glue_shape_ok(int C, int D, int Hs) {
  const int vpr = C / 8;
  return C % 8 == 0 && vpr >= 1 && vpr <= GLUE_THREADS && GLUE_THREADS % vpr == 0 && D % 8 == 0 && Hs % 8 == 0 &&
         D >= 0 && D <= C && Hs >= 0 && Hs <= C;
}

}  // namespace trs

using namespace trs;

#define TRS_GLUE_COMMON(name)                                                                              \
  TRS_REQUIRE(B >= 0 && E > 0 && C > 0, TRS_EINVAL, name ": bad size");                                    \
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, name ": bf16 only (dtype %d)", dtype);                        \
  TRS_REQUIRE(glue_shape_ok(C, D, Hs), TRS_ESHAPE,                                                         \
              name ": C = %d must be 8 x a divisor of 256, D = %d and Hs = %d multiples of 8 within C", C, D, Hs)

extern "C" int32_t trs_cin_glue_blocks(int64_t B) { return B > 0 ? glue_grid(B) : 0; }

/* partial: (trs_cin_glue_blocks(B), 2, C) fp32 = per-workgroup column sums and sums of squares of y (B,E,C) */
extern "C" int trs_cin_glue_stats(const void* y, int64_t B, int32_t E, int32_t C, int32_t dtype, float* partial,
                                  trs_stream_t stream) {
  const int D = 0, Hs = 0;
  TRS_GLUE_COMMON("cin_glue_stats");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(y && partial && aligned16(y), TRS_EINVAL, "cin_glue_stats: NULL or unaligned pointer");
  hipLaunchKernelGGL(glue_stats_kernel, dim3(glue_grid(B)), dim3(GLUE_THREADS), (size_t)GLUE_THREADS * 8 * 4,
                     (hipStream_t)stream, (const bf16_t*)y, B, E, C, partial);
  return check_launch("cin_glue_stats");
}

extern "C" int trs_cin_glue_fwd(const void* y, const float* scale, const float* shift, int64_t B, int32_t E, int32_t C,
                                int32_t D, int32_t Hs, int32_t dtype, void* hidden, void* pooled, trs_stream_t stream) {
  TRS_GLUE_COMMON("cin_glue_fwd");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(y && scale && shift && (hidden || Hs == C) && (pooled || D == 0), TRS_EINVAL, "cin_glue_fwd: NULL pointer");
  TRS_REQUIRE(aligned16(y) && aligned16(hidden) && aligned16(pooled), TRS_EALIGN, "cin_glue_fwd: 16-byte alignment");
  hipLaunchKernelGGL(glue_apply_fwd_kernel, dim3(glue_grid(B)), dim3(GLUE_THREADS), (size_t)GLUE_THREADS * 8 * 4,
                     (hipStream_t)stream, (const bf16_t*)y, scale, shift, B, E, C, D, Hs, (bf16_t*)hidden,
                     (bf16_t*)pooled);
  return check_launch("cin_glue_fwd");
}

extern "C" int trs_cin_glue_bwd_reduce(const void* y, const void* g_hidden, const void* g_pooled, const float* scale,
                                       const float* shift, const float* mean, const float* invstd, int64_t B, int32_t E,
                                       int32_t C, int32_t D, int32_t Hs, int32_t dtype, float* partial,
                                       trs_stream_t stream) {
  TRS_GLUE_COMMON("cin_glue_bwd_reduce");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(y && scale && shift && mean && invstd && partial, TRS_EINVAL, "cin_glue_bwd_reduce: NULL pointer");
  TRS_REQUIRE(aligned16(y) && aligned16(g_hidden) && aligned16(g_pooled), TRS_EALIGN, "cin_glue_bwd_reduce: alignment");
  hipLaunchKernelGGL(glue_bwd_reduce_kernel, dim3(glue_grid(B)), dim3(GLUE_THREADS), (size_t)GLUE_THREADS * 8 * 4,
                     (hipStream_t)stream, (const bf16_t*)y, (const bf16_t*)g_hidden, (const bf16_t*)g_pooled, scale,
                     shift, mean, invstd, B, E, C, D, Hs, partial);
  return check_launch("cin_glue_bwd_reduce");
}

extern "C" int trs_cin_glue_bwd_apply(const void* y, const void* g_hidden, const void* g_pooled, const float* scale,
                                      const float* shift, const float* mean, const float* invstd, const float* c1,
                                      const float* c2, int64_t B, int32_t E, int32_t C, int32_t D, int32_t Hs,
                                      int32_t dtype, void* gy, trs_stream_t stream) {
  TRS_GLUE_COMMON("cin_glue_bwd_apply");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(y && scale && shift && mean && invstd && c1 && c2 && gy, TRS_EINVAL, "cin_glue_bwd_apply: NULL pointer");
  TRS_REQUIRE(aligned16(y) && aligned16(g_hidden) && aligned16(g_pooled) && aligned16(gy), TRS_EALIGN,
              "cin_glue_bwd_apply: alignment");
  hipLaunchKernelGGL(glue_bwd_apply_kernel, dim3(glue_grid(B)), dim3(GLUE_THREADS), 0, (hipStream_t)stream,
                     (const bf16_t*)y, (const bf16_t*)g_hidden, (const bf16_t*)g_pooled, scale, shift, mean, invstd, c1, c2,
                     B, E, C, D, Hs, (bf16_t*)gy);
  return check_launch("cin_glue_bwd_apply");
}

/* channels-first by-products: see glue_apply_fwd_cf_kernel */
extern "C" int trs_cin_glue_cf_supported(int32_t E, int32_t C) { return glue_cf_ok(C, E) ? 1 : 0; }

extern "C" int trs_cin_glue_fwd_cf(const void* y, const float* scale, const float* shift, int64_t B, int32_t E, int32_t C,
                                   int32_t D, int32_t Hs, int32_t dtype, void* hidden, void* hidden_cf, void* pooled,
                                   trs_stream_t stream) {
  TRS_GLUE_COMMON("cin_glue_fwd_cf");
  TRS_REQUIRE(glue_cf_ok(C, E), TRS_ESHAPE, "cin_glue_fwd_cf: needs E == 8 * (256 / (C / 8)) (E = %d, C = %d)", E, C);
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(y && scale && shift && ((hidden && hidden_cf) || Hs == C) && (pooled || D == 0), TRS_EINVAL,
              "cin_glue_fwd_cf: NULL pointer");
  TRS_REQUIRE(aligned16(y) && aligned16(hidden) && aligned16(hidden_cf) && aligned16(pooled), TRS_EALIGN,
              "cin_glue_fwd_cf: 16-byte alignment");
  hipLaunchKernelGGL(glue_apply_fwd_cf_kernel, dim3(glue_grid(B)), dim3(GLUE_THREADS),
                     (size_t)GLUE_THREADS * 8 * 4 + (size_t)(C - Hs) * (E + 8) * 2, (hipStream_t)stream, (const bf16_t*)y, scale, shift, B, E, C, D, Hs, (bf16_t*)hidden,
                     (bf16_t*)hidden_cf, (bf16_t*)pooled);
  return check_launch("cin_glue_fwd_cf");
}

extern "C" int trs_cin_glue_bwd_apply_cf(const void* y, const void* g_hidden, const void* g_pooled, const float* scale,
                                         const float* shift, const float* mean, const float* invstd, const float* c1,
                                         const float* c2, int64_t B, int32_t E, int32_t C, int32_t D, int32_t Hs,
                                         int32_t dtype, void* gy, void* gy_cf, float* colsum_partial,
                                         trs_stream_t stream) {
  TRS_GLUE_COMMON("cin_glue_bwd_apply_cf");
  TRS_REQUIRE(glue_cf_ok(C, E), TRS_ESHAPE, "cin_glue_bwd_apply_cf: needs E == 8 * (256 / (C / 8)) (E = %d, C = %d)", E,
              C);
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(y && scale && shift && mean && invstd && c1 && c2 && gy && gy_cf, TRS_EINVAL,
              "cin_glue_bwd_apply_cf: NULL pointer");
  TRS_REQUIRE(aligned16(y) && aligned16(g_hidden) && aligned16(g_pooled) && aligned16(gy) && aligned16(gy_cf), TRS_EALIGN,
              "cin_glue_bwd_apply_cf: alignment");
  // the LDS tile [C][E+8] bf16 is also the scratch of the final column-sum reduction ([rows per pass][C] fp32)
  const size_t lds = std::max((size_t)C * (E + 8) * 2, (size_t)(GLUE_THREADS / (C / 8)) * C * 4);
  hipLaunchKernelGGL(glue_bwd_apply_cf_kernel, dim3(glue_grid(B)), dim3(GLUE_THREADS), lds, (hipStream_t)stream,
                     (const bf16_t*)y, (const bf16_t*)g_hidden, (const bf16_t*)g_pooled, scale, shift, mean, invstd, c1, c2,
                     B, E, C, D, Hs, (bf16_t*)gy, (bf16_t*)gy_cf, colsum_partial);
  return check_launch("cin_glue_bwd_apply_cf");
}

// ------------------------------------------------------------------------------------------------
// Batched 2-D transposition with zero padding, 2-byte elements: out[b][c][r] = in[b][r][c] for r < R, c < Cc and
// out[b][c][r] = 0 for R <= r < ld_out.  in (B, R, ld_in), out (B, Cc, ld_out); ld_in, ld_out multiples of 8.
// The CIN layer keeps its activations channels-last (compress_interaction_network.py:105 aligns to ('B','E','N')):
// x0 (B,N,E) -> x0T (B,E,ld0) with N padded to the 32-wide k-step (R = N, Cc = E, ld_out = ld0), and the gradient back
// (R = E, Cc = N, ld_in = ld0, ld_out = E).  One pass each instead of new_zeros + a strided slice copy forward and
// CopySlices' clone + slice clone backward (0.43 + 0.37 ms per step at 65 536 x 39 x 64).
namespace trs {
constexpr int TP_PITCH = 66;      // elements per LDS row: (8 rb + j) * 33 + c / 2 spreads a wave's 2-byte reads over the banks
__global__ __launch_bounds__(256) void transpose_pad_kernel(const unsigned short* __restrict__ in, int R, int Cc,
                                                            int ld_in, unsigned short* __restrict__ out, int ld_out,
                                                            int64_t B) {
  __shared__ unsigned short tile[64 * TP_PITCH];
  const int vin = ld_in / 8, vout = ld_out / 8;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint4* src = reinterpret_cast<const uint4*>(in + (size_t)b * R * ld_in);
    for (int v = threadIdx.x; v < R * vin; v += 256) {
      const int r = v / vin, c8 = (v - r * vin) * 8;
      const uint4 x = load_stream(&src[v]);
      const unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // two elements per word; columns >= Cc of a padded input row are never read back
        *reinterpret_cast<unsigned*>(&tile[r * TP_PITCH + c8 + 2 * k]) = w[k];
      }
    }
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(out + (size_t)b * Cc * ld_out);
    for (int v = threadIdx.x; v < Cc * vout; v += 256) {
      const int c = v / vout, r0 = (v - c * vout) * 8;
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (r0 + j) < R ? tile[(r0 + j) * TP_PITCH + c] : (unsigned short)0;
      uint4 o;
      o.x = e[0] | ((unsigned)e[1] << 16);
      o.y = e[2] | ((unsigned)e[3] << 16);
      o.z = e[4] | ((unsigned)e[5] << 16);
      o.w = e[6] | ((unsigned)e[7] << 16);
      dst[v] = o;
    }
    __syncthreads();
  }
}
}  // namespace trs

extern "C" int trs_transpose_pad(const void* in, int64_t B, int32_t R, int32_t Cc, int32_t ld_in, void* out,
                                 int32_t ld_out, int32_t dtype, trs_stream_t stream) {
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "transpose_pad: bf16 only (dtype %d)", dtype);
  TRS_REQUIRE(B >= 0 && R > 0 && Cc > 0 && R <= 64 && Cc <= 64 && ld_in >= Cc && ld_out >= R && ld_in <= 64 &&
                  ld_in % 8 == 0 && ld_out % 8 == 0,
              TRS_ESHAPE, "transpose_pad: R = %d, Cc = %d, ld_in = %d, ld_out = %d (at most 64 x 64, rows of whole 16-byte vectors)",
              R, Cc, ld_in, ld_out);
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(in && out, TRS_EINVAL, "transpose_pad: NULL pointer");
  TRS_REQUIRE(aligned16(in) && aligned16(out), TRS_EALIGN, "transpose_pad: 16-byte alignment");
  hipLaunchKernelGGL(transpose_pad_kernel, dim3((int)std::min<int64_t>(B, 256 * 16)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)in, R, Cc, ld_in, (unsigned short*)out, ld_out, B);
  return check_launch("transpose_pad");
}
