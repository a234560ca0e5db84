// K1+K2(+K8): fused embedding lookup + FM second order (+ first-order sum), FM layer fwd/bwd.
//
// Layout: one table row of E values is split into L = E*sizeof(T)/16 sixteen-byte vectors; a GROUP
// of L adjacent lanes owns one sample and walks its N fields, so every lane keeps the running
// sum / sum-of-squares of its own VE columns in registers -- no cross-lane traffic, no LDS.  A wave
// therefore covers 64/L samples per instruction and each row read is one full 16*L-byte segment
// (128 B = one cache line for bf16 E=64).  Rows are fetched CH = 4 at a time (4 independent 16-byte loads per lane
// in flight; HBM latency ~1 us on a random row): 60 VGPRs, 8 waves per SIMD, so the 2048 workgroups of the
// B = 65 536 launch are all resident at once.  CH = 8 needed 83 VGPRs (5 waves per SIMD: the grid then ran as one
// full round plus a 60 % one) and was 8 % slower inside the training step (142 -> 130 us) although each wave had
// twice the loads in flight.  Round 4, for the HBM-resident case (32 M rows = 4 GiB, where every row is a DRAM page miss):
// the row ids of the NEXT chunk requested beside the current chunk's rows (a chunk is otherwise two dependent round
// trips) with 4 or 8 rows in flight -- 137.4 / 136.9 us against 137.1 with the block, 77.1 / 78.3 against 75.4 without:
// no gain, removed again.  The walk is not short of loads in flight (128 KB per CU); 4.7-5.0 TB/s is what random 128-byte
// rows get out of the memory system here (the device-to-device copy ceiling is 6.3 TB/s).
// HBM-bound: algorithmic bytes per sample = N*(idx 8 + E*s) read, E*s (+N*E*s with the block) written.
#include <algorithm>
#include <cstdlib>

#include "trs_common.hpp"

namespace trs {

// STREAM: the table is far larger than the caches (EMBED_STREAM_BYTES): its rows are fetched with streaming loads -- a row
// that will not be looked up again before it is evicted anyway should not displace the index / output traffic in L2 (4 GiB
// table, FM only: 72.1-74.8 -> 70.6-71.5 us; on a cache-resident table the same loads cost 115 -> 128 us: rows ARE reused)
template <typename T, typename IdxT, int LOG2L, bool GATHER, bool STREAM = false>
__global__ __launch_bounds__(256) void embed_fm_group_kernel(
    const uint4* __restrict__ src,  // GATHER: table (V x E); else x (B x N x E)
    const IdxT* __restrict__ idx, const int64_t* __restrict__ offsets, int64_t B, int N, int64_t V,
    uint4* __restrict__ emb, uint4* __restrict__ fm, float* __restrict__ fm_sum,
    const T* __restrict__ first_table, T* __restrict__ first, int32_t* __restrict__ err_flag,
    T* __restrict__ first_vals = nullptr /* (B,N): the companion table's value of every lookup */) {
  constexpr int L = 1 << LOG2L;
  constexpr int VE = Vec16<T>::VE;
  constexpr int CH = 4;
  const int lane_v = threadIdx.x & (L - 1);
  const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> LOG2L;
  for (int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> LOG2L; b < B; b += groups) {
    float s[VE], q[VE];
#pragma unroll
    for (int k = 0; k < VE; ++k) { s[k] = 0.f; q[k] = 0.f; }
    float f1 = 0.f;
    for (int n0 = 0; n0 < N; n0 += CH) {
      int64_t r[CH];
      uint4 v[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int n = n0 + c;
        r[c] = -1;
        if (n < N) {
          if (GATHER) {
            r[c] = load_row_id(idx, offsets, b * N + n, n);
            if (err_flag != nullptr && (r[c] < 0 || r[c] >= V)) { *err_flag = 1; r[c] = -1; }
          } else {
            r[c] = b * N + n;
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        v[c] = make_uint4(0, 0, 0, 0);
        if (r[c] >= 0) v[c] = STREAM ? load_stream(&src[r[c] * L + lane_v]) : src[r[c] * L + lane_v];
      }
      if (GATHER && first_table != nullptr) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if ((((n0 + c) & (L - 1)) == lane_v) && n0 + c < N) {
            const T fv = r[c] >= 0 ? first_table[r[c]] : T{};
            f1 += to_f32(fv);
            if (first_vals != nullptr) first_vals[b * N + n0 + c] = fv;
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int n = n0 + c;
        if (n < N) {
          float x[VE];
          Vec16<T>::unpack(v[c], x);
#pragma unroll
          for (int k = 0; k < VE; ++k) { s[k] += x[k]; q[k] = fmaf(x[k], x[k], q[k]); }
          if (GATHER && emb != nullptr) {
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            const u32x4 w = {v[c].x, v[c].y, v[c].z, v[c].w};
            // streaming store: the block is consumed by a later kernel, keep L2 for the table rows
            __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(&emb[(b * N + n) * L + lane_v]));
          }
        }
      }
    }
    if (fm != nullptr) {
      float o[VE];
#pragma unroll
      for (int k = 0; k < VE; ++k) o[k] = 0.5f * (s[k] * s[k] - q[k]);
      fm[b * L + lane_v] = Vec16<T>::pack(o);
    }
    if (fm_sum != nullptr) {
      float4* dst = reinterpret_cast<float4*>(fm_sum + (b * L + lane_v) * VE);
#pragma unroll
      for (int k = 0; k < VE; k += 4) dst[k / 4] = make_float4(s[k], s[k + 1], s[k + 2], s[k + 3]);
    }
    if (GATHER && first_table != nullptr) {
#pragma unroll
      for (int m = L >> 1; m >= 1; m >>= 1) f1 += __shfl_xor(f1, m, 64);
      if (lane_v == 0 && first != nullptr) first[b] = from_f32<T>(f1);
    }
  }
}

// generic path: any E; one thread per (b, e)
template <typename T, typename IdxT, bool GATHER>
__global__ __launch_bounds__(256) void embed_fm_elem_kernel(
    const T* __restrict__ src, const IdxT* __restrict__ idx, const int64_t* __restrict__ offsets, int64_t B,
    int N, int E, int64_t V, T* __restrict__ emb, T* __restrict__ fm, float* __restrict__ fm_sum,
    const T* __restrict__ first_table, T* __restrict__ first, int32_t* __restrict__ err_flag,
    T* __restrict__ first_vals = nullptr) {
  const int64_t total = B * E;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool f32 = total < ((int64_t)1 << 32);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = udiv_fast(t, E, f32);
    const int e = (int)(t - b * E);
    float s = 0.f, q = 0.f, f1 = 0.f;
    for (int n = 0; n < N; ++n) {
      int64_t r = b * N + n;
      bool ok = true;
      if (GATHER) {
        r = load_row_id(idx, offsets, b * N + n, n);
        if (err_flag != nullptr && (r < 0 || r >= V)) { *err_flag = 1; ok = false; }
      }
      T raw = T{};
      if (ok) raw = src[r * E + e];
      const float x = to_f32(raw);
      s += x;
      q = fmaf(x, x, q);
      if (GATHER && emb != nullptr) emb[(b * N + n) * E + e] = raw;
      if (GATHER && first_table != nullptr && e == 0) {
        const T fv = ok ? first_table[r] : T{};
        f1 += to_f32(fv);
        if (first_vals != nullptr) first_vals[b * N + n] = fv;
      }
    }
    if (fm != nullptr) fm[t] = from_f32<T>(0.5f * (s * s - q));
    if (fm_sum != nullptr) fm_sum[t] = s;
    if (GATHER && first_table != nullptr && e == 0 && first != nullptr) first[b] = from_f32<T>(f1);
  }
}

constexpr size_t EMBED_STREAM_BYTES = (size_t)512 << 20;      // twice the Infinity Cache

static int log2_lanes(int row_bytes) {
  if (row_bytes % 16 != 0) return -1;
  const int L = row_bytes / 16;
  if (!is_pow2(L) || L > 64) return -1;
  int l = 0;
  while ((1 << l) < L) ++l;
  return l;
}

template <typename T, typename IdxT, bool GATHER>
static int embed_fm_launch(const void* src, const IdxT* idx, const int64_t* offsets, int64_t B, int N, int E,
                           int64_t V, void* emb, void* fm, float* fm_sum, const void* first_table, void* first,
                           int32_t* err_flag, hipStream_t s, void* first_vals = nullptr) {
  const int lg = log2_lanes(E * (int)sizeof(T));
  const bool al = aligned16(src) && aligned16(emb) && aligned16(fm) && aligned16(fm_sum);
  if (lg >= 0 && al) {
    const int L = 1 << lg;
    const int grid = stream_grid(B * L, 256, 256 * 16);
    const bool stream = GATHER && (size_t)V * E * sizeof(T) > EMBED_STREAM_BYTES;
#define TRS_EF2(LG, ST)                                                                                   \
  hipLaunchKernelGGL((embed_fm_group_kernel<T, IdxT, LG, GATHER, ST>), dim3(grid), dim3(256), 0, s,        \
                     (const uint4*)src, idx, offsets, B, N, V, (uint4*)emb, (uint4*)fm, fm_sum,            \
                     (const T*)first_table, (T*)first, err_flag, (T*)first_vals)
#define TRS_EF(LG)                 \
  if (GATHER && stream) {          \
    TRS_EF2(LG, GATHER);           \
  } else {                         \
    TRS_EF2(LG, false);            \
  }
    switch (lg) {
      case 0: TRS_EF(0); break;
      case 1: TRS_EF(1); break;
      case 2: TRS_EF(2); break;
      case 3: TRS_EF(3); break;
      case 4: TRS_EF(4); break;
      case 5: TRS_EF(5); break;
      default: TRS_EF(6); break;
    }
#undef TRS_EF
#undef TRS_EF2
  } else {
    const int grid = stream_grid(B * E, 256, 256 * 16);
    hipLaunchKernelGGL((embed_fm_elem_kernel<T, IdxT, GATHER>), dim3(grid), dim3(256), 0, s, (const T*)src, idx,
                       offsets, B, N, E, V, (T*)emb, (T*)fm, fm_sum, (const T*)first_table, (T*)first, err_flag,
                       (T*)first_vals);
  }
  return check_launch(GATHER ? "embed_fm" : "fm_fwd");
}

// ---- FM backward on a materialised block: dx = g * (S - x)
template <typename T>
__global__ __launch_bounds__(256) void fm_bwd_vec_kernel(const uint4* __restrict__ x, const uint4* __restrict__ g,
                                                         const float* __restrict__ fm_sum, uint4* __restrict__ dx,
                                                         int64_t total_vecs, int N, int vpr) {
  constexpr int VE = Vec16<T>::VE;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int per_b = N * vpr;
  const bool f32 = total_vecs < ((int64_t)1 << 32);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total_vecs; t += stride) {
    const int64_t b = udiv_fast(t, per_b, f32);
    const int64_t row = udiv_fast(t, vpr, f32);
    const int lv = (int)(t - row * vpr);
    float xv[VE], gv[VE], o[VE];
    Vec16<T>::unpack(x[t], xv);
    Vec16<T>::unpack(g[b * vpr + lv], gv);
    const float* sp = fm_sum + (b * vpr + lv) * VE;
#pragma unroll
    for (int k = 0; k < VE; ++k) o[k] = gv[k] * (sp[k] - xv[k]);
    store_stream(&dx[t], Vec16<T>::pack(o));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void fm_bwd_elem_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                          const float* __restrict__ fm_sum, T* __restrict__ dx,
                                                          int64_t total, int N, int E) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t per_b = (int64_t)N * E;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / per_b;
    const int e = (int)(t % E);
    dx[t] = from_f32<T>(to_f32(g[b * E + e]) * (fm_sum[b * E + e] - to_f32(x[t])));
  }
}

template <typename T>
static int fm_bwd_launch(const void* x, const void* g, const float* fm_sum, int64_t B, int N, int E, void* dx,
                         hipStream_t s) {
  const int row_bytes = E * (int)sizeof(T);
  if (row_bytes % 16 == 0 && aligned16(x) && aligned16(g) && aligned16(dx)) {
    const int vpr = row_bytes / 16;
    const int64_t total = B * N * vpr;
    hipLaunchKernelGGL((fm_bwd_vec_kernel<T>), dim3(stream_grid(total, 256, 256 * 32)), dim3(256), 0, s,
                       (const uint4*)x, (const uint4*)g, fm_sum, (uint4*)dx, total, N, vpr);
  } else {
    const int64_t total = B * N * E;
    hipLaunchKernelGGL((fm_bwd_elem_kernel<T>), dim3(stream_grid(total, 256, 256 * 32)), dim3(256), 0, s,
                       (const T*)x, (const T*)g, fm_sum, (T*)dx, total, N, E);
  }
  return check_launch("fm_bwd");
}

// rows of d(block) in exchange order for the sharded lookup:
//   out[k,:] = g_block[pos[k],:] + g_fm[b,:] * (fm_sum[b,:] - x[pos[k],:]),   b = pos[k] / N
// (either term optional) -- one pass instead of fm_bwd + add + permute.
template <typename T>
__global__ __launch_bounds__(256) void permute_grad_vec_kernel(const uint4* __restrict__ g_block,
                                                               const uint4* __restrict__ g_fm,
                                                               const float* __restrict__ fm_sum,
                                                               const uint4* __restrict__ x, const int32_t* __restrict__ pos,
                                                               uint4* __restrict__ out, int64_t K, int N, int vpr) {
  constexpr int VE = Vec16<T>::VE;
  const int64_t total = K * vpr, stride = (int64_t)gridDim.x * blockDim.x;
  const bool f32 = total < ((int64_t)1 << 32);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t k = udiv_fast(t, vpr, f32);
    const int lv = (int)(t - k * vpr);
    const int64_t p = pos[k];
    float o[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) o[i] = 0.f;
    if (p < 0) {      // a padding slot of a fixed-capacity exchange: a zero gradient row
      store_stream(&out[t], Vec16<T>::pack(o));
      continue;
    }
    if (g_block != nullptr) Vec16<T>::unpack(g_block[p * vpr + lv], o);
    if (g_fm != nullptr) {
      const int64_t b = (int64_t)((unsigned)p / (unsigned)N);       // pos is int32
      float gf[VE], xv[VE];
      Vec16<T>::unpack(g_fm[b * vpr + lv], gf);
      Vec16<T>::unpack(x[p * vpr + lv], xv);
      const float* sp = fm_sum + (b * vpr + lv) * VE;
#pragma unroll
      for (int i = 0; i < VE; ++i) o[i] = fmaf(gf[i], sp[i] - xv[i], o[i]);
    }
    store_stream(&out[t], Vec16<T>::pack(o));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void permute_grad_elem_kernel(const T* __restrict__ g_block, const T* __restrict__ g_fm,
                                                                const float* __restrict__ fm_sum, const T* __restrict__ x,
                                                                const int32_t* __restrict__ pos, T* __restrict__ out,
                                                                int64_t K, int N, int E) {
  const int64_t total = K * E, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t k = t / E;
    const int e = (int)(t - k * E);
    const int64_t p = pos[k];
    if (p < 0) {      // padding slot
      out[t] = from_f32<T>(0.f);
      continue;
    }
    float o = g_block ? to_f32(g_block[p * E + e]) : 0.f;
    if (g_fm != nullptr) {
      const int64_t b = p / N;
      o = fmaf(to_f32(g_fm[b * E + e]), fm_sum[b * E + e] - to_f32(x[p * E + e]), o);
    }
    out[t] = from_f32<T>(o);
  }
}

}  // namespace trs

using namespace trs;

static int embed_fm_entry(const void* table, int64_t V, int32_t E, int32_t dtype, const void* idx, int32_t idx_dtype,
                          const int64_t* offsets, int64_t B, int32_t N, void* emb, void* fm, float* fm_sum,
                          const void* first_table, void* first, void* first_vals, int32_t* err_flag,
                          trs_stream_t stream) {
  if (B == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  TRS_REQUIRE(table && idx, TRS_EINVAL, "embed_fm: NULL pointer");
  TRS_REQUIRE(V > 0 && E > 0 && B >= 0 && N > 0, TRS_EINVAL, "embed_fm: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "embed_fm: dtype %d", dtype);
  TRS_REQUIRE(idx_dtype == TRS_I64 || idx_dtype == TRS_I32, TRS_EDTYPE, "embed_fm: idx dtype %d", idx_dtype);
  TRS_REQUIRE((first_table == nullptr) == (first == nullptr && first_vals == nullptr), TRS_EINVAL,
              "embed_fm: first_table and its output must be given together");
  hipStream_t s = (hipStream_t)stream;
#define TRS_CALL(T, I)                                                                                    \
  return embed_fm_launch<T, I, true>(table, (const I*)idx, offsets, B, N, E, V, emb, fm, fm_sum, first_table, \
                                     first, err_flag, s, first_vals)
  if (dtype == TRS_F32) {
    if (idx_dtype == TRS_I64) TRS_CALL(float, int64_t);
    TRS_CALL(float, int32_t);
  }
  if (idx_dtype == TRS_I64) TRS_CALL(bf16_t, int64_t);
  TRS_CALL(bf16_t, int32_t);
#undef TRS_CALL
}

extern "C" int trs_embed_fm(const void* table, int64_t V, int32_t E, int32_t dtype, const void* idx,
                            int32_t idx_dtype, const int64_t* offsets, int64_t B, int32_t N, void* emb, void* fm,
                            float* fm_sum, const void* first_table, void* first, int32_t* err_flag,
                            trs_stream_t stream) {
  return embed_fm_entry(table, V, E, dtype, idx, idx_dtype, offsets, B, N, emb, fm, fm_sum, first_table, first, nullptr,
                        err_flag, stream);
}

/* see include/trs_abi.h: the same pass, with the companion table's value of every lookup (B,N) instead of their sum */
extern "C" int trs_embed_fm_fields(const void* table, int64_t V, int32_t E, int32_t dtype, const void* idx,
                                   int32_t idx_dtype, const int64_t* offsets, int64_t B, int32_t N, void* emb, void* fm,
                                   float* fm_sum, const void* first_table, void* first_vals, int32_t* err_flag,
                                   trs_stream_t stream) {
  TRS_REQUIRE(B == 0 || (first_table && first_vals), TRS_EINVAL, "embed_fm_fields: NULL first-order pointer");
  return embed_fm_entry(table, V, E, dtype, idx, idx_dtype, offsets, B, N, emb, fm, fm_sum, first_table, nullptr,
                        first_vals, err_flag, stream);
}

extern "C" int trs_fm_fwd(const void* x, int64_t B, int32_t N, int32_t E, int32_t dtype, void* fm, float* fm_sum,
                          trs_stream_t stream) {
  if (B == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  TRS_REQUIRE(x && (fm || fm_sum), TRS_EINVAL, "fm_fwd: NULL pointer");
  TRS_REQUIRE(E > 0 && B >= 0 && N > 0, TRS_EINVAL, "fm_fwd: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "fm_fwd: dtype %d", dtype);
  if (B == 0) return TRS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TRS_F32)
    return embed_fm_launch<float, int64_t, false>(x, nullptr, nullptr, B, N, E, 0, nullptr, fm, fm_sum, nullptr,
                                                  nullptr, nullptr, s);
  return embed_fm_launch<bf16_t, int64_t, false>(x, nullptr, nullptr, B, N, E, 0, nullptr, fm, fm_sum, nullptr,
                                                 nullptr, nullptr, s);
}

extern "C" int trs_fm_bwd(const void* x, const void* g, const float* fm_sum, int64_t B, int32_t N, int32_t E,
                          int32_t dtype, void* dx, trs_stream_t stream) {
  if (B == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  TRS_REQUIRE(x && g && fm_sum && dx, TRS_EINVAL, "fm_bwd: NULL pointer");
  TRS_REQUIRE(E > 0 && B >= 0 && N > 0, TRS_EINVAL, "fm_bwd: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "fm_bwd: dtype %d", dtype);
  if (B == 0) return TRS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TRS_F32) return fm_bwd_launch<float>(x, g, fm_sum, B, N, E, dx, s);
  return fm_bwd_launch<bf16_t>(x, g, fm_sum, B, N, E, dx, s);
}

extern "C" int trs_permute_grad(const void* g_block, const void* g_fm, const float* fm_sum, const void* x,
                                const int32_t* pos, int64_t K, int32_t N, int32_t E, int32_t dtype, void* out,
                                trs_stream_t stream) {
  if (K == 0) return TRS_OK;
  TRS_REQUIRE(pos && out && (g_block || g_fm), TRS_EINVAL, "permute_grad: NULL pointer");
  TRS_REQUIRE(g_fm == nullptr || (fm_sum && x), TRS_EINVAL, "permute_grad: g_fm needs fm_sum and x");
  TRS_REQUIRE(K > 0 && N > 0 && E > 0, TRS_EINVAL, "permute_grad: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "permute_grad: dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  const int rb = E * dtype_size(dtype);
  const bool vec = rb % 16 == 0 && aligned16(g_block) && aligned16(g_fm) && aligned16(x) && aligned16(out) &&
                   aligned16(fm_sum);
  if (vec) {
    const int vpr = rb / 16;
    const int grid = stream_grid(K * vpr, 256, 256 * 32);
    if (dtype == TRS_F32)
      hipLaunchKernelGGL((permute_grad_vec_kernel<float>), dim3(grid), dim3(256), 0, s, (const uint4*)g_block,
                         (const uint4*)g_fm, fm_sum, (const uint4*)x, pos, (uint4*)out, K, N, vpr);
    else
      hipLaunchKernelGGL((permute_grad_vec_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const uint4*)g_block,
                         (const uint4*)g_fm, fm_sum, (const uint4*)x, pos, (uint4*)out, K, N, vpr);
  } else {
    const int grid = stream_grid(K * E, 256, 256 * 32);
    if (dtype == TRS_F32)
      hipLaunchKernelGGL((permute_grad_elem_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)g_block,
                         (const float*)g_fm, fm_sum, (const float*)x, pos, (float*)out, K, N, E);
    else
      hipLaunchKernelGGL((permute_grad_elem_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)g_block,
                         (const bf16_t*)g_fm, fm_sum, (const bf16_t*)x, pos, (bf16_t*)out, K, N, E);
  }
  return check_launch("permute_grad");
}
