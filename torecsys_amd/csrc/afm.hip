// AttentionalFactorizationMachineLayer, layers/ctr/attentional_factorization_machine.py:86-125
//   prod[b,p,:] = x[b,i_p,:] * x[b,j_p,:]                                   (i<j pairs, NC2 of them)
//   score[b,p]  = softmax_p( w2 . relu(W1 prod[b,p,:] + b1) + b2 )           (attention MLP E -> A -> 1)
//   out[b,:]    = sum_p score[b,p] * prod[b,p,:]
// (both nn.Dropout modules stay in Python).  The reference materialises prod (B,NC2,E) -- 6.2 GB at B = 65 536,
// N = 39, E = 64 bf16 -- and runs the attention MLP as two nn.Linear over it; here a workgroup keeps the (N x E)
// block of one sample, the attention weights and the NC2 logits in LDS and the products never leave registers.
// Generic path (any dtype / shape with E, A <= 128): lanes along the attention units for the hidden layer
// (wavefront reduction for the logit), lanes along e for the weighted sum and the backward.
#include <stdlib.h>

#include <algorithm>

#include <type_traits>

#include "trs_common.hpp"
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace trs {

__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide reductions over 256 threads through 4 LDS slots (+ broadcast)
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

constexpr int AFM_MAX = 128;     // E and A limit of the generic path (two 64-lane slabs)

struct AfmLds {
  float* xs;    // [N][E]
  float* w1t;   // [E][A]   (lanes along a)
  float* w1;    // [A][E]   (lanes along e; backward only)
  float* b1;    // [A]
  float* w2;    // [A]
  float* lg;    // [P]  logits -> scores
  float* aux;   // [P]  backward: d(score) -> d(logit)
  float* msk;   // [P]  score-dropout multiplier of every pair (0 or keep_scale; 1 without dropout)
  float* vec;   // [E]  forward: output accumulator; backward: g_out
  float* dh;    // [4][A] per-wave d(hidden)
  float* red;   // [4]
};

__device__ __forceinline__ AfmLds afm_carve(float* smem, int N, int E, int A, int P, bool bwd) {
  AfmLds l;
  float* p = smem;
  l.xs = p; p += N * E;
  l.w1t = p; p += E * A;
  l.w1 = p; p += bwd ? A * E : 0;
  l.b1 = p; p += A;
  l.w2 = p; p += A;
  l.lg = p; p += P;
  l.aux = p; p += bwd ? P : 0;
  l.msk = p; p += P;
  l.vec = p; p += E;
  l.dh = p; p += bwd ? 4 * A : 0;
  l.red = p;
  return l;
}
__host__ __device__ inline size_t afm_lds_floats(int N, int E, int A, int P, bool bwd) {
  return (size_t)N * E + (size_t)E * A * (bwd ? 2 : 1) + 2 * A + (size_t)P * (bwd ? 3 : 2) + E + (bwd ? 4 * A : 0) + 8;
}

template <typename T>
__device__ __forceinline__ void afm_stage_weights(const AfmLds& l, const T* W1, const T* b1, const T* w2, int E, int A,
                                                  bool bwd) {
  for (int k = threadIdx.x; k < A * E; k += blockDim.x) {
    const int a = k / E, e = k - a * E;
    const float w = to_f32(W1[k]);
    l.w1t[e * A + a] = w;
    if (bwd) l.w1[k] = w;
  }
  for (int a = threadIdx.x; a < A; a += blockDim.x) {
    l.b1[a] = to_f32(b1[a]);
    l.w2[a] = to_f32(w2[a]);
  }
}

// hidden pre-activations of one pair for this lane's attention units (slabs of 64): acc[s] for a = 64 s + lane
__device__ __forceinline__ void afm_hidden(const AfmLds& l, int i, int j, int E, int A, int lane, float* acc /*[2]*/) {
  const float* xi = l.xs + i * E;
  const float* xj = l.xs + j * E;
  const int a0 = lane, a1 = 64 + lane;
  acc[0] = a0 < A ? l.b1[a0] : 0.f;
  acc[1] = a1 < A ? l.b1[a1] : 0.f;
  for (int e = 0; e < E; ++e) {
    const float pr = xi[e] * xj[e];
    if (a0 < A) acc[0] = fmaf(pr, l.w1t[e * A + a0], acc[0]);
    if (a1 < A) acc[1] = fmaf(pr, l.w1t[e * A + a1], acc[1]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void afm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ W1,
                                                      const T* __restrict__ b1, const T* __restrict__ w2,
                                                      const T* __restrict__ b2, int64_t B, int N, int E, int A,
                                                      T* __restrict__ out, T* __restrict__ attn,
                                                      const uint8_t* __restrict__ keep, float keep_scale,
                                                      T* __restrict__ attn_drop) {
  extern __shared__ float smem[];
  const int P = N * (N - 1) / 2;
  const AfmLds l = afm_carve(smem, N, E, A, P, false);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  afm_stage_weights(l, W1, b1, w2, E, A, false);
  const float bias2 = to_f32(b2[0]);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    for (int k = threadIdx.x; k < N * E; k += 256) l.xs[k] = to_f32(x[b * N * E + k]);
    for (int k = threadIdx.x; k < E; k += 256) l.vec[k] = 0.f;
    __syncthreads();
    // logits: one wave per pair, lanes along the attention units
    for (int p = wave; p < P; p += 4) {
      int i, j;
      pair_ij(p, N, &i, &j);
      float acc[2];
      afm_hidden(l, i, j, E, A, lane, acc);
      float part = 0.f;
      if (lane < A) part = fmaf(fmaxf(acc[0], 0.f), l.w2[lane], part);
      if (64 + lane < A) part = fmaf(fmaxf(acc[1], 0.f), l.w2[64 + lane], part);
      part = wave_sum(part);
      if (lane == 0) l.lg[p] = part + bias2;
    }
    __syncthreads();
    // softmax over the pairs of the sample
    float m = -INFINITY;
    for (int p = threadIdx.x; p < P; p += 256) m = fmaxf(m, l.lg[p]);
    m = block_max(m, l.red);
    float s = 0.f;
    for (int p = threadIdx.x; p < P; p += 256) {
      const float e = __expf(l.lg[p] - m);
      l.lg[p] = e;
      s += e;
    }
    s = block_sum(s, l.red);
    const float inv = 1.f / s;
    for (int p = threadIdx.x; p < P; p += 256) {
      float sc = l.lg[p] * inv;
      attn[b * P + p] = from_f32<T>(sc);
      if (keep != nullptr) {      // dropout on the scores (attentional_factorization_machine.py:82): the sum uses the dropped ones
        sc = keep[b * P + p] ? sc * keep_scale : 0.f;
        attn_drop[b * P + p] = from_f32<T>(sc);
      }
      l.lg[p] = sc;
    }
    __syncthreads();
    // out[e] = sum_p score[p] x_i[e] x_j[e]: each wave takes every 4th pair, lanes along e
    float o0 = 0.f, o1 = 0.f;
    for (int p = wave; p < P; p += 4) {
      int i, j;
      pair_ij(p, N, &i, &j);
      const float sc = l.lg[p];
      if (lane < E) o0 = fmaf(sc, l.xs[i * E + lane] * l.xs[j * E + lane], o0);
      if (64 + lane < E) o1 = fmaf(sc, l.xs[i * E + 64 + lane] * l.xs[j * E + 64 + lane], o1);
    }
    if (lane < E) atomicAdd(&l.vec[lane], o0);
    if (64 + lane < E) atomicAdd(&l.vec[64 + lane], o1);
    __syncthreads();
    for (int k = threadIdx.x; k < E; k += 256) out[b * E + k] = from_f32<T>(l.vec[k]);
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 fast path, forward.  The attention hidden layer is a GEMM over (pairs x E) @ (E x A) per sample:
//   hidden^T[a][p] = W1[a][:] . prod[p][:]         (MFMA 16x16x32 bf16, fp32 accumulate)
// A operand: W1 rows (16 attention units per tile), fragments resident in registers for the whole kernel;
// B operand: prod^T for 16 pairs, built on the fly -- lane (pair n = l&15, e chunk q = l>>4) multiplies the two
// 16-byte runs x_i[8q..8q+7], x_j[8q..8q+7] it reads from the LDS copy of the sample (rows padded to E*2+16 bytes:
// conflict-free ds_read_b128) and packs them to bf16.  D: lane holds, for its pair, four attention units per tile;
// relu / w2 / the reduction over units happen in registers + two cross-lane adds; logits land in LDS for the softmax.
// The pair products never exist in memory.
typedef __attribute__((ext_vector_type(8))) __bf16 afm_bf16x8;
typedef __attribute__((ext_vector_type(4))) float afm_f32x4;

template <int AT /* A/16 */, int KS /* E/32 */>
__global__ __launch_bounds__(256) void afm_fwd_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W1,
                                                           const bf16_t* __restrict__ b1, const bf16_t* __restrict__ w2,
                                                           const bf16_t* __restrict__ b2, int64_t B, int N,
                                                           bf16_t* __restrict__ out, bf16_t* __restrict__ attn,
                                                           const uint8_t* __restrict__ keep, float keep_scale,
                                                           bf16_t* __restrict__ attn_drop) {
  constexpr int E = 32 * KS, A = 16 * AT, RS = E * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int P = N * (N - 1) / 2, PT = (P + 15) / 16;
  char* xs = smem_raw;                                           // [N][RS] bf16 rows
  float* lg = reinterpret_cast<float*>(smem_raw + ((N * RS + 15) & ~15));      // [PT*16]
  int* lut = reinterpret_cast<int*>(lg + PT * 16);               // [PT*16]  (i << 16) | j, padded with pair 0
  float* vec = reinterpret_cast<float*>(lut + PT * 16);          // [E]
  float* red = vec + E;                                          // [8]
  float* mk = red + 8;                                           // [PT*16] score-dropout multipliers (dropout only)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
  for (int p = threadIdx.x; p < PT * 16; p += 256) {
    int i = 0, j = 1;
    if (p < P) pair_ij(p, N, &i, &j);
    lut[p] = (i << 16) | j;
  }
  // resident A fragments (W1) and this lane's b1 / w2 values (attention unit a = 16 mt + 4 q + r)
  uint4 Wf[AT][KS];
  float b1v[AT][4], w2v[AT][4];
#pragma unroll
  for (int mt = 0; mt < AT; ++mt) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      Wf[mt][ks] = *reinterpret_cast<const uint4*>(W1 + (size_t)(16 * mt + n) * E + 32 * ks + 8 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      b1v[mt][r] = to_f32(b1[16 * mt + 4 * q + r]);
      w2v[mt][r] = to_f32(w2[16 * mt + 4 * q + r]);
    }
  }
  const float bias2 = to_f32(b2[0]);
  constexpr int VPR = E / 8;                                     // 16-byte vectors per x row
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    for (int v = threadIdx.x; v < N * VPR; v += 256) {
      const int row = v / VPR, col = v - row * VPR;
      *reinterpret_cast<uint4*>(xs + row * RS + col * 16) =
          *reinterpret_cast<const uint4*>(x + (b * N + row) * (int64_t)E + col * 8);
    }
    for (int k = threadIdx.x; k < E; k += 256) vec[k] = 0.f;
    if (keep != nullptr)
      for (int p = threadIdx.x; p < PT * 16; p += 256) mk[p] = (p < P && keep[b * P + p]) ? keep_scale : 0.f;
    __syncthreads();
    // online softmax (flash-attention style): running max / normaliser per wave, the weighted sum of the pair
    // products accumulates in registers (lane: pair column n, e runs 8q..8q+7 per k slab) while the logits are made
    float run_m = -INFINITY, run_l = 0.f;
    float ov[KS][8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int k = 0; k < 8; ++k) ov[ks][k] = 0.f;
    for (int pt = wave; pt < PT; pt += 4) {
      const int ij = lut[pt * 16 + n], i = ij >> 16, j = ij & 0xffff;
      afm_f32x4 acc[AT];
#pragma unroll
      for (int mt = 0; mt < AT; ++mt) acc[mt] = afm_f32x4{0.f, 0.f, 0.f, 0.f};
      float pr[KS][8];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float xj[8];
        Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(xs + i * RS + (32 * ks + 8 * q) * 2), pr[ks]);
        Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(xs + j * RS + (32 * ks + 8 * q) * 2), xj);
#pragma unroll
        for (int k = 0; k < 8; ++k) pr[ks][k] *= xj[k];
        const uint4 bf = Vec16<bf16_t>::pack(pr[ks]);
#pragma unroll
        for (int mt = 0; mt < AT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(afm_bf16x8, Wf[mt][ks]),
                                                            __builtin_bit_cast(afm_bf16x8, bf), acc[mt], 0, 0, 0);
      }
      float part = 0.f;
#pragma unroll
      for (int mt = 0; mt < AT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part = fmaf(fmaxf(acc[mt][r] + b1v[mt][r], 0.f), w2v[mt][r], part);
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);
      const bool valid = pt * 16 + n < P;
      const float lgv = valid ? part + bias2 : -INFINITY;
      if (q == 0) lg[pt * 16 + n] = lgv;
      // lane-local online softmax over this lane's pair column (no cross-lane traffic per tile)
      const float new_m = fmaxf(run_m, lgv);
      const float scale = new_m == -INFINITY ? 1.f : __expf(run_m - new_m);      // 0 on the first valid pair
      const float w = valid ? __expf(lgv - new_m) : 0.f;
      run_l = run_l * scale + w;
      run_m = new_m;
      // score dropout scales a pair's weight in the SUM only; the normaliser is the undropped softmax's
      const float wd = keep != nullptr ? w * mk[pt * 16 + n] : w;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int k = 0; k < 8; ++k) ov[ks][k] = fmaf(wd, pr[ks][k], ov[ks][k] * scale);
    }
    // combine the 16 pair columns of each wave and the four waves
    float wm = run_m;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) wm = fmaxf(wm, __shfl_xor(wm, o, 64));
    if (lane == 0) red[wave] = wm;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float f = run_m == -INFINITY ? 0.f : __expf(run_m - m);
    float ls = run_l * f;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) ls += __shfl_xor(ls, o, 64);
    if (lane == 0) red[4 + wave] = ls;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float v = ov[ks][k] * f;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);             // over the 16 pair columns
        if (n == 0) atomicAdd(&vec[32 * ks + 8 * q + k], v);
      }
    __syncthreads();
    const float sum = red[4] + red[5] + red[6] + red[7];
    const float inv = 1.f / sum;
    for (int p = threadIdx.x; p < P; p += 256) {
      const float sc = __expf(lg[p] - m) * inv;
      attn[b * P + p] = from_f32<bf16_t>(sc);
      if (keep != nullptr) attn_drop[b * P + p] = from_f32<bf16_t>(sc * mk[p]);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < E; k += 256) out[b * E + k] = from_f32<bf16_t>(vec[k] * inv);
  }
}

static size_t afm_fwd_mfma_lds(int N, int E) {
  const int P = N * (N - 1) / 2, PT = (P + 15) / 16;
  return (size_t)((N * (E * 2 + 16) + 15) & ~15) + (size_t)PT * 16 * 12 + (size_t)E * 4 + 64;
}

// Conflict-free pair schedule (trs_common.hpp): rounds of pairs that share no field, so the four waves can add into
// the per-field LDS gradient block with plain read-modify-writes inside a round.
__host__ __device__ inline int afm_rounds(int N) { return sched_rounds(N); }
__host__ __device__ inline int afm_width(int N) { return sched_width(N); }
__device__ __forceinline__ void afm_build_schedule(int* sched, int N) {
  const int R = sched_rounds(N), H = sched_width(N);
  for (int t = threadIdx.x; t < R * H; t += blockDim.x) sched[t] = sched_entry(t / H, t % H, N);
}

// Backward.  Per sample: d(score)_p = g_attn_p + g_out . prod_p;  softmax backward;  then per pair
//   dh_a = d(logit)_p w2_a [h_a > 0];   dprod = score_p g_out + W1^T dh;   dx_i += dprod * x_j, dx_j += dprod * x_i
// dW1[a][e] += dh_a prod_e is accumulated in registers (lane = e, AMAX values per e slab) over every pair and sample
// of the workgroup; all parameter gradients leave as partials [grid][A*E + 2A + 1] reduced by a second kernel.
template <typename T, int AMAX, int ES>
__global__ __launch_bounds__(256) void afm_bwd_kernel(const T* __restrict__ g_out, const T* __restrict__ g_attn,
                                                      const T* __restrict__ x, const T* __restrict__ attn,
                                                      const T* __restrict__ W1, const T* __restrict__ b1,
                                                      const T* __restrict__ w2, int64_t B, int N, int E, int A,
                                                      T* __restrict__ gx, float* __restrict__ partial,
                                                      const uint8_t* __restrict__ keep, float keep_scale) {
  extern __shared__ float smem[];
  const int P = N * (N - 1) / 2;
  const AfmLds l = afm_carve(smem, N, E, A, P, true);
  float* gxs = smem + afm_lds_floats(N, E, A, P, true);        // [N][E]
  float* dw1 = gxs + N * E;                                   // [A][E]  (block reduction at the end)
  const int R = afm_rounds(N), H = afm_width(N);
  int* sched = reinterpret_cast<int*>(dw1 + A * E);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  afm_stage_weights(l, W1, b1, w2, E, A, true);
  afm_build_schedule(sched, N);
  for (int k = threadIdx.x; k < A * E; k += 256) dw1[k] = 0.f;
  float dw1r[ES][AMAX];
#pragma unroll
  for (int es = 0; es < ES; ++es)
#pragma unroll
    for (int a = 0; a < AMAX; ++a) dw1r[es][a] = 0.f;
  float db1r[2] = {0.f, 0.f}, dw2r[2] = {0.f, 0.f}, db2r = 0.f;
  float* dh = l.dh + wave * A;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    for (int k = threadIdx.x; k < N * E; k += 256) {
      l.xs[k] = to_f32(x[b * N * E + k]);
      gxs[k] = 0.f;
    }
    for (int k = threadIdx.x; k < E; k += 256) l.vec[k] = g_out != nullptr ? to_f32(g_out[b * E + k]) : 0.f;
    for (int p = threadIdx.x; p < P; p += 256) {
      l.lg[p] = to_f32(attn[b * P + p]);
      l.msk[p] = keep == nullptr ? 1.f : (keep[b * P + p] ? keep_scale : 0.f);
    }
    __syncthreads();
    // d(score): with score dropout the sum and the returned scores see score * msk
    for (int p = wave; p < P; p += 4) {
      int i, j;
      pair_ij(p, N, &i, &j);
      float d = 0.f;
      if (lane < E) d = l.vec[lane] * l.xs[i * E + lane] * l.xs[j * E + lane];
      if (64 + lane < E) d = fmaf(l.vec[64 + lane], l.xs[i * E + 64 + lane] * l.xs[j * E + 64 + lane], d);
      d = wave_sum(d);
      if (lane == 0) l.aux[p] = (d + (g_attn != nullptr ? to_f32(g_attn[b * P + p]) : 0.f)) * l.msk[p];
    }
    __syncthreads();
    float s = 0.f;
    for (int p = threadIdx.x; p < P; p += 256) s += l.lg[p] * l.aux[p];
    s = block_sum(s, l.red);
    for (int p = threadIdx.x; p < P; p += 256) l.aux[p] = l.lg[p] * (l.aux[p] - s);      // d(logit)
    __syncthreads();
    for (int r = 0; r < R; ++r) {
      for (int k = wave; k < H; k += 4) {
        const int ij = sched[r * H + k];
        if (ij < 0) continue;
        const int i = ij >> 16, j = ij & 0xffff, p = i * (2 * N - i - 1) / 2 + j - i - 1;
        const float dl = l.aux[p], sc = l.lg[p] * l.msk[p];
        float acc[2];
        afm_hidden(l, i, j, E, A, lane, acc);
        if (lane < A) {
          const float h = fmaxf(acc[0], 0.f), d = acc[0] > 0.f ? dl * l.w2[lane] : 0.f;
          dh[lane] = d;
          db1r[0] += d;
          dw2r[0] = fmaf(dl, h, dw2r[0]);
        }
        if (64 + lane < A) {
          const float h = fmaxf(acc[1], 0.f), d = acc[1] > 0.f ? dl * l.w2[64 + lane] : 0.f;
          dh[64 + lane] = d;
          db1r[1] += d;
          dw2r[1] = fmaf(dl, h, dw2r[1]);
        }
        if (lane == 0) db2r += dl;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int es = 0; es < ES; ++es) {
          const int e = 64 * es + lane;
          if (e < E) {
            const float xi = l.xs[i * E + e], xj = l.xs[j * E + e], pr = xi * xj;
            float dp = sc * l.vec[e];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
              if (a < A) {
                const float d = dh[a];
                dp = fmaf(d, l.w1[a * E + e], dp);
                dw1r[es][a] = fmaf(d, pr, dw1r[es][a]);
              }
            }
            gxs[i * E + e] = fmaf(dp, xj, gxs[i * E + e]);
            gxs[j * E + e] = fmaf(dp, xi, gxs[j * E + e]);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
    }
    for (int k = threadIdx.x; k < N * E; k += 256) gx[b * N * E + k] = from_f32<T>(gxs[k]);
  }
  __syncthreads();
  // block reduction of the register accumulators (four waves) through LDS, once per workgroup
#pragma unroll
  for (int es = 0; es < ES; ++es) {
    const int e = 64 * es + lane;
    if (e < E) {
#pragma unroll
      for (int a = 0; a < AMAX; ++a)
        if (a < A) atomicAdd(&dw1[a * E + e], dw1r[es][a]);
    }
  }
  __syncthreads();
  float* mine = partial + (size_t)blockIdx.x * (A * E + 2 * A + 1);
  for (int k = threadIdx.x; k < A * E; k += 256) mine[k] = dw1[k];
  __syncthreads();
  float* vb1 = dw1;
  float* vw2 = dw1 + A;
  for (int k = threadIdx.x; k < 2 * A + 1; k += 256) dw1[k] = 0.f;
  __syncthreads();
  if (lane < A) { atomicAdd(&vb1[lane], db1r[0]); atomicAdd(&vw2[lane], dw2r[0]); }
  if (64 + lane < A) { atomicAdd(&vb1[64 + lane], db1r[1]); atomicAdd(&vw2[64 + lane], dw2r[1]); }
  if (lane == 0) atomicAdd(&dw1[2 * A], db2r);
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * A + 1; k += 256) mine[A * E + k] = dw1[k];
}

// Tiles of the backward pass: 16 pairs that share no field (so a wave adds dprod * x into its per-field gradient rows
// with plain read-modify-writes).  The round-robin rounds of trs_common.hpp give such sets, but a round of N = 39 fields
// has 19 pairs = one full tile + one with 3 pairs: 78 tiles for 741 pairs.  A greedy pass over the pairs in round order
// (take a pair when both of its fields are still free in the tile being filled) packs them into ceil(P / 16) = 47 tiles
// for N = 39 (and reaches that bound for every N > 32 tried: 33, 40, 48, 64); for N <= 32 a tile cannot hold more
// disjoint pairs than a round has and the rounds are used as they are.  The table is built once per N on the host and
// travels in the kernel arguments (by value: nothing to upload or keep alive, and a captured launch carries its copy).
constexpr int AFM_TILE_MAX = 112;                       // 3.5 KB of the 4 KB kernel-argument block
struct AfmTiles {
  uint16_t e[AFM_TILE_MAX * 16];                        // (i << 8) | j, 0xffff = empty slot
};

static const AfmTiles* afm_packed_tiles(int N, int* ntiles) {
  struct Entry { AfmTiles t; int n; };
  static std::mutex mu;
  static std::map<int, Entry*> cache;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(N);
  if (it == cache.end()) {
    const int M = (N & 1) ? N : N - 1;                  // players of the circle method
    std::vector<std::pair<int, int>> rem;
    for (int r = 0; r < M; ++r) {
      for (int d = 1; d <= M / 2; ++d) {
        const int a = ((r - d) % M + M) % M, b = (r + d) % M;
        rem.emplace_back(std::min(a, b), std::max(a, b));
      }
      if (!(N & 1)) rem.emplace_back(r, N - 1);
    }
    Entry* en = new Entry;
    for (auto& v : en->t.e) v = 0xffff;
    int nt = 0;
    while (!rem.empty() && nt < AFM_TILE_MAX) {
      uint64_t used = 0;
      int cnt = 0;
      std::vector<std::pair<int, int>> keep;
      for (auto& pr : rem) {
        const uint64_t m = (1ull << pr.first) | (1ull << pr.second);
        if (cnt < 16 && !(used & m)) {
          en->t.e[nt * 16 + cnt++] = (uint16_t)((pr.first << 8) | pr.second);
          used |= m;
        } else {
          keep.push_back(pr);
        }
      }
      rem.swap(keep);
      ++nt;
    }
    en->n = rem.empty() ? nt : 0;                       // 0: does not fit the argument block -> rounds
    it = cache.emplace(N, en).first;
  }
  *ntiles = it->second->n;
  return &it->second->t;
}

// ---------------------------------------------------------------------------------------------
// bf16 fast path, backward: three chained GEMMs per tile of 16 pairs, all on MFMA 16x16x32 bf16.
//   (1) hidden^T[a][p]  = W1 . prod^T              rows of W1 fed in a permuted order so that a lane's outputs are
//                                                   8-element runs of attention units: they ARE the B operand of (2)
//   (2) dprod^T[e][p]   = W1^T . dh                 dh = d(logit)_p w2 [hidden > 0]
//   (3) dW1[a][e]      += dh^T(a x pairs) . prod(pairs x e)      K = pairs: both operands go through a small per-wave
//                                                   LDS transpose (2-byte stores, 16-byte fragment loads), 32 pairs a step
// Tiles are taken from the conflict-free round schedule (16 field-disjoint pairs), every wave owns whole rounds and
// a private fp32 copy of the (N x E) input-gradient block, so dx_i += dprod * x_j needs no atomics at all.
template <int AT /* A/16, even */, int KS /* E/32 */, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void afm_bwd_mfma_kernel(const bf16_t* __restrict__ g_out,
                                                           const bf16_t* __restrict__ g_attn,
                                                           const bf16_t* __restrict__ x, const bf16_t* __restrict__ attn,
                                                           const bf16_t* __restrict__ W1, const bf16_t* __restrict__ b1,
                                                           const bf16_t* __restrict__ w2, int64_t B, int N,
                                                           bf16_t* __restrict__ gx, float* __restrict__ partial,
                                                           const uint8_t* __restrict__ keep, float keep_scale,
                                                           int T /* packed tiles, 0 = the rounds */, AfmTiles packed) {
  constexpr int E = 32 * KS, A = 16 * AT, ET = 2 * KS, AKS = AT / 2, RS = E * 2 + 16, TS = 80, NTHR = 64 * WAVES;
  constexpr int GS = E + 4;      // row stride (floats) of the per-wave gradient blocks.  The 16 lanes of a tile update 16
                                 // different field rows with float4 accesses: with E + 4 floats a row starts at bank
                                 // 4 (i mod 16), so only fields 16 apart collide (E: every row on the same banks; E + 8:
                                 // 8 (i mod 8), half the slots -- 1253 vs 1091 us for the whole backward at B = 8192)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int P = N * (N - 1) / 2, PP = (P + 15) & ~15;
  const int R = afm_rounds(N), H = afm_width(N), TPR = (H + 15) / 16;
  const int NT = T > 0 ? T : R * TPR;            // tiles of a sample
  char* sp = smem_raw;
  char* xs = sp; sp += (N * RS + 15) & ~15;
  float* lg = reinterpret_cast<float*>(sp); sp += PP * 4;
  float* aux = reinterpret_cast<float*>(sp); sp += PP * 4;
  float* msk = reinterpret_cast<float*>(sp); sp += PP * 4;       // score-dropout multipliers (1 without dropout)
  int* lutp = reinterpret_cast<int*>(sp); sp += PP * 4;
  float* go = reinterpret_cast<float*>(sp); sp += E * 4;
  float* red = reinterpret_cast<float*>(sp); sp += 64;
  int* sched = reinterpret_cast<int*>(sp); sp += NT * 16 * 4;
  float* gxs_all = reinterpret_cast<float*>(sp); sp += WAVES * N * GS * 4;
  char* dhT_all = sp; sp += WAVES * A * TS;
  char* prT_all = sp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
  float* gxs = gxs_all + wave * N * GS;
  char* dhT = dhT_all + wave * A * TS;
  char* prT = prT_all + wave * E * TS;
  for (int p = threadIdx.x; p < PP; p += NTHR) {
    int i = 0, j = 1;
    if (p < P) pair_ij(p, N, &i, &j);
    lutp[p] = (i << 16) | j;
  }
  for (int t = threadIdx.x; t < NT * 16; t += NTHR) {
    if (T > 0) {
      const int e = packed.e[t];
      sched[t] = e == 0xffff ? -1 : ((e >> 8) << 16) | (e & 0xff);
    } else {
      const int r = t / (TPR * 16), k = t - r * (TPR * 16);
      sched[t] = k < H ? sched_entry(r, k, N) : -1;
    }
  }
  // resident operands.  amap(t, m): row m of tile t <-> unit 32 (t>>1) + 8 (m>>2) + 4 (t&1) + (m&3)
  uint4 Wf[AT][KS], Vf[ET][AKS];
  float b1v[AT][4], w2v[AT][4];
#pragma unroll
  for (int mt = 0; mt < AT; ++mt) {
    const int arow = 32 * (mt >> 1) + 8 * (n >> 2) + 4 * (mt & 1) + (n & 3);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      Wf[mt][ks] = *reinterpret_cast<const uint4*>(W1 + (size_t)arow * E + 32 * ks + 8 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int a = 32 * (mt >> 1) + 8 * q + 4 * (mt & 1) + r;
      b1v[mt][r] = to_f32(b1[a]);
      w2v[mt][r] = to_f32(w2[a]);
    }
  }
#pragma unroll
  for (int et = 0; et < ET; ++et) {
    const int erow = 32 * (et >> 1) + 8 * (n >> 2) + 4 * (et & 1) + (n & 3);
#pragma unroll
    for (int u = 0; u < AKS; ++u) {
      uint32_t w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t lo = W1[(size_t)(32 * u + 8 * q + 2 * k) * E + erow].v;
        const uint32_t hi = W1[(size_t)(32 * u + 8 * q + 2 * k + 1) * E + erow].v;
        w[k] = lo | (hi << 16);
      }
      Vf[et][u] = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  afm_f32x4 acc3[AT][ET];
#pragma unroll
  for (int mt = 0; mt < AT; ++mt)
#pragma unroll
    for (int et = 0; et < ET; ++et) acc3[mt][et] = afm_f32x4{0.f, 0.f, 0.f, 0.f};
  float db1r[AT][4], dw2r[AT][4], db2r = 0.f;
#pragma unroll
  for (int mt = 0; mt < AT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { db1r[mt][r] = 0.f; dw2r[mt][r] = 0.f; }
  constexpr int VPR = E / 8;

  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    for (int v = threadIdx.x; v < N * VPR; v += NTHR) {
      const int row = v / VPR, col = v - row * VPR;
      *reinterpret_cast<uint4*>(xs + row * RS + col * 16) =
          *reinterpret_cast<const uint4*>(x + (b * N + row) * (int64_t)E + col * 8);
    }
    for (int k = threadIdx.x; k < E; k += NTHR) go[k] = g_out != nullptr ? to_f32(g_out[b * E + k]) : 0.f;
    for (int p = threadIdx.x; p < P; p += NTHR) {
      lg[p] = to_f32(attn[b * P + p]);
      msk[p] = keep == nullptr ? 1.f : (keep[b * P + p] ? keep_scale : 0.f);
    }
    for (int k = lane; k < N * GS; k += 64) gxs[k] = 0.f;
    __syncthreads();
    // d(score)_p = g_attn_p + g_out . prod_p      (one pair per thread, 16-byte row reads)
    for (int p = threadIdx.x; p < P; p += NTHR) {
      const int ij = lutp[p], i = ij >> 16, j = ij & 0xffff;
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < VPR; ++c) {
        float xi[8], xj[8];
        Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(xs + i * RS + c * 16), xi);
        Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(xs + j * RS + c * 16), xj);
#pragma unroll
        for (int k = 0; k < 8; ++k) d = fmaf(go[8 * c + k], xi[k] * xj[k], d);
      }
      aux[p] = (d + (g_attn != nullptr ? to_f32(g_attn[b * P + p]) : 0.f)) * msk[p];
    }
    __syncthreads();
    float s = 0.f;
    for (int p = threadIdx.x; p < P; p += NTHR) s += lg[p] * aux[p];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += red[w];
    for (int p = threadIdx.x; p < P; p += NTHR) aux[p] = lg[p] * (aux[p] - s);       // d(logit)
    __syncthreads();
    int tiles_done = 0;
    {
      for (int tile = wave; tile < NT; tile += WAVES) {
        const int ent = sched[tile * 16 + n];
        const bool valid = ent >= 0;
        const int i = valid ? ent >> 16 : 0, j = valid ? ent & 0xffff : 1;
        const int p = i * (2 * N - i - 1) / 2 + j - i - 1;
        const float dlv = valid ? aux[p] : 0.f, scv = valid ? lg[p] * msk[p] : 0.f;
        float xi[KS][8], xj[KS][8];
        afm_f32x4 acc1[AT];
#pragma unroll
        for (int mt = 0; mt < AT; ++mt) acc1[mt] = afm_f32x4{0.f, 0.f, 0.f, 0.f};
        const int col = 16 * (tiles_done & 1) + n;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(xs + i * RS + (32 * ks + 8 * q) * 2), xi[ks]);
          Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(xs + j * RS + (32 * ks + 8 * q) * 2), xj[ks]);
          float pr[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) pr[k] = xi[ks][k] * xj[ks][k];
          const uint4 bf = Vec16<bf16_t>::pack(pr);
          // the ReLU mask is a discontinuity: a product rounded to bf16 (2^-9) flips it for pre-activations near 0 and
          // every flip moves the gradients by a whole term.  Feed the rounding residual as a second bf16 operand
          // (hi + lo carries ~16 mantissa bits), so the pre-activations match an fp32 evaluation of the bf16 inputs.
          float hi[8];
          Vec16<bf16_t>::unpack(bf, hi);
#pragma unroll
          for (int k = 0; k < 8; ++k) hi[k] = pr[k] - hi[k];
          const uint4 bl = Vec16<bf16_t>::pack(hi);
          // (all tiles with the rounded product, then all with the residual: no MFMA waits for the one before it)
#pragma unroll
          for (int mt = 0; mt < AT; ++mt)
            acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(afm_bf16x8, Wf[mt][ks]),
                                                               __builtin_bit_cast(afm_bf16x8, bf), acc1[mt], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < AT; ++mt)
            acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(afm_bf16x8, Wf[mt][ks]),
                                                               __builtin_bit_cast(afm_bf16x8, bl), acc1[mt], 0, 0, 0);
          // prod^T for GEMM (3): element (e = 32 ks + 8 q + k, pair column col)
          const uint32_t words[4] = {bf.x, bf.y, bf.z, bf.w};
#pragma unroll
          for (int k = 0; k < 8; ++k)
            *reinterpret_cast<uint16_t*>(prT + (32 * ks + 8 * q + k) * TS + col * 2) =
                (uint16_t)(words[k >> 1] >> (16 * (k & 1)));
        }
        // d(hidden), parameter-vector gradients, B operand of GEMM (2)
        uint4 Bdh[AKS];
#pragma unroll
        for (int u = 0; u < AKS; ++u) {
          float run[8];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int mt = 2 * u + h2;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const float pre = acc1[mt][r4] + b1v[mt][r4];
              const float d = pre > 0.f ? dlv * w2v[mt][r4] : 0.f;
              db1r[mt][r4] += d;
              dw2r[mt][r4] = fmaf(dlv, fmaxf(pre, 0.f), dw2r[mt][r4]);
              run[4 * h2 + r4] = d;
            }
          }
          Bdh[u] = Vec16<bf16_t>::pack(run);
          const uint32_t words[4] = {Bdh[u].x, Bdh[u].y, Bdh[u].z, Bdh[u].w};
#pragma unroll
          for (int k = 0; k < 8; ++k)
            *reinterpret_cast<uint16_t*>(dhT + (32 * u + 8 * q + k) * TS + col * 2) =
                (uint16_t)(words[k >> 1] >> (16 * (k & 1)));
        }
        if (q == 0) db2r += dlv;
        // GEMM (2): dprod^T = W1^T . dh
        afm_f32x4 acc2[ET];
#pragma unroll
        for (int et = 0; et < ET; ++et) acc2[et] = afm_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < AKS; ++u)
#pragma unroll
          for (int et = 0; et < ET; ++et)
            acc2[et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(afm_bf16x8, Vf[et][u]),
                                                               __builtin_bit_cast(afm_bf16x8, Bdh[u]), acc2[et], 0, 0, 0);
        if (valid) {
#pragma unroll
          for (int v = 0; v < KS; ++v) {
            const int e0 = 32 * v + 8 * q;
            float* gi = gxs + i * GS + e0;
            float* gj = gxs + j * GS + e0;
            float4 a0 = *reinterpret_cast<float4*>(gi), a1 = *reinterpret_cast<float4*>(gi + 4);
            float4 c0 = *reinterpret_cast<float4*>(gj), c1 = *reinterpret_cast<float4*>(gj + 4);
            float dp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
              dp[k] = (k < 4 ? acc2[2 * v][k & 3] : acc2[2 * v + 1][k & 3]) + scv * go[e0 + k];
            a0.x = fmaf(dp[0], xj[v][0], a0.x); a0.y = fmaf(dp[1], xj[v][1], a0.y);
            a0.z = fmaf(dp[2], xj[v][2], a0.z); a0.w = fmaf(dp[3], xj[v][3], a0.w);
            a1.x = fmaf(dp[4], xj[v][4], a1.x); a1.y = fmaf(dp[5], xj[v][5], a1.y);
            a1.z = fmaf(dp[6], xj[v][6], a1.z); a1.w = fmaf(dp[7], xj[v][7], a1.w);
            c0.x = fmaf(dp[0], xi[v][0], c0.x); c0.y = fmaf(dp[1], xi[v][1], c0.y);
            c0.z = fmaf(dp[2], xi[v][2], c0.z); c0.w = fmaf(dp[3], xi[v][3], c0.w);
            c1.x = fmaf(dp[4], xi[v][4], c1.x); c1.y = fmaf(dp[5], xi[v][5], c1.y);
            c1.z = fmaf(dp[6], xi[v][6], c1.z); c1.w = fmaf(dp[7], xi[v][7], c1.w);
            *reinterpret_cast<float4*>(gi) = a0; *reinterpret_cast<float4*>(gi + 4) = a1;
            *reinterpret_cast<float4*>(gj) = c0; *reinterpret_cast<float4*>(gj + 4) = c1;
          }
        }
        ++tiles_done;
        const bool last = tile + WAVES >= NT;
        if ((tiles_done & 1) == 0 || last) {
          if (tiles_done & 1) {
            // odd tile count: the second half of the 32-pair step is empty
#pragma unroll
            for (int u = 0; u < AKS; ++u)
#pragma unroll
              for (int k = 0; k < 8; ++k)
                *reinterpret_cast<uint16_t*>(dhT + (32 * u + 8 * q + k) * TS + (16 + n) * 2) = 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
              for (int k = 0; k < 8; ++k)
                *reinterpret_cast<uint16_t*>(prT + (32 * ks + 8 * q + k) * TS + (16 + n) * 2) = 0;   // 0 * stale NaN
          }
          __builtin_amdgcn_wave_barrier();
          // GEMM (3): dW1 += dh^T . prod over the 32 staged pairs
          uint4 Bf[ET];
#pragma unroll
          for (int et = 0; et < ET; ++et) Bf[et] = *reinterpret_cast<const uint4*>(prT + (16 * et + n) * TS + q * 16);
#pragma unroll
          for (int mt = 0; mt < AT; ++mt) {
            const uint4 Af = *reinterpret_cast<const uint4*>(dhT + (16 * mt + n) * TS + q * 16);
#pragma unroll
            for (int et = 0; et < ET; ++et)
              acc3[mt][et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(afm_bf16x8, Af),
                                                                     __builtin_bit_cast(afm_bf16x8, Bf[et]), acc3[mt][et],
                                                                     0, 0, 0);
          }
          __builtin_amdgcn_wave_barrier();
          if (tiles_done & 1) ++tiles_done;        // keep the column parity aligned for the next sample
        }
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < N * E; k += NTHR) {
      const int row = k / E, o = row * GS + (k - row * E);
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) t += gxs_all[w * N * GS + o];
      gx[b * N * E + k] = from_f32<bf16_t>(t);
    }
  }
  // ---- parameter gradients of this workgroup -> partial[A*E + 2A + 1]
  __syncthreads();
  float* dw1 = gxs_all;                         // [A][E] + [2A+1], reuses the gradient blocks
  for (int k = threadIdx.x; k < A * E + 2 * A + 1; k += NTHR) dw1[k] = 0.f;
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < AT; ++mt)
#pragma unroll
    for (int et = 0; et < ET; ++et)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&dw1[(16 * mt + 4 * q + r) * E + 16 * et + n], acc3[mt][et][r]);
#pragma unroll
  for (int mt = 0; mt < AT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v1 = db1r[mt][r], v2 = dw2r[mt][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64); }
      if (n == 0) {
        const int a = 32 * (mt >> 1) + 8 * q + 4 * (mt & 1) + r;
        atomicAdd(&dw1[A * E + a], v1);
        atomicAdd(&dw1[A * E + A + a], v2);
      }
    }
  db2r = wave_sum(db2r);
  if (lane == 0) atomicAdd(&dw1[A * E + 2 * A], db2r);
  __syncthreads();
  float* mine = partial + (size_t)blockIdx.x * (A * E + 2 * A + 1);
  for (int k = threadIdx.x; k < A * E + 2 * A + 1; k += NTHR) mine[k] = dw1[k];
}

static size_t afm_bwd_mfma_lds(int N, int E, int A, int T = 0, int waves = 4) {
  const int P = N * (N - 1) / 2, PP = (P + 15) & ~15;
  const int R = afm_rounds(N), H = afm_width(N), TPR = (H + 15) / 16;
  const int NT = T > 0 ? T : R * TPR;
  const size_t grad = std::max<size_t>((size_t)waves * N * (E + 4) * 4, ((size_t)A * E + 2 * A + 1) * 4);
  return (size_t)((N * (E * 2 + 16) + 15) & ~15) + (size_t)PP * 16 + (size_t)E * 4 + 64 + (size_t)NT * 64 + grad +
         (size_t)waves * (A + E) * 80 + 64;
}

__global__ __launch_bounds__(256) void afm_reduce_partials_kernel(const float* __restrict__ part, int nparts, int n,
                                                                  int A, int E, float* __restrict__ gW1,
                                                                  float* __restrict__ gb1, float* __restrict__ gw2,
                                                                  float* __restrict__ gb2) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
    if (i < A * E) gW1[i] += s;
    else if (i < A * E + A) gb1[i - A * E] += s;
    else if (i < A * E + 2 * A) gw2[i - A * E - A] += s;
    else gb2[0] += s;
  }
}

static int afm_grid(int64_t B) { return (int)std::min<int64_t>(B, 256 * 4); }

}  // namespace trs

using namespace trs;

#define TRS_AFM_COMMON(name)                                                                          \
  TRS_REQUIRE(B >= 0 && N >= 0 && E > 0 && A > 0, TRS_EINVAL, name ": bad size");                     \
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, name ": dtype %d", dtype);           \
  TRS_REQUIRE(E <= AFM_MAX && A <= AFM_MAX, TRS_ESHAPE, name ": E = %d, A = %d (both <= %d)", E, A, AFM_MAX)

extern "C" int trs_afm_fwd(const void* x, const void* W1, const void* b1, const void* w2, const void* b2, int64_t B,
                           int32_t N, int32_t E, int32_t A, int32_t dtype, void* out, void* attn, trs_stream_t stream) {
  return trs_afm_fwd_dropout(x, W1, b1, w2, b2, nullptr, 1.f, B, N, E, A, dtype, out, attn, nullptr, stream);
}

extern "C" int trs_afm_fwd_dropout(const void* x, const void* W1, const void* b1, const void* w2, const void* b2,
                                   const uint8_t* keep, float keep_scale, int64_t B, int32_t N, int32_t E, int32_t A,
                                   int32_t dtype, void* out, void* attn, void* attn_drop, trs_stream_t stream) {
  TRS_AFM_COMMON("afm_fwd");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(N >= 2, TRS_ESHAPE, "afm_fwd: needs at least two fields (N = %d)", N);
  TRS_REQUIRE(x && W1 && b1 && w2 && b2 && out && attn, TRS_EINVAL, "afm_fwd: NULL pointer");
  TRS_REQUIRE(keep == nullptr || attn_drop != nullptr, TRS_EINVAL, "afm_fwd: a keep mask needs the attn_drop output");
  const int P = N * (N - 1) / 2;
  hipStream_t s = (hipStream_t)stream;
  static const bool no_mfma = getenv("TRS_AFM_GENERIC") != nullptr;      // tests pin the MFMA path to the generic one
  if (!no_mfma && dtype == TRS_BF16 && (E == 32 || E == 64 || E == 128) && A % 16 == 0 && A <= 128 &&
      afm_fwd_mfma_lds(N, E) <= 64 * 1024 && aligned16(x) && aligned16(W1)) {
    const size_t lds = afm_fwd_mfma_lds(N, E);
    // one round of workgroups, each walking its share of the samples (A = E = 64: 162 registers = 3 workgroups per CU;
    // the fixed 1024 this replaces ran as a round of 768 and one of 256)
#define TRS_AFM_M(AT_, KS_)                                                                                       \
  do {                                                                                                            \
    auto kern = afm_fwd_mfma_kernel<AT_, KS_>;                                                                    \
    static size_t cap_lds = ~(size_t)0;                                                                           \
    static int cap = 0;                                                                                           \
    if (cap_lds != lds) { cap = resident_blocks((const void*)kern, 256, lds); cap_lds = lds; }                    \
    const int grid = (int)std::min<int64_t>(B, cap);                                                              \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)W1, (const bf16_t*)b1, \
                       (const bf16_t*)w2, (const bf16_t*)b2, B, N, (bf16_t*)out, (bf16_t*)attn, keep, keep_scale,  \
                       (bf16_t*)attn_drop);                                                                       \
  } while (0)
#define TRS_AFM_MK(AT_)                                 \
  do {                                                  \
    if (E == 32) TRS_AFM_M(AT_, 1);                     \
    else if (E == 64) TRS_AFM_M(AT_, 2);                \
    else TRS_AFM_M(AT_, 4);                             \
  } while (0)
    switch (A / 16) {
      case 1: TRS_AFM_MK(1); break;
      case 2: TRS_AFM_MK(2); break;
      case 3: TRS_AFM_MK(3); break;
      case 4: TRS_AFM_MK(4); break;
      case 5: TRS_AFM_MK(5); break;
      case 6: TRS_AFM_MK(6); break;
      case 7: TRS_AFM_MK(7); break;
      default: TRS_AFM_MK(8); break;
    }
#undef TRS_AFM_MK
#undef TRS_AFM_M
    return check_launch("afm_fwd(mfma)");
  }
  const size_t lds = afm_lds_floats(N, E, A, P, false) * 4;
  TRS_REQUIRE(lds <= 160 * 1024, TRS_ESHAPE, "afm_fwd: N = %d, E = %d, A = %d need %zu bytes of LDS", N, E, A, lds);
#define TRS_AFM_F(T_)                                                                                                 \
  do {                                                                                                                \
    auto kern = afm_fwd_kernel<T_>;                                                                                   \
    if (lds > 64 * 1024 &&                                                                                            \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)  \
      return check_launch("afm_fwd: LDS attribute");                                                                  \
    hipLaunchKernelGGL(kern, dim3(afm_grid(B)), dim3(256), lds, s, (const T_*)x, (const T_*)W1, (const T_*)b1,        \
                       (const T_*)w2, (const T_*)b2, B, N, E, A, (T_*)out, (T_*)attn, keep, keep_scale,               \
                       (T_*)attn_drop);                                                                               \
  } while (0)
  if (dtype == TRS_F32) TRS_AFM_F(float);
  else TRS_AFM_F(bf16_t);
#undef TRS_AFM_F
  return check_launch("afm_fwd");
}

extern "C" int trs_afm_pair_tiles(int32_t N, uint16_t* tiles, int32_t capacity, int32_t* ntiles) {
  TRS_REQUIRE(N >= 2 && N <= 64 && tiles && ntiles && capacity >= 0, TRS_EINVAL, "afm_pair_tiles: bad argument");
  int T = 0;
  const AfmTiles* t = afm_packed_tiles(N, &T);
  *ntiles = T;
  TRS_REQUIRE(T * 16 <= capacity, TRS_EWORKSPACE, "afm_pair_tiles: %d entries needed, room for %d", T * 16, capacity);
  for (int i = 0; i < T * 16; ++i) tiles[i] = t->e[i];
  return TRS_OK;
}

extern "C" size_t trs_afm_bwd_workspace_bytes(int64_t B, int32_t N, int32_t E, int32_t A) {
  if (B <= 0 || E <= 0 || A <= 0) return 256;
  return (size_t)afm_grid(B) * ((size_t)A * E + 2 * A + 1) * 4 + 256;
}

extern "C" int trs_afm_bwd(const void* g_out, const void* g_attn, const void* x, const void* attn, const void* W1,
                           const void* b1, const void* w2, int64_t B, int32_t N, int32_t E, int32_t A, int32_t dtype,
                           void* gx, float* gW1, float* gb1, float* gw2, float* gb2, void* workspace, size_t ws_bytes,
                           trs_stream_t stream) {
  return trs_afm_bwd_dropout(g_out, g_attn, x, attn, nullptr, 1.f, W1, b1, w2, B, N, E, A, dtype, gx, gW1, gb1, gw2, gb2,
                             workspace, ws_bytes, stream);
}

extern "C" int trs_afm_bwd_dropout(const void* g_out, const void* g_attn, const void* x, const void* attn,
                                   const uint8_t* keep, float keep_scale, const void* W1, const void* b1, const void* w2,
                                   int64_t B, int32_t N, int32_t E, int32_t A, int32_t dtype, void* gx, float* gW1,
                                   float* gb1, float* gw2, float* gb2, void* workspace, size_t ws_bytes,
                                   trs_stream_t stream) {
  TRS_AFM_COMMON("afm_bwd");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(N >= 2, TRS_ESHAPE, "afm_bwd: needs at least two fields (N = %d)", N);
  TRS_REQUIRE(x && attn && W1 && b1 && w2 && gx && gW1 && gb1 && gw2 && gb2, TRS_EINVAL, "afm_bwd: NULL pointer");
  TRS_REQUIRE(workspace != nullptr && ws_bytes >= trs_afm_bwd_workspace_bytes(B, N, E, A), TRS_EWORKSPACE,
              "afm_bwd: workspace too small");
  const int P = N * (N - 1) / 2;
  const size_t lds = (afm_lds_floats(N, E, A, P, true) + (size_t)N * E + (size_t)A * E +
                      (size_t)afm_rounds(N) * afm_width(N)) * 4;
  TRS_REQUIRE(lds <= 160 * 1024, TRS_ESHAPE, "afm_bwd: N = %d, E = %d, A = %d need %zu bytes of LDS", N, E, A, lds);
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  static const bool no_mfma = getenv("TRS_AFM_GENERIC") != nullptr;
  if (!no_mfma && dtype == TRS_BF16 && (E == 32 || E == 64 || E == 128) && A % 32 == 0 && A <= 128 &&
      (A / 16) * (E / 32) <= 12 &&
      afm_bwd_mfma_lds(N, E, A) <= 160 * 1024 && aligned16(x) && aligned16(W1)) {
    int T = 0;
    static const bool rounds_only = getenv("TRS_AFM_ROUNDS") != nullptr;
    const AfmTiles* packed = afm_packed_tiles(N, &T);
    if (N <= 32 || rounds_only) T = 0;               // a round IS a maximal tile there
    // (four waves = one per SIMD: A = 64, E = 64 already holds 388 registers per wave -- W1 / W1^T fragments, the dW1
    // accumulators, both x rows of the tile -- so a second wave per SIMD would spill; measured with six: 536 B of scratch)
    const int waves = 4;
    const size_t mlds = afm_bwd_mfma_lds(N, E, A, T, waves);
    const int mgrid = (int)std::min<int64_t>(B, 256);
    // only the (A / 16, E / 32) combinations the test above admits are instantiated: the others need more than 512
    // registers per wave (125-589 spilled) and were never dispatched
    int rc_launch = 0;
    auto launch = [&](auto at_c, auto ks_c) {
      constexpr int AT_ = decltype(at_c)::value, KS_ = decltype(ks_c)::value;
      if constexpr (AT_ * KS_ <= 12) {
        auto kern = afm_bwd_mfma_kernel<AT_, KS_, 4>;
        if (mlds > 64 * 1024 &&
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds) != hipSuccess) {
          rc_launch = check_launch("afm_bwd: LDS attribute");
          return;
        }
        hipLaunchKernelGGL(kern, dim3(mgrid), dim3(64 * waves), mlds, s, (const bf16_t*)g_out, (const bf16_t*)g_attn,
                           (const bf16_t*)x, (const bf16_t*)attn, (const bf16_t*)W1, (const bf16_t*)b1, (const bf16_t*)w2,
                           B, N, (bf16_t*)gx, part, keep, keep_scale, T, *packed);
      }
    };
    auto launch_at = [&](auto at_c) {
      if (E == 32) launch(at_c, std::integral_constant<int, 1>{});
      else if (E == 64) launch(at_c, std::integral_constant<int, 2>{});
      else launch(at_c, std::integral_constant<int, 4>{});
    };
    switch (A / 32) {
      case 1: launch_at(std::integral_constant<int, 2>{}); break;
      case 2: launch_at(std::integral_constant<int, 4>{}); break;
      case 3: launch_at(std::integral_constant<int, 6>{}); break;
      default: launch_at(std::integral_constant<int, 8>{}); break;
    }
    if (rc_launch != 0) return rc_launch;
    const int n = A * E + 2 * A + 1;
    hipLaunchKernelGGL(afm_reduce_partials_kernel, dim3((n + 255) / 256), dim3(256), 0, s, part, mgrid, n, A, E, gW1,
                       gb1, gw2, gb2);
    return check_launch("afm_bwd(mfma)");
  }
  const int grid = afm_grid(B);
#define TRS_AFM_B(T_, AMAX_, ES_)                                                                                     \
  do {                                                                                                                \
    auto kern = afm_bwd_kernel<T_, AMAX_, ES_>;                                                                       \
    if (lds > 64 * 1024 &&                                                                                            \
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)  \
      return check_launch("afm_bwd: LDS attribute");                                                                  \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, (const T_*)g_out, (const T_*)g_attn, (const T_*)x,        \
                       (const T_*)attn, (const T_*)W1, (const T_*)b1, (const T_*)w2, B, N, E, A, (T_*)gx, part, keep,  \
                       keep_scale);                                                                                   \
  } while (0)
#define TRS_AFM_BT(T_)                                   \
  do {                                                   \
    if (E <= 64) {                                       \
      if (A <= 32) TRS_AFM_B(T_, 32, 1);                 \
      else if (A <= 64) TRS_AFM_B(T_, 64, 1);            \
      else TRS_AFM_B(T_, 128, 1);                        \
    } else {                                             \
      if (A <= 32) TRS_AFM_B(T_, 32, 2);                 \
      else if (A <= 64) TRS_AFM_B(T_, 64, 2);            \
      else TRS_AFM_B(T_, 128, 2);                        \
    }                                                    \
  } while (0)
  if (dtype == TRS_F32) TRS_AFM_BT(float);
  else TRS_AFM_BT(bf16_t);
#undef TRS_AFM_BT
#undef TRS_AFM_B
  const int n = A * E + 2 * A + 1;
  hipLaunchKernelGGL(afm_reduce_partials_kernel, dim3((n + 255) / 256), dim3(256), 0, s, part, grid, n, A, E, gW1, gb1,
                     gw2, gb2);
  return check_launch("afm_bwd");
}
