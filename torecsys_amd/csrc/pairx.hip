// the other layers built on the (i<j) field-pair pattern of the inner product (pair.hip):
//   OuterProductNetworkLayer   layers/ctr/outer_product_network.py:94-129   ('mat' / 'vec' / 'num' kernels)
//   BilinearInteractionLayer   layers/ctr/bilinear_interaction.py:230-255   ('all' / 'each')
//   (AttentionalFactorizationMachineLayer lives in afm.hip)
// The reference gathers p = x[:, row_idx] and q = x[:, col_idx] -- two (B, NC2, E) tensors, 6.2 GB each at
// B = 65 536, N = 39, E = 64 bf16 -- before any arithmetic.  Here a workgroup keeps the (N x E) block of a
// sample in LDS and walks the pairs; nothing of size B*NC2*E exists unless the layer's own output has it.
//
// Kernels in this file (fp32 math, T = float | bf16):
//   pairw_dot_*      out[b,p]   = sum_e x_i[e] x_j[e] k[p,e]              OPN 'vec' / 'num'
//   pair_mul_*       out[b,p,:] = a_i[:] * c_j[:] + bias                   Bilinear 'all' after T = x W (one GEMM)
//   pair_bil_*       T = x_i W_p;  out[b,p] = sum_h T_h x_j[h]  (OPN 'mat')  |  out[b,p,:] = T * x_j + bias_p ('each')
// Lane layout: a group of EL lanes (EL = min(64, pow2 >= E)) owns one pair at a time, lanes run along e / h, so
// per-pair parameter rows are read coalesced and the E-reductions are wavefront reductions (DPP / ds_swizzle).
#include <algorithm>

#include "trs_common.hpp"

namespace trs {

__device__ __forceinline__ float group_reduce(float v, int EL) {
  // sum over the EL-lane group (EL a power of two <= 64)
  for (int o = EL >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T>
__device__ __forceinline__ void stage_block(const T* __restrict__ src, float* __restrict__ dst, int n) {
  for (int e = threadIdx.x; e < n; e += blockDim.x) dst[e] = to_f32(src[e]);
}

static inline int pow2_lanes(int E) {
  int el = 1;
  while (el < E && el < 64) el <<= 1;
  return el;
}

// ---------------------------------------------------------------------------------------------
// OPN 'vec' / 'num':  out[b,p] = sum_e x_i[e] x_j[e] k[p*kp + e*ke]      (vec: kp=E, ke=1; num: kp=1, ke=0)
// (i,j) of every pair, packed (i << 16) | j, built once per workgroup
__device__ __forceinline__ void build_pair_lut(int* lut, int N) {
  const int P = N * (N - 1) / 2;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    int i, j;
    pair_ij(p, N, &i, &j);
    lut[p] = (i << 16) | j;
  }
}

// Conflict-free pair schedule (see trs_common.hpp): sched[r*H + k] = (i << 16) | j, or -1.
__device__ __forceinline__ void build_round_schedule(int* sched, int N) {
  const int R = sched_rounds(N), H = sched_width(N);
  for (int t = threadIdx.x; t < R * H; t += blockDim.x) sched[t] = sched_entry(t / H, t % H, N);
}
__device__ __forceinline__ int pair_index(int i, int j, int N) { return i * (2 * N - i - 1) / 2 + j - i - 1; }

template <typename T>
__global__ __launch_bounds__(256) void pairw_dot_fwd_kernel(const T* __restrict__ x, const T* __restrict__ kern, int kp,
                                                            int ke, int64_t B, int N, int E, int EL,
                                                            T* __restrict__ out) {
  extern __shared__ float smem[];
  float* xs = smem;                      // [N][E]
  const int P = N * (N - 1) / 2;
  int* lut = reinterpret_cast<int*>(smem + N * E);
  const int groups = blockDim.x / EL, grp = threadIdx.x / EL, e0 = threadIdx.x % EL;
  build_pair_lut(lut, N);
  constexpr int U = 4;                   // pairs in flight per lane group: their parameter rows are independent loads
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    stage_block(x + b * N * E, xs, N * E);
    __syncthreads();
    for (int p0 = grp; p0 < P; p0 += groups * U) {
      float kv[U], acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = p0 + u * groups;
        kv[u] = (p < P && e0 < E) ? to_f32(kern[(int64_t)p * kp + e0 * ke]) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = p0 + u * groups;
        acc[u] = 0.f;
        if (p < P) {
          const int ij = lut[p], i = ij >> 16, j = ij & 0xffff;
          if (e0 < E) acc[u] = xs[i * E + e0] * xs[j * E + e0] * kv[u];
          for (int e = e0 + EL; e < E; e += EL)
            acc[u] = fmaf(xs[i * E + e] * xs[j * E + e], to_f32(kern[(int64_t)p * kp + e * ke]), acc[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = p0 + u * groups;
        const float r = group_reduce(acc[u], EL);
        if (e0 == 0 && p < P) out[b * P + p] = from_f32<T>(r);
      }
    }
  }
}

// forward, 16-byte rows: a thread owns whole pairs and walks e in 16-byte chunks -- x_i / x_j chunks from the LDS copy of
// the sample (kept in the table dtype, rows padded by 16 bytes: the lanes of a wave read one x_i row (broadcast) and
// consecutive x_j rows, all on different banks), the k chunk of its pair from L2 -- and sums in registers: no cross-lane
// reduction, 16-byte LDS reads instead of the 4-byte ones of the variant below (which is LDS-issue-bound: 229 us at
// B = 8192, N = 39, E = 64).
template <typename T>
__global__ __launch_bounds__(256) void pairw_dot_fwd_own_kernel(const T* __restrict__ x, const T* __restrict__ kern,
                                                                int is_num, int64_t B, int N, int E, T* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  constexpr int VE = Vec16<T>::VE;
  const int P = N * (N - 1) / 2, vpr = E / VE;
  const int RS = E * (int)sizeof(T) + 16;               // LDS row stride (bytes)
  char* xs = smem_c;
  int* lut = reinterpret_cast<int*>(smem_c + ((N * RS + 15) & ~15));
  build_pair_lut(lut, N);
  const uint4* kv = reinterpret_cast<const uint4*>(kern);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    for (int v = threadIdx.x; v < N * vpr; v += blockDim.x) {
      const int row = v / vpr, c = v - row * vpr;
      *reinterpret_cast<uint4*>(xs + row * RS + c * 16) =
          *(reinterpret_cast<const uint4*>(x + (b * N + row) * (int64_t)E) + c);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
      const int ij = lut[p], i = ij >> 16, j = ij & 0xffff;
      float acc = 0.f;
      for (int c0 = 0; c0 < vpr; c0 += 4) {
        uint4 kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (!is_num && c0 + u < vpr) kk[u] = kv[(int64_t)p * vpr + c0 + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (c0 + u < vpr) {
            float xi[VE], xj[VE];
            Vec16<T>::unpack(*reinterpret_cast<const uint4*>(xs + i * RS + (c0 + u) * 16), xi);
            Vec16<T>::unpack(*reinterpret_cast<const uint4*>(xs + j * RS + (c0 + u) * 16), xj);
            if (is_num) {
#pragma unroll
              for (int q = 0; q < VE; ++q) acc = fmaf(xi[q], xj[q], acc);
            } else {
              float kf[VE];
              Vec16<T>::unpack(kk[u], kf);
#pragma unroll
              for (int q = 0; q < VE; ++q) acc = fmaf(xi[q] * xj[q], kf[q], acc);
            }
          }
        }
      }
      if (is_num) acc *= to_f32(kern[p]);
      out[b * P + p] = from_f32<T>(acc);
    }
  }
}

// Register-resident variant for the common case (E <= 64, NC2 <= 1024): a 1024-thread workgroup in which every LANE
// owns one pair for every sample the workgroup processes.  The lane walks e itself, so there is no cross-lane
// reduction at all; the x block sits in LDS with a row stride of E+1 floats, which spreads the lanes' different
// field rows over the banks (same field -> same address -> broadcast).
// MODE 0 (forward): the lane's parameter row kern[p][0..E) lives in registers -- nothing is re-read per sample -- and
// the results of a sample leave as one coalesced run.  MODE 1 (weight gradient): gk[p][e] += g[b,p] x_i[e] x_j[e]
// accumulates in the same registers over all samples: one pass over the batch, no atomics, no LDS partials;
// per-workgroup partials [grid][NC2][E] are reduced afterwards.
constexpr int PAIRW_EMAX = 64;

template <typename T, int MODE>
__global__ __launch_bounds__(1024) void pairw_reg_kernel(const T* __restrict__ x, const T* __restrict__ kern /* MODE 0 */,
                                                         const T* __restrict__ g /* MODE 1 */, int64_t B, int N, int E,
                                                         T* __restrict__ out /* MODE 0 */,
                                                         float* __restrict__ partial /* MODE 1 */) {
  extern __shared__ float smem[];
  float* xs = smem;                                   // [N][RS]
  // E % 4 == 0: rows padded to E+4 floats and read as float4 (rows of different fields start 4 banks apart: conflict
  // free for 16-byte reads, one field's row is a broadcast) -- a quarter of the LDS instructions of the scalar form
  const bool vec4 = (E & 3) == 0;
  const int P = N * (N - 1) / 2, RS = vec4 ? E + 4 : E + 1;
  const int p = threadIdx.x;
  const bool own = p < P;
  int i = 0, j = 1;
  if (own) pair_ij(p, N, &i, &j);
  float reg[PAIRW_EMAX];
#pragma unroll
  for (int e = 0; e < PAIRW_EMAX; ++e) reg[e] = (MODE == 0 && own && e < E) ? to_f32(kern[(int64_t)p * E + e]) : 0.f;
  const float* xi = xs + i * RS;
  const float* xj = xs + j * RS;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    for (int k = threadIdx.x; k < N * E; k += blockDim.x) {
      const int row = k / E;
      xs[row * RS + (k - row * E)] = to_f32(x[b * N * E + k]);
    }
    __syncthreads();
    if (own) {
      if (MODE == 0) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < PAIRW_EMAX; ++e)
          if (e < E) acc = fmaf(xi[e] * xj[e], reg[e], acc);
        out[b * P + p] = from_f32<T>(acc);
      } else if (vec4) {
        const float gp = to_f32(g[b * P + p]);
#pragma unroll
        for (int e = 0; e < PAIRW_EMAX; e += 4)
          if (e < E) {
            const float4 a = *reinterpret_cast<const float4*>(xi + e), c = *reinterpret_cast<const float4*>(xj + e);
            reg[e] = fmaf(gp, a.x * c.x, reg[e]);
            reg[e + 1] = fmaf(gp, a.y * c.y, reg[e + 1]);
            reg[e + 2] = fmaf(gp, a.z * c.z, reg[e + 2]);
            reg[e + 3] = fmaf(gp, a.w * c.w, reg[e + 3]);
          }
      } else {
        const float gp = to_f32(g[b * P + p]);
#pragma unroll
        for (int e = 0; e < PAIRW_EMAX; ++e)
          if (e < E) reg[e] = fmaf(gp, xi[e] * xj[e], reg[e]);
      }
    }
  }
  if (MODE == 1 && own) {
    float* mine = partial + ((size_t)blockIdx.x * P + p) * E;
#pragma unroll
    for (int e = 0; e < PAIRW_EMAX; ++e)
      if (e < E) mine[e] = reg[e];
  }
}

static bool pairw_reg_ok(int N, int E, bool is_num) {
  const int P = N * (N - 1) / 2;
  return !is_num && E <= PAIRW_EMAX && P <= 1024 && (size_t)N * (E + 4) * 4 <= 64 * 1024;
}
static int pairw_reg_grid(int64_t B) { return (int)std::min<int64_t>(256, std::max<int64_t>(1, B / 4)); }

// data gradient: gx_i[e] += g k x_j[e], gx_j[e] += g k x_i[e]   (per-sample LDS accumulators, conflict-free rounds)
template <typename T>
__global__ __launch_bounds__(256) void pairw_dot_bwd_data_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                                 const T* __restrict__ kern, int kp, int ke, int64_t B,
                                                                 int N, int E, int EL, T* __restrict__ gx) {
  extern __shared__ float smem[];
  float* xs = smem;
  float* gs = smem + N * E;
  const int P = N * (N - 1) / 2;
  const int R = sched_rounds(N), H = sched_width(N);
  int* sched = reinterpret_cast<int*>(smem + 2 * N * E);
  const int groups = blockDim.x / EL, grp = threadIdx.x / EL, e0 = threadIdx.x % EL;
  build_round_schedule(sched, N);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    stage_block(x + b * N * E, xs, N * E);
    for (int e = threadIdx.x; e < N * E; e += blockDim.x) gs[e] = 0.f;
    __syncthreads();
    for (int r = 0; r < R; ++r) {
      for (int k = grp; k < H; k += groups) {
        const int ij = sched[r * H + k];
        if (ij < 0) continue;
        const int i = ij >> 16, j = ij & 0xffff, p = pair_index(i, j, N);
        const float gp = to_f32(g[b * P + p]);
        for (int e = e0; e < E; e += EL) {
          const float w = gp * to_f32(kern[(int64_t)p * kp + e * ke]);
          gs[i * E + e] = fmaf(w, xs[j * E + e], gs[i * E + e]);
          gs[j * E + e] = fmaf(w, xs[i * E + e], gs[j * E + e]);
        }
      }
      __syncthreads();
    }
    for (int e = threadIdx.x; e < N * E; e += blockDim.x) gx[b * N * E + e] = from_f32<T>(gs[e]);
  }
}

// data gradient, owner-computes (16-byte rows): a thread owns one 16-byte chunk v of one gradient row r and sums the N-1
// pairs of its field in registers: gx[r, e] = sum_{o != r} g[p(r,o)] k[p(r,o), e] x[o, e].  No LDS accumulators, no
// rounds (the kernel above runs N barriers per sample), two barriers per sample.  The per-pair scalars w[p] = g[b,p]
// ('vec': k enters per element from L2) or g[b,p] k[p] ('num') are staged in LDS once per sample.
template <typename T>
__global__ __launch_bounds__(256) void pairw_dot_bwd_data_own_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                                     const T* __restrict__ kern, int is_num, int64_t B,
                                                                     int N, int E, T* __restrict__ gx) {
  extern __shared__ float smem[];
  float* xs = smem;                 // [N][E]
  float* ws = smem + N * E;         // [P]
  constexpr int VE = Vec16<T>::VE;
  const int P = N * (N - 1) / 2;
  const int vpr = E / VE;
  const uint4* kv = reinterpret_cast<const uint4*>(kern);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    stage_block(x + b * N * E, xs, N * E);
    for (int p = threadIdx.x; p < P; p += blockDim.x)
      ws[p] = to_f32(g[b * P + p]) * (is_num ? to_f32(kern[p]) : 1.f);
    __syncthreads();
    uint4* go = reinterpret_cast<uint4*>(gx + b * N * (int64_t)E);
    for (int it = threadIdx.x; it < N * vpr; it += blockDim.x) {
      const int r = it / vpr, v = it - r * vpr;
      float acc[VE];
#pragma unroll
      for (int q = 0; q < VE; ++q) acc[q] = 0.f;
      constexpr int U = 4;
      for (int k0 = 0; k0 < N - 1; k0 += U) {
        uint4 kk[U];
        int o[U], pp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = k0 + u < N - 1 ? k0 + u : N - 2;
          o[u] = k < r ? k : k + 1;                                   // the other field
          pp[u] = o[u] < r ? pair_index_of(o[u], r, N) : pair_index_of(r, o[u], N);
          if (!is_num) kk[u] = kv[(int64_t)pp[u] * vpr + v];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (k0 + u < N - 1) {
            const float w = ws[pp[u]];
            const float* xo = xs + o[u] * E + v * VE;
            if (is_num) {
#pragma unroll
              for (int q = 0; q < VE; ++q) acc[q] = fmaf(w, xo[q], acc[q]);
            } else {
              float kf[VE];
              Vec16<T>::unpack(kk[u], kf);
#pragma unroll
              for (int q = 0; q < VE; ++q) acc[q] = fmaf(w * kf[q], xo[q], acc[q]);
            }
          }
        }
      }
      go[it] = Vec16<T>::pack(acc);
    }
  }
}

// weight gradient: gk[p,e] = sum_b g[b,p] x_i[e] x_j[e].  A workgroup owns a range of samples and walks the pairs
// in chunks of PC whose fp32 partial (PC x E) lives in LDS; every (pair, e) slot is owned by one fixed lane, so
// the accumulation needs no atomics and is deterministic.  Partials per workgroup go to the workspace.
template <typename T>
__global__ __launch_bounds__(256) void pairw_dot_bwd_weight_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                                   int64_t B, int N, int E, int EL, int PC,
                                                                   float* __restrict__ partial /* [grid][P][E] */) {
  extern __shared__ float smem[];
  float* xs = smem;                 // [N][E]
  float* acc = smem + N * E;        // [PC][E]
  const int P = N * (N - 1) / 2;
  int* lut = reinterpret_cast<int*>(smem + N * E + PC * E);
  const int groups = blockDim.x / EL, grp = threadIdx.x / EL, e0 = threadIdx.x % EL;
  build_pair_lut(lut, N);
  const int64_t per = (B + gridDim.x - 1) / gridDim.x;
  const int64_t b_lo = (int64_t)blockIdx.x * per, b_hi = b_lo + per < B ? b_lo + per : B;
  float* mine = partial + (size_t)blockIdx.x * P * E;
  for (int p0 = 0; p0 < P; p0 += PC) {
    const int pc = P - p0 < PC ? P - p0 : PC;
    __syncthreads();
    for (int e = threadIdx.x; e < pc * E; e += blockDim.x) acc[e] = 0.f;
    for (int64_t b = b_lo; b < b_hi; ++b) {
      __syncthreads();
      stage_block(x + b * N * E, xs, N * E);
      __syncthreads();
      for (int q = grp; q < pc; q += groups) {
        const int ij = lut[p0 + q], i = ij >> 16, j = ij & 0xffff;
        const float gp = to_f32(g[b * P + p0 + q]);
        for (int e = e0; e < E; e += EL) acc[q * E + e] = fmaf(gp, xs[i * E + e] * xs[j * E + e], acc[q * E + e]);
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < pc * E; e += blockDim.x) mine[(size_t)p0 * E + e] = acc[e];
  }
}

__global__ __launch_bounds__(256) void pairx_reduce_partials_kernel(const float* __restrict__ part, int nparts, int64_t n,
                                                                    float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
    out[i] += s;
  }
}

// ---------------------------------------------------------------------------------------------
// rows product with bias (the stand-alone forward of FieldAllTypeBilinear / FieldEachTypeBilinear, which receive the two
// pair operands ALREADY gathered as (B, P, E) tensors -- bilinear_interaction.py:72-76, 144-149):
//   out[r, e] = a[r, e] * c[r, e] + bias[(bp ? r % P : 0), e]          r over B*P rows
// and its data gradients ga = g * c, gc = g * a.  Plain element-wise streaming passes (any E).
template <typename T>
__global__ __launch_bounds__(256) void rows_mul_bias_fwd_kernel(const T* __restrict__ a, const T* __restrict__ c,
                                                                const T* __restrict__ bias, int bp, int64_t rows, int P,
                                                                int E, T* __restrict__ out) {
  const int64_t total = rows * E, stride = (int64_t)gridDim.x * blockDim.x;
  const bool f32 = total < ((int64_t)1 << 32);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t r = udiv_fast(t, E, f32);
    const int e = (int)(t - r * E);
    float v = to_f32(a[t]) * to_f32(c[t]);
    if (bias != nullptr) v += to_f32(bias[(bp ? (r % P) * E : 0) + e]);
    out[t] = from_f32<T>(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rows_mul_bwd_kernel(const T* __restrict__ g, const T* __restrict__ a,
                                                           const T* __restrict__ c, int64_t total, T* __restrict__ ga,
                                                           T* __restrict__ gc) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const float gv = to_f32(g[t]);
    if (ga != nullptr) ga[t] = from_f32<T>(gv * to_f32(c[t]));
    if (gc != nullptr) gc[t] = from_f32<T>(gv * to_f32(a[t]));
  }
}

// ---------------------------------------------------------------------------------------------
// pair product with optional bias:  out[b,p,:] = a[b,i,:] * c[b,j,:] + bias[p*bp, :]     (bp = 0 shared, 1 per pair)
template <typename T>
__global__ __launch_bounds__(256) void pair_mul_fwd_kernel(const T* __restrict__ a, const T* __restrict__ c,
                                                           const T* __restrict__ bias, int bp, int64_t B, int N, int E,
                                                           T* __restrict__ out) {
  extern __shared__ float smem[];
  float* as = smem;
  float* cs = smem + N * E;
  constexpr int VE = Vec16<T>::VE;
  const int P = N * (N - 1) / 2;
  int* lut = reinterpret_cast<int*>(smem + 2 * N * E);
  const int vpr = E / VE;                     // 16-byte vectors per row (E % VE == 0 checked by the host)
  const int groups = blockDim.x / vpr, grp = threadIdx.x / vpr, v = threadIdx.x % vpr;
  const bool active = grp < groups;
  build_pair_lut(lut, N);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    stage_block(a + b * N * E, as, N * E);
    stage_block(c + b * N * E, cs, N * E);
    __syncthreads();
    if (!active) continue;
    uint4* orow = reinterpret_cast<uint4*>(out + (b * P) * (int64_t)E);
    float bsh[VE];                          // shared bias: loaded once
#pragma unroll
    for (int k = 0; k < VE; ++k) bsh[k] = (bias != nullptr && bp == 0) ? to_f32(bias[v * VE + k]) : 0.f;
    for (int p = grp; p < P; p += groups) {
      const int ij = lut[p], i = ij >> 16, j = ij & 0xffff;
      float r[VE];
#pragma unroll
      for (int k = 0; k < VE; ++k) {
        const int e = v * VE + k;
        r[k] = fmaf(as[i * E + e], cs[j * E + e], bsh[k]);
        if (bias != nullptr && bp != 0) r[k] += to_f32(bias[(int64_t)p * E + e]);
      }
      const uint4 u = Vec16<T>::pack(r);
      __builtin_nontemporal_store(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, u),
                                  reinterpret_cast<__attribute__((ext_vector_type(4))) unsigned*>(orow + (int64_t)p * vpr + v));
    }
  }
}

// ga[b,i,:] = sum_{p: i_p = i} g[b,p,:] c[b,j_p,:],   gc[b,j,:] = sum_{p: j_p = j} g[b,p,:] a[b,i_p,:]
// Owner-computes: a thread owns one 16-byte column chunk v of TWO output rows whose pair counts add up to N-1 (rows i
// and N-2-i of ga, rows j and N-j of gc: every thread walks the same number of pairs) and sums its pairs in registers:
// no LDS accumulators, no zero fill, two barriers per sample.  (The first version walked conflict-free rounds of
// disjoint pairs with read-modify-writes on LDS accumulators -- N barriers per sample, 60 % of the threads busy.)
// g is read twice per sample (94 KB at N = 39, E = 64: the second pass hits L2).
template <typename T>
__device__ __forceinline__ void pair_mul_row_sum(const uint4* __restrict__ grow, const float* __restrict__ other, int N, int E,
                                                 int vpr, int v, int row, bool as_i, float (&acc)[Vec16<T>::VE]) {
  constexpr int VE = Vec16<T>::VE;
  // as_i: pairs (row, j), j = row+1 .. N-1, weights other[j];  else: pairs (i, row), i = 0 .. row-1, weights other[i]
  const int cnt = as_i ? N - 1 - row : row;
  constexpr int U = 4;
  for (int k0 = 0; k0 < cnt; k0 += U) {
    uint4 gv[U];
    int o[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u < cnt ? k0 + u : cnt - 1;
      o[u] = as_i ? row + 1 + k : k;
      const int pidx = as_i ? pair_index_of(row, o[u], N) : pair_index_of(o[u], row, N);
      gv[u] = grow[(int64_t)pidx * vpr + v];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k0 + u < cnt) {
        float f[VE];
        Vec16<T>::unpack(gv[u], f);
#pragma unroll
        for (int q = 0; q < VE; ++q) acc[q] = fmaf(f[q], other[o[u] * E + v * VE + q], acc[q]);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pair_mul_bwd_kernel(const T* __restrict__ g, const T* __restrict__ a,
                                                           const T* __restrict__ c, int64_t B, int N, int E,
                                                           T* __restrict__ ga, T* __restrict__ gc) {
  extern __shared__ float smem[];
  float* as = smem;
  float* cs = smem + N * E;
  constexpr int VE = Vec16<T>::VE;
  const int P = N * (N - 1) / 2;
  const int vpr = E / VE;
  // work items: (kind, row pair, v).  kind 0 = ga rows {r, N-2-r}, r < ceil((N-1)/2) (row N-1 of ga is zero);
  // kind 1 = gc rows {r, N-r}, r = 1 .. ceil((N-1)/2)   (row 0 of gc is zero; r == N-r once when N is even)
  const int half = N / 2;                     // number of row pairs of either kind
  const int items = 2 * half * vpr;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    stage_block(a + b * N * E, as, N * E);
    stage_block(c + b * N * E, cs, N * E);
    __syncthreads();
    const uint4* grow = reinterpret_cast<const uint4*>(g + (b * P) * (int64_t)E);
    uint4* gao = reinterpret_cast<uint4*>(ga + b * N * (int64_t)E);
    uint4* gco = reinterpret_cast<uint4*>(gc + b * N * (int64_t)E);
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
      const int v = it % vpr, rp = (it / vpr) % half, kind = it / (vpr * half);
      float acc[VE];
      if (kind == 0) {
        const int r0 = rp, r1 = N - 2 - rp;
#pragma unroll
        for (int q = 0; q < VE; ++q) acc[q] = 0.f;
        pair_mul_row_sum<T>(grow, cs, N, E, vpr, v, r0, true, acc);
        gao[r0 * vpr + v] = Vec16<T>::pack(acc);
        if (r1 != r0) {
#pragma unroll
          for (int q = 0; q < VE; ++q) acc[q] = 0.f;
          pair_mul_row_sum<T>(grow, cs, N, E, vpr, v, r1, true, acc);
          gao[r1 * vpr + v] = Vec16<T>::pack(acc);
        }
      } else {
        const int r0 = rp + 1, r1 = N - 1 - rp;
#pragma unroll
        for (int q = 0; q < VE; ++q) acc[q] = 0.f;
        pair_mul_row_sum<T>(grow, as, N, E, vpr, v, r0, false, acc);
        gco[r0 * vpr + v] = Vec16<T>::pack(acc);
        if (r1 != r0) {
#pragma unroll
          for (int q = 0; q < VE; ++q) acc[q] = 0.f;
          pair_mul_row_sum<T>(grow, as, N, E, vpr, v, r1, false, acc);
          gco[r1 * vpr + v] = Vec16<T>::pack(acc);
        }
      }
    }
    // the rows without pairs
    for (int v = threadIdx.x; v < vpr; v += blockDim.x) {
      gao[(N - 1) * vpr + v] = make_uint4(0, 0, 0, 0);
      gco[v] = make_uint4(0, 0, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// per-pair bilinear form, generic path:  T[h] = sum_e x_i[e] W[p*wp][e][h]
//   MODE 0 (OPN 'mat'):        out[b,p]   = sum_h T[h] x_j[h]
//   MODE 1 (Bilinear 'each'):  out[b,p,h] = T[h] x_j[h] + bias[p*bp][h]
// A workgroup stages S samples and every wave walks the pairs with its lanes along h, so a row of W_p is read
// once (coalesced) per S samples.  This is the any-shape / fp32 path; bf16 with E = 64 takes pair_bil_mfma_*.
template <typename T, int MODE, int S>
__global__ __launch_bounds__(256) void pair_bil_fwd_kernel(const T* __restrict__ x, const T* __restrict__ W, int wp,
                                                           const T* __restrict__ bias, int bp, int64_t B, int N, int E,
                                                           T* __restrict__ out) {
  extern __shared__ float smem[];
  float* xs = smem;                    // [S][N][E]
  const int P = N * (N - 1) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (int64_t b0 = (int64_t)blockIdx.x * S; b0 < B; b0 += (int64_t)gridDim.x * S) {
    const int ns = (int)(B - b0 < S ? B - b0 : S);
    __syncthreads();
    stage_block(x + b0 * N * E, xs, ns * N * E);
    __syncthreads();
    for (int p = wave; p < P; p += nwaves) {
      int i, j;
      pair_ij(p, N, &i, &j);
      const T* Wp = W + (int64_t)p * wp * E * E;
      float rsum[S];
#pragma unroll
      for (int s = 0; s < S; ++s) rsum[s] = 0.f;
      for (int h0 = 0; h0 < E; h0 += 64) {
        const int h = h0 + lane;
        float t[S];
#pragma unroll
        for (int s = 0; s < S; ++s) t[s] = 0.f;
        if (h < E) {
          for (int e = 0; e < E; ++e) {
            const float w = to_f32(Wp[(int64_t)e * E + h]);
#pragma unroll
            for (int s = 0; s < S; ++s) t[s] = fmaf(xs[(s * N + i) * E + e], w, t[s]);
          }
#pragma unroll
          for (int s = 0; s < S; ++s) {
            if (s < ns) {
              const float tv = t[s] * xs[(s * N + j) * E + h];
              if (MODE == 0) {
                rsum[s] += tv;
              } else {
                float r = tv;
                if (bias != nullptr) r += to_f32(bias[(int64_t)p * bp * E + h]);
                out[((b0 + s) * P + p) * (int64_t)E + h] = from_f32<T>(r);
              }
            }
          }
        }
      }
      if (MODE == 0) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float r = group_reduce(rsum[s], 64);        // wavefront reduction over h
          if (lane == 0 && s < ns) out[(b0 + s) * P + p] = from_f32<T>(r);
        }
      }
    }
  }
}

// data gradient of the bilinear form (recomputes T):
//   MODE 0: gT[h] = g[b,p] x_j[h];    gx_j[h] += g[b,p] T[h]
//   MODE 1: gT[h] = g[b,p,h] x_j[h];  gx_j[h] += g[b,p,h] T[h]
//   gx_i[e] += sum_h gT[h] W_p[e][h];  gT is also written out (B,P,E) when gT_out != NULL (weight gradient GEMMs)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void pair_bil_bwd_data_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                                const T* __restrict__ W, int wp, int64_t B, int N,
                                                                int E, T* __restrict__ gx, T* __restrict__ gT_out) {
  extern __shared__ float smem[];
  float* xs = smem;                    // [N][E]
  float* gs = smem + N * E;            // [N][E]
  float* gts = smem + 2 * N * E;       // [nwaves][E]
  const int P = N * (N - 1) / 2;
  const int R = sched_rounds(N), H = sched_width(N);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  int* sched = reinterpret_cast<int*>(gts + nwaves * E);
  float* gt = gts + wave * E;
  build_round_schedule(sched, N);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    stage_block(x + b * N * E, xs, N * E);
    for (int e = threadIdx.x; e < N * E; e += blockDim.x) gs[e] = 0.f;
    __syncthreads();
    for (int r = 0; r < R; ++r) {
      for (int k = wave; k < H; k += nwaves) {
        const int ij = sched[r * H + k];
        if (ij < 0) continue;
        const int i = ij >> 16, j = ij & 0xffff, p = pair_index(i, j, N);
        const T* Wp = W + (int64_t)p * wp * E * E;
        for (int h = lane; h < E; h += 64) {
          float t = 0.f;
          for (int e = 0; e < E; ++e) t = fmaf(xs[i * E + e], to_f32(Wp[(int64_t)e * E + h]), t);
          const float gv = MODE == 0 ? to_f32(g[b * P + p]) : to_f32(g[(b * P + p) * (int64_t)E + h]);
          const float gth = gv * xs[j * E + h];
          gt[h] = gth;
          if (gT_out != nullptr) gT_out[(b * P + p) * (int64_t)E + h] = from_f32<T>(gth);
          gs[j * E + h] = fmaf(gv, t, gs[j * E + h]);
        }
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < E; e += 64) {
          float acc = 0.f;
          const T* row = Wp + (int64_t)e * E;
          for (int h = 0; h < E; ++h) acc = fmaf(gt[h], to_f32(row[h]), acc);
          gs[i * E + e] += acc;
        }
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
    }
    for (int e = threadIdx.x; e < N * E; e += blockDim.x) gx[b * N * E + e] = from_f32<T>(gs[e]);
  }
}

// ---------------------------------------------------------------------------------------------
// Epilogues for the GEMM route of the per-pair bilinear form.  At training batch sizes T[b,p,:] = x[b,i_p,:] W_p is
// produced by one plain GEMM per field i (all pairs (i, j>i) are adjacent columns: (B x E) @ (E x n_i*E), hipBLASLt)
// into a (B, NC2, E) buffer; these passes turn it into the layer output and, in the backward, into dL/dT (in
// place) and the x_j half of the input gradient.
//   fwd  MODE 0: out[b,p]  = sum_h T[b,p,h] x[b,j_p,h]          MODE 1: T[b,p,h] = T[b,p,h] x[b,j_p,h] + bias (in place)
//   bwd  MODE 0: gv = g[b,p]   MODE 1: gv = g[b,p,h]:   gxj[b,j_p,h] += gv T[b,p,h];   T[b,p,h] <- gv x[b,j_p,h]
template <typename T, int MODE>
__global__ __launch_bounds__(256) void pair_epi_fwd_kernel(T* __restrict__ Tb, const T* __restrict__ x,
                                                           const T* __restrict__ bias, int bp, int64_t B, int N, int E,
                                                           T* __restrict__ out) {
  extern __shared__ float smem[];
  float* xs = smem;
  constexpr int VE = Vec16<T>::VE;
  const int P = N * (N - 1) / 2;
  int* lut = reinterpret_cast<int*>(smem + N * E);
  const int vpr = E / VE;
  const int groups = blockDim.x / vpr, grp = threadIdx.x / vpr, v = threadIdx.x % vpr;
  const bool active = grp < groups;
  build_pair_lut(lut, N);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    stage_block(x + b * N * E, xs, N * E);
    __syncthreads();
    if (!active) continue;
    uint4* trow = reinterpret_cast<uint4*>(Tb + (b * P) * (int64_t)E);
    for (int p = grp; p < P; p += groups) {
      const int j = lut[p] & 0xffff;
      float t[VE];
      Vec16<T>::unpack(trow[(int64_t)p * vpr + v], t);
      if (MODE == 0) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < VE; ++k) acc = fmaf(t[k], xs[j * E + v * VE + k], acc);
        for (int o = vpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);      // vpr is a power of two (host)
        if (v == 0) out[b * P + p] = from_f32<T>(acc);
      } else {
#pragma unroll
        for (int k = 0; k < VE; ++k) {
          const int e = v * VE + k;
          t[k] = t[k] * xs[j * E + e];
          if (bias != nullptr) t[k] += to_f32(bias[(int64_t)p * bp * E + e]);
        }
        trow[(int64_t)p * vpr + v] = Vec16<T>::pack(t);
      }
    }
  }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void pair_epi_bwd_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                           T* __restrict__ Tb, int64_t B, int N, int E,
                                                           T* __restrict__ gxj) {
  extern __shared__ float smem[];
  float* xs = smem;
  float* gs = smem + N * E;
  constexpr int VE = Vec16<T>::VE;
  const int P = N * (N - 1) / 2;
  const int R = sched_rounds(N), H = sched_width(N);
  int* sched = reinterpret_cast<int*>(smem + 2 * N * E);
  const int vpr = E / VE;
  const int groups = blockDim.x / vpr, grp = threadIdx.x / vpr, v = threadIdx.x % vpr;
  const bool active = grp < groups;
  build_round_schedule(sched, N);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    stage_block(x + b * N * E, xs, N * E);
    for (int e = threadIdx.x; e < N * E; e += blockDim.x) gs[e] = 0.f;
    __syncthreads();
    uint4* trow = reinterpret_cast<uint4*>(Tb + (b * P) * (int64_t)E);
    const uint4* grow = reinterpret_cast<const uint4*>(g + (b * P) * (int64_t)E);      // MODE 1 only
    for (int r = 0; r < R; ++r) {
      if (active) {
        for (int k = grp; k < H; k += groups) {
          const int ij = sched[r * H + k];
          if (ij < 0) continue;
          const int i = ij >> 16, j = ij & 0xffff, p = pair_index(i, j, N);
          float t[VE], gv[VE];
          Vec16<T>::unpack(trow[(int64_t)p * vpr + v], t);
          if (MODE == 0) {
            const float gp = to_f32(g[b * P + p]);
#pragma unroll
            for (int q = 0; q < VE; ++q) gv[q] = gp;
          } else {
            Vec16<T>::unpack(grow[(int64_t)p * vpr + v], gv);
          }
#pragma unroll
          for (int q = 0; q < VE; ++q) {
            const int e = v * VE + q;
            gs[j * E + e] = fmaf(gv[q], t[q], gs[j * E + e]);
            t[q] = gv[q] * xs[j * E + e];
          }
          trow[(int64_t)p * vpr + v] = Vec16<T>::pack(t);
        }
      }
      __syncthreads();
    }
    for (int e = threadIdx.x; e < N * E; e += blockDim.x) gxj[b * N * E + e] = from_f32<T>(gs[e]);
  }
}

static int sample_grid(int64_t B, int per_block = 1) {
  const int64_t need = (B + per_block - 1) / per_block;
  return (int)std::min<int64_t>(need, 256 * 8);
}

}  // namespace trs

using namespace trs;

#define TRS_PAIRX_COMMON(name)                                                                             \
  TRS_REQUIRE(B >= 0 && N >= 0 && E > 0, TRS_EINVAL, name ": bad size");                                   \
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, name ": dtype %d", dtype);                \
  TRS_REQUIRE(N <= 256, TRS_ESHAPE, name ": N = %d > 256 fields", N)

extern "C" int trs_opn_vec_fwd(const void* x, const void* kern, int32_t kern_is_num, int64_t B, int32_t N, int32_t E,
                               int32_t dtype, void* out, trs_stream_t stream) {
  TRS_PAIRX_COMMON("opn_vec_fwd");
  if (B == 0 || N < 2) return TRS_OK;
  TRS_REQUIRE(x && kern && out, TRS_EINVAL, "opn_vec_fwd: NULL pointer");
  const size_t lds = (size_t)N * E * 4 + (size_t)N * (N - 1) / 2 * 4;
  TRS_REQUIRE(lds <= 64 * 1024, TRS_ESHAPE, "opn_vec_fwd: N = %d, E = %d exceed the 64 KB LDS block", N, E);
  const int EL = pow2_lanes(E), kp = kern_is_num ? 1 : E, ke = kern_is_num ? 0 : 1;
  hipStream_t s = (hipStream_t)stream;
  const int VEf = dtype == TRS_F32 ? 4 : 8;
  const size_t own_lds = (size_t)((N * (E * (dtype == TRS_F32 ? 4 : 2) + 16) + 15) & ~15) + (size_t)N * (N - 1) / 2 * 4;
  if (E % VEf == 0 && aligned16(x) && (kern_is_num || aligned16(kern)) && own_lds <= 64 * 1024) {
    if (dtype == TRS_F32)
      hipLaunchKernelGGL((pairw_dot_fwd_own_kernel<float>), dim3(sample_grid(B)), dim3(256), own_lds, s, (const float*)x,
                         (const float*)kern, kern_is_num ? 1 : 0, B, N, E, (float*)out);
    else
      hipLaunchKernelGGL((pairw_dot_fwd_own_kernel<bf16_t>), dim3(sample_grid(B)), dim3(256), own_lds, s,
                         (const bf16_t*)x, (const bf16_t*)kern, kern_is_num ? 1 : 0, B, N, E, (bf16_t*)out);
    return check_launch("opn_vec_fwd(own)");
  }
  if (pairw_reg_ok(N, E, kern_is_num != 0) && B >= 64) {
    const size_t rl = (size_t)N * (E + 4) * 4;
    if (dtype == TRS_F32)
      hipLaunchKernelGGL((pairw_reg_kernel<float, 0>), dim3(pairw_reg_grid(B)), dim3(1024), rl, s, (const float*)x,
                         (const float*)kern, (const float*)nullptr, B, N, E, (float*)out, (float*)nullptr);
    else
      hipLaunchKernelGGL((pairw_reg_kernel<bf16_t, 0>), dim3(pairw_reg_grid(B)), dim3(1024), rl, s, (const bf16_t*)x,
                         (const bf16_t*)kern, (const bf16_t*)nullptr, B, N, E, (bf16_t*)out, (float*)nullptr);
    return check_launch("opn_vec_fwd(reg)");
  }
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((pairw_dot_fwd_kernel<float>), dim3(sample_grid(B)), dim3(256), lds, s, (const float*)x,
                       (const float*)kern, kp, ke, B, N, E, EL, (float*)out);
  else
    hipLaunchKernelGGL((pairw_dot_fwd_kernel<bf16_t>), dim3(sample_grid(B)), dim3(256), lds, s, (const bf16_t*)x,
                       (const bf16_t*)kern, kp, ke, B, N, E, EL, (bf16_t*)out);
  return check_launch("opn_vec_fwd");
}

static int opn_vec_weight_blocks(int64_t B) { return (int)std::min<int64_t>(256, std::max<int64_t>(1, B / 16)); }
static int opn_vec_ws_blocks(int64_t B) { return std::max(opn_vec_weight_blocks(B), pairw_reg_grid(B)); }
static int opn_vec_chunk(int N, int E) {
  const int budget = 48 * 1024 / 4 - N * E - N * (N - 1) / 2;   // floats left after the x block and the pair table
  return std::max(1, budget / E);
}

extern "C" size_t trs_opn_vec_bwd_workspace_bytes(int64_t B, int32_t N, int32_t E) {
  if (B <= 0 || N < 2 || E <= 0) return 256;
  return (size_t)opn_vec_ws_blocks(B) * (size_t)(N * (N - 1) / 2) * E * 4 + 256;
}

extern "C" int trs_opn_vec_bwd(const void* g, const void* x, const void* kern, int32_t kern_is_num, int64_t B, int32_t N,
                               int32_t E, int32_t dtype, void* gx, float* gkern_vec, void* workspace, size_t ws_bytes,
                               trs_stream_t stream) {
  TRS_PAIRX_COMMON("opn_vec_bwd");
  if (B == 0 || N < 2) return TRS_OK;
  TRS_REQUIRE(g && x && kern, TRS_EINVAL, "opn_vec_bwd: NULL pointer");
  const size_t lut_bytes = (size_t)N * (N - 1) / 2 * 4;
  const size_t sched_bytes = (size_t)sched_rounds(N) * sched_width(N) * 4;
  TRS_REQUIRE((size_t)N * E * 4 * 2 + sched_bytes <= 64 * 1024, TRS_ESHAPE,
              "opn_vec_bwd: N = %d, E = %d exceed the 64 KB LDS block", N, E);
  const int EL = pow2_lanes(E), kp = kern_is_num ? 1 : E, ke = kern_is_num ? 0 : 1;
  hipStream_t s = (hipStream_t)stream;
  const int VEb = dtype == TRS_F32 ? 4 : 8;
  if (gx != nullptr && E % VEb == 0 && aligned16(gx) && aligned16(x) && (kern_is_num || aligned16(kern)) &&
      (size_t)(N * E + N * (N - 1) / 2) * 4 <= 64 * 1024) {
    const size_t lds = (size_t)(N * E + N * (N - 1) / 2) * 4;
    if (dtype == TRS_F32)
      hipLaunchKernelGGL((pairw_dot_bwd_data_own_kernel<float>), dim3(sample_grid(B)), dim3(256), lds, s, (const float*)g,
                         (const float*)x, (const float*)kern, kern_is_num ? 1 : 0, B, N, E, (float*)gx);
    else
      hipLaunchKernelGGL((pairw_dot_bwd_data_own_kernel<bf16_t>), dim3(sample_grid(B)), dim3(256), lds, s,
                         (const bf16_t*)g, (const bf16_t*)x, (const bf16_t*)kern, kern_is_num ? 1 : 0, B, N, E,
                         (bf16_t*)gx);
  } else if (gx != nullptr) {
    const size_t lds = (size_t)N * E * 4 * 2 + sched_bytes;
    if (dtype == TRS_F32)
      hipLaunchKernelGGL((pairw_dot_bwd_data_kernel<float>), dim3(sample_grid(B)), dim3(256), lds, s, (const float*)g,
                         (const float*)x, (const float*)kern, kp, ke, B, N, E, EL, (float*)gx);
    else
      hipLaunchKernelGGL((pairw_dot_bwd_data_kernel<bf16_t>), dim3(sample_grid(B)), dim3(256), lds, s,
                         (const bf16_t*)g, (const bf16_t*)x, (const bf16_t*)kern, kp, ke, B, N, E, EL, (bf16_t*)gx);
  }
  if (gkern_vec != nullptr) {
    // gkern_vec: (NC2, E) fp32, accumulated into ('num': the caller sums it over e)
    TRS_REQUIRE(workspace != nullptr && ws_bytes >= trs_opn_vec_bwd_workspace_bytes(B, N, E), TRS_EWORKSPACE,
                "opn_vec_bwd: workspace too small");
    const int P = N * (N - 1) / 2;
    int nblk = opn_vec_weight_blocks(B);
    const int PC = std::min(P, opn_vec_chunk(N, E));
    const size_t lds = (size_t)(N * E + PC * E) * 4 + lut_bytes;
    float* part = (float*)workspace;
    if (pairw_reg_ok(N, E, false) && B >= 64) {
      nblk = pairw_reg_grid(B);
      const size_t rl = (size_t)N * (E + 4) * 4;
      if (dtype == TRS_F32)
        hipLaunchKernelGGL((pairw_reg_kernel<float, 1>), dim3(nblk), dim3(1024), rl, s, (const float*)x,
                           (const float*)nullptr, (const float*)g, B, N, E, (float*)nullptr, part);
      else
        hipLaunchKernelGGL((pairw_reg_kernel<bf16_t, 1>), dim3(nblk), dim3(1024), rl, s, (const bf16_t*)x,
                           (const bf16_t*)nullptr, (const bf16_t*)g, B, N, E, (bf16_t*)nullptr, part);
    } else if (dtype == TRS_F32)
      hipLaunchKernelGGL((pairw_dot_bwd_weight_kernel<float>), dim3(nblk), dim3(256), lds, s, (const float*)g,
                         (const float*)x, B, N, E, EL, PC, part);
    else
      hipLaunchKernelGGL((pairw_dot_bwd_weight_kernel<bf16_t>), dim3(nblk), dim3(256), lds, s, (const bf16_t*)g,
                         (const bf16_t*)x, B, N, E, EL, PC, part);
    const int64_t n = (int64_t)P * E;
    hipLaunchKernelGGL(pairx_reduce_partials_kernel, dim3((int)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, s,
                       part, nblk, n, gkern_vec);
  }
  return check_launch("opn_vec_bwd");
}

extern "C" int trs_rows_mul_bias_fwd(const void* a, const void* c, const void* bias, int32_t bias_per_pair, int64_t rows,
                                     int32_t P, int32_t E, int32_t dtype, void* out, trs_stream_t stream) {
  TRS_REQUIRE(rows >= 0 && P > 0 && E > 0, TRS_EINVAL, "rows_mul_bias_fwd: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "rows_mul_bias_fwd: dtype %d", dtype);
  if (rows == 0) return TRS_OK;
  TRS_REQUIRE(a && c && out, TRS_EINVAL, "rows_mul_bias_fwd: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  const int grid = stream_grid(rows * E, 256);
  const int bp = bias_per_pair ? 1 : 0;
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((rows_mul_bias_fwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)a, (const float*)c,
                       (const float*)bias, bp, rows, P, E, (float*)out);
  else
    hipLaunchKernelGGL((rows_mul_bias_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)a,
                       (const bf16_t*)c, (const bf16_t*)bias, bp, rows, P, E, (bf16_t*)out);
  return check_launch("rows_mul_bias_fwd");
}

extern "C" int trs_rows_mul_bwd(const void* g, const void* a, const void* c, int64_t rows, int32_t E, int32_t dtype,
                                void* ga, void* gc, trs_stream_t stream) {
  TRS_REQUIRE(rows >= 0 && E > 0, TRS_EINVAL, "rows_mul_bwd: bad size");
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, "rows_mul_bwd: dtype %d", dtype);
  if (rows == 0 || (ga == nullptr && gc == nullptr)) return TRS_OK;
  TRS_REQUIRE(g && a && c, TRS_EINVAL, "rows_mul_bwd: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  const int grid = stream_grid(rows * E, 256);
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((rows_mul_bwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)g, (const float*)a,
                       (const float*)c, rows * E, (float*)ga, (float*)gc);
  else
    hipLaunchKernelGGL((rows_mul_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)a,
                       (const bf16_t*)c, rows * E, (bf16_t*)ga, (bf16_t*)gc);
  return check_launch("rows_mul_bwd");
}

extern "C" int trs_pair_mul_fwd(const void* a, const void* c, const void* bias, int32_t bias_per_pair, int64_t B,
                                int32_t N, int32_t E, int32_t dtype, void* out, trs_stream_t stream) {
  TRS_PAIRX_COMMON("pair_mul_fwd");
  if (B == 0 || N < 2) return TRS_OK;
  TRS_REQUIRE(a && c && out, TRS_EINVAL, "pair_mul_fwd: NULL pointer");
  const int VE = dtype == TRS_F32 ? 4 : 8;
  TRS_REQUIRE(E % VE == 0 && E / VE <= 256, TRS_ESHAPE, "pair_mul_fwd: E = %d must be a multiple of %d (16-byte rows)", E,
              VE);
  TRS_REQUIRE(aligned16(a) && aligned16(c) && aligned16(out), TRS_EALIGN, "pair_mul_fwd: pointers must be 16-byte aligned");
  const size_t lds = (size_t)N * E * 4 * 2 + (size_t)N * (N - 1) / 2 * 4;
  TRS_REQUIRE(lds <= 64 * 1024, TRS_ESHAPE, "pair_mul_fwd: N = %d, E = %d exceed the 64 KB LDS block", N, E);
  hipStream_t s = (hipStream_t)stream;
  const int bp = bias_per_pair ? 1 : 0;
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((pair_mul_fwd_kernel<float>), dim3(sample_grid(B)), dim3(256), lds, s, (const float*)a,
                       (const float*)c, (const float*)bias, bp, B, N, E, (float*)out);
  else
    hipLaunchKernelGGL((pair_mul_fwd_kernel<bf16_t>), dim3(sample_grid(B)), dim3(256), lds, s, (const bf16_t*)a,
                       (const bf16_t*)c, (const bf16_t*)bias, bp, B, N, E, (bf16_t*)out);
  return check_launch("pair_mul_fwd");
}

extern "C" int trs_pair_mul_bwd(const void* g, const void* a, const void* c, int64_t B, int32_t N, int32_t E,
                                int32_t dtype, void* ga, void* gc, trs_stream_t stream) {
  TRS_PAIRX_COMMON("pair_mul_bwd");
  if (B == 0 || N < 2) return TRS_OK;
  TRS_REQUIRE(g && a && c && ga && gc, TRS_EINVAL, "pair_mul_bwd: NULL pointer");
  const int VE = dtype == TRS_F32 ? 4 : 8;
  TRS_REQUIRE(E % VE == 0 && E / VE <= 256, TRS_ESHAPE, "pair_mul_bwd: E = %d must be a multiple of %d (16-byte rows)", E,
              VE);
  TRS_REQUIRE(aligned16(g) && aligned16(ga) && aligned16(gc), TRS_EALIGN, "pair_mul_bwd: g, ga, gc must be 16-byte aligned");
  const size_t lds = (size_t)N * E * 4 * 2;
  TRS_REQUIRE(lds <= 64 * 1024, TRS_ESHAPE, "pair_mul_bwd: N = %d, E = %d exceed the 64 KB LDS block", N, E);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == TRS_F32)
    hipLaunchKernelGGL((pair_mul_bwd_kernel<float>), dim3(sample_grid(B)), dim3(256), lds, s, (const float*)g,
                       (const float*)a, (const float*)c, B, N, E, (float*)ga, (float*)gc);
  else
    hipLaunchKernelGGL((pair_mul_bwd_kernel<bf16_t>), dim3(sample_grid(B)), dim3(256), lds, s, (const bf16_t*)g,
                       (const bf16_t*)a, (const bf16_t*)c, B, N, E, (bf16_t*)ga, (bf16_t*)gc);
  return check_launch("pair_mul_bwd");
}

extern "C" int trs_pair_bilinear_fwd(const void* x, const void* W, int32_t w_per_pair, const void* bias,
                                     int32_t bias_per_pair, int32_t mode, int64_t B, int32_t N, int32_t E, int32_t dtype,
                                     void* out, trs_stream_t stream) {
  TRS_PAIRX_COMMON("pair_bilinear_fwd");
  if (B == 0 || N < 2) return TRS_OK;
  TRS_REQUIRE(x && W && out, TRS_EINVAL, "pair_bilinear_fwd: NULL pointer");
  TRS_REQUIRE(mode == 0 || mode == 1, TRS_EINVAL, "pair_bilinear_fwd: mode %d (0 = sum over h, 1 = per-h output)", mode);
  TRS_REQUIRE((size_t)N * E * 4 <= 64 * 1024, TRS_ESHAPE, "pair_bilinear_fwd: N*E = %d exceeds the LDS block", N * E);
  const int S = (size_t)N * E * 4 * 4 <= 64 * 1024 && B >= 4 ? 4 : 1;
  const size_t lds = (size_t)S * N * E * 4;
  const int wp = w_per_pair ? 1 : 0, bp = bias_per_pair ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  const int grid = sample_grid(B, S);
#define TRS_BIL(T_, MODE_, S_)                                                                                       \
  hipLaunchKernelGGL((pair_bil_fwd_kernel<T_, MODE_, S_>), dim3(grid), dim3(256), lds, s, (const T_*)x, (const T_*)W, \
                     wp, (const T_*)bias, bp, B, N, E, (T_*)out)
  if (dtype == TRS_F32) {
    if (mode == 0) { if (S == 4) TRS_BIL(float, 0, 4); else TRS_BIL(float, 0, 1); }
    else { if (S == 4) TRS_BIL(float, 1, 4); else TRS_BIL(float, 1, 1); }
  } else {
    if (mode == 0) { if (S == 4) TRS_BIL(bf16_t, 0, 4); else TRS_BIL(bf16_t, 0, 1); }
    else { if (S == 4) TRS_BIL(bf16_t, 1, 4); else TRS_BIL(bf16_t, 1, 1); }
  }
#undef TRS_BIL
  return check_launch("pair_bilinear_fwd");
}

extern "C" int trs_pair_bilinear_bwd_data(const void* g, const void* x, const void* W, int32_t w_per_pair, int32_t mode,
                                          int64_t B, int32_t N, int32_t E, int32_t dtype, void* gx, void* gT,
                                          trs_stream_t stream) {
  TRS_PAIRX_COMMON("pair_bilinear_bwd_data");
  if (B == 0 || N < 2) return TRS_OK;
  TRS_REQUIRE(g && x && W && gx, TRS_EINVAL, "pair_bilinear_bwd_data: NULL pointer");
  TRS_REQUIRE(mode == 0 || mode == 1, TRS_EINVAL, "pair_bilinear_bwd_data: mode %d", mode);
  const size_t lds = ((size_t)2 * N * E + 4 * E + (size_t)sched_rounds(N) * sched_width(N)) * 4;
  TRS_REQUIRE(lds <= 64 * 1024, TRS_ESHAPE, "pair_bilinear_bwd_data: N*E = %d exceeds the LDS block", N * E);
  const int wp = w_per_pair ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  const int grid = sample_grid(B);
#define TRS_BILB(T_, MODE_)                                                                                          \
  hipLaunchKernelGGL((pair_bil_bwd_data_kernel<T_, MODE_>), dim3(grid), dim3(256), lds, s, (const T_*)g, (const T_*)x, \
                     (const T_*)W, wp, B, N, E, (T_*)gx, (T_*)gT)
  if (dtype == TRS_F32) { if (mode == 0) TRS_BILB(float, 0); else TRS_BILB(float, 1); }
  else { if (mode == 0) TRS_BILB(bf16_t, 0); else TRS_BILB(bf16_t, 1); }
#undef TRS_BILB
  return check_launch("pair_bilinear_bwd_data");
}

static bool pow2_i(int v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int trs_pair_epilogue_fwd(void* T, const void* x, const void* bias, int32_t bias_per_pair, int32_t mode,
                                     int64_t B, int32_t N, int32_t E, int32_t dtype, void* out, trs_stream_t stream) {
  TRS_PAIRX_COMMON("pair_epilogue_fwd");
  if (B == 0 || N < 2) return TRS_OK;
  TRS_REQUIRE(T && x && (mode == 1 || out), TRS_EINVAL, "pair_epilogue_fwd: NULL pointer");
  TRS_REQUIRE(mode == 0 || mode == 1, TRS_EINVAL, "pair_epilogue_fwd: mode %d", mode);
  const int VE = dtype == TRS_F32 ? 4 : 8;
  TRS_REQUIRE(E % VE == 0 && pow2_i(E / VE) && E / VE <= 64, TRS_ESHAPE,
              "pair_epilogue_fwd: E = %d must be %d times a power of two <= 64", E, VE);
  TRS_REQUIRE(aligned16(T), TRS_EALIGN, "pair_epilogue_fwd: T must be 16-byte aligned");
  const size_t lds = (size_t)N * E * 4 + (size_t)N * (N - 1) / 2 * 4;
  TRS_REQUIRE(lds <= 64 * 1024, TRS_ESHAPE, "pair_epilogue_fwd: N = %d, E = %d exceed the 64 KB LDS block", N, E);
  hipStream_t s = (hipStream_t)stream;
  const int bp = bias_per_pair ? 1 : 0, grid = sample_grid(B);
#define TRS_EPI(T_, M_)                                                                                               \
  hipLaunchKernelGGL((pair_epi_fwd_kernel<T_, M_>), dim3(grid), dim3(256), lds, s, (T_*)T, (const T_*)x, (const T_*)bias, \
                     bp, B, N, E, (T_*)out)
  if (dtype == TRS_F32) { if (mode == 0) TRS_EPI(float, 0); else TRS_EPI(float, 1); }
  else { if (mode == 0) TRS_EPI(bf16_t, 0); else TRS_EPI(bf16_t, 1); }
#undef TRS_EPI
  return check_launch("pair_epilogue_fwd");
}

extern "C" int trs_pair_epilogue_bwd(const void* g, const void* x, void* T, int32_t mode, int64_t B, int32_t N,
                                     int32_t E, int32_t dtype, void* gxj, trs_stream_t stream) {
  TRS_PAIRX_COMMON("pair_epilogue_bwd");
  if (B == 0 || N < 2) return TRS_OK;
  TRS_REQUIRE(g && x && T && gxj, TRS_EINVAL, "pair_epilogue_bwd: NULL pointer");
  TRS_REQUIRE(mode == 0 || mode == 1, TRS_EINVAL, "pair_epilogue_bwd: mode %d", mode);
  const int VE = dtype == TRS_F32 ? 4 : 8;
  TRS_REQUIRE(E % VE == 0 && E / VE <= 256, TRS_ESHAPE, "pair_epilogue_bwd: E = %d must be a multiple of %d", E, VE);
  TRS_REQUIRE(aligned16(T) && (mode == 0 || aligned16(g)), TRS_EALIGN, "pair_epilogue_bwd: 16-byte alignment");
  const size_t lds = (size_t)N * E * 4 * 2 + (size_t)sched_rounds(N) * sched_width(N) * 4;
  TRS_REQUIRE(lds <= 64 * 1024, TRS_ESHAPE, "pair_epilogue_bwd: N = %d, E = %d exceed the 64 KB LDS block", N, E);
  hipStream_t s = (hipStream_t)stream;
  const int grid = sample_grid(B);
#define TRS_EPIB(T_, M_)                                                                                              \
  hipLaunchKernelGGL((pair_epi_bwd_kernel<T_, M_>), dim3(grid), dim3(256), lds, s, (const T_*)g, (const T_*)x, (T_*)T, B, \
                     N, E, (T_*)gxj)
  if (dtype == TRS_F32) { if (mode == 0) TRS_EPIB(float, 0); else TRS_EPIB(float, 1); }
  else { if (mode == 0) TRS_EPIB(bf16_t, 0); else TRS_EPIB(bf16_t, 1); }
#undef TRS_EPIB
  return check_launch("pair_epilogue_bwd");
}
