// K7 inner-product network (IPN) and K3 field-aware FM pair products (FFM).
//
// IPN: per sample the N x E block X is staged in LDS (fp32) and the strict upper triangle of X*X^T
// is computed with 4x4 register tiles: one lane owns a 4x4 tile of field pairs, streams both row
// groups with ds_read_b128 along E and does 64 FMAs per 8 LDS reads.  Results are staged in LDS and
// stored as one contiguous N(N-1)/2 run per sample.  Backward is (G_sym * X) with G_sym the
// symmetric zero-diagonal matrix of the incoming pair gradients, same tiling.
// FFM: pure element-wise, HBM-bound: one lane = one 16-byte vector of an output row.
#include <algorithm>

#include "trs_common.hpp"

namespace trs {

__device__ __forceinline__ int pair_index(int i, int j, int N) {  // i < j, lexicographic
  return i * N - (i * (i + 1)) / 2 + (j - i - 1);
}

// --------------------------------------------------------------------------------------------
// stage one sample's (N x E) block into LDS as fp32, row stride ES floats; rows N..NP-1 zeroed
template <typename T>
__device__ __forceinline__ void stage_block(const T* __restrict__ x, float* __restrict__ X, int N, int NP, int E,
                                            int ES, int lane) {
  constexpr int VE = Vec16<T>::VE;
  if ((E % VE) == 0 && ((((uintptr_t)x) & 15u) == 0)) {
    const int vpr = E / VE;
    const uint4* xv = reinterpret_cast<const uint4*>(x);
    for (int v = lane; v < N * vpr; v += 64) {
      const int n = v / vpr, lv = v - n * vpr;
      float f[VE];
      Vec16<T>::unpack(xv[v], f);
      float4* dst = reinterpret_cast<float4*>(X + n * ES + lv * VE);
#pragma unroll
      for (int k = 0; k < VE; k += 4) dst[k / 4] = make_float4(f[k], f[k + 1], f[k + 2], f[k + 3]);
    }
  } else {
    for (int v = lane; v < N * E; v += 64) {
      const int n = v / E, e = v - n * E;
      X[n * ES + e] = to_f32(x[v]);
    }
  }
  for (int v = lane; v < (NP - N) * E; v += 64) {
    const int n = N + v / E, e = v % E;
    X[n * ES + e] = 0.f;
  }
}

// --------------------------------------------------------------------------------------------
// IPN forward.  Block = 4 waves, one sample per wave per iteration (block-uniform trip count).
template <typename T>
__global__ __launch_bounds__(256) void pair_dot_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t B,
                                                           int N, int E) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int TN = (N + 3) >> 2, NP = TN * 4, ES = E + 4;
  const int P = N * (N - 1) / 2;
  const int ntiles = TN * (TN + 1) / 2;
  const int per_wave = NP * ES + ((P + 3) & ~3);
  // layout: [tile table: ntiles ushort2 (padded to 16 B)] [wave 0: X | out stage] [wave 1] ...
  unsigned short* tile_ij = reinterpret_cast<unsigned short*>(smem);
  float* base = reinterpret_cast<float*>(smem + ((ntiles * 4 + 15) & ~15));
  float* X = base + wave * per_wave;
  float* O = X + NP * ES;
  for (int q = threadIdx.x; q < ntiles; q += blockDim.x) {
    int ti = 0, rem = q;
    while (rem >= TN - ti) { rem -= TN - ti; ++ti; }
    tile_ij[2 * q] = (unsigned short)ti;
    tile_ij[2 * q + 1] = (unsigned short)(ti + rem);
  }
  const int64_t bstride = (int64_t)gridDim.x * 4;
  for (int64_t b0 = (int64_t)blockIdx.x * 4; b0 < B; b0 += bstride) {
    const int64_t b = b0 + wave;
    __syncthreads();  // previous iteration's readers are done with X / O
    if (b < B) stage_block<T>(x + b * N * E, X, N, NP, E, ES, lane);
    __syncthreads();
    if (b < B) {
      for (int q = lane; q < ntiles; q += 64) {
        const int ti = tile_ij[2 * q], tj = tile_ij[2 * q + 1];
        float acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
        const float* xa = X + (4 * ti) * ES;
        const float* xb = X + (4 * tj) * ES;
        for (int e = 0; e < E; e += 4) {
          float4 a[4], bb[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            a[r] = *reinterpret_cast<const float4*>(xa + r * ES + e);
            bb[r] = *reinterpret_cast<const float4*>(xb + r * ES + e);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              acc[r][c] = fmaf(a[r].x, bb[c].x, acc[r][c]);
              acc[r][c] = fmaf(a[r].y, bb[c].y, acc[r][c]);
              acc[r][c] = fmaf(a[r].z, bb[c].z, acc[r][c]);
              acc[r][c] = fmaf(a[r].w, bb[c].w, acc[r][c]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int i = 4 * ti + r, j = 4 * tj + c;
            if (i < j && j < N) O[pair_index(i, j, N)] = acc[r][c];
          }
      }
    }
    __syncthreads();
    if (b < B) {
      T* o = out + b * P;
      for (int k = lane; k < P; k += 64) o[k] = from_f32<T>(O[k]);
    }
  }
}

// IPN backward: dx[i,:] = sum_j Gs[i][j] * x[j,:]; lane owns a (4 fields) x (4 columns) tile of dx.
template <typename T>
__global__ __launch_bounds__(256) void pair_dot_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                           T* __restrict__ dx, int64_t B, int N, int E) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int TN = (N + 3) >> 2, NP = TN * 4, ES = E + 4, GS = NP + 4;
  const int P = N * (N - 1) / 2;
  const int per_wave = NP * ES + NP * GS;
  float* X = reinterpret_cast<float*>(smem) + wave * per_wave;
  float* G = X + NP * ES;  // G[j][i], symmetric, zero diagonal and zero padding
  const int E4 = E >> 2;
  const int ntiles = TN * E4;
  const int64_t bstride = (int64_t)gridDim.x * 4;
  for (int64_t b0 = (int64_t)blockIdx.x * 4; b0 < B; b0 += bstride) {
    const int64_t b = b0 + wave;
    __syncthreads();
    if (b < B) {
      stage_block<T>(x + b * N * E, X, N, NP, E, ES, lane);
      const T* gb = g + b * P;
      for (int v = lane; v < NP * NP; v += 64) {
        const int i = v / NP, j = v - i * NP;
        float val = 0.f;
        if (i != j && i < N && j < N) val = to_f32(gb[i < j ? pair_index(i, j, N) : pair_index(j, i, N)]);
        G[i * GS + j] = val;
      }
    }
    __syncthreads();
    if (b < B) {
      for (int q = lane; q < ntiles; q += 64) {
        const int ti = q / E4, e = (q - ti * E4) * 4;
        float acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
        for (int j = 0; j < N; ++j) {
          const float4 gv = *reinterpret_cast<const float4*>(G + j * GS + 4 * ti);
          const float4 xv = *reinterpret_cast<const float4*>(X + j * ES + e);
          const float gr[4] = {gv.x, gv.y, gv.z, gv.w};
          const float xc[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(gr[r], xc[c], acc[r][c]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 4 * ti + r;
          if (i < N) {
            T* d = dx + (b * N + i) * E + e;
            if (sizeof(T) == 4) {
              *reinterpret_cast<float4*>(d) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
            } else {
              *reinterpret_cast<uint2*>(d) = make_uint2(f32x2_to_bf16x2_bits(acc[r][0], acc[r][1]),
                                                        f32x2_to_bf16x2_bits(acc[r][2], acc[r][3]));
            }
          }
        }
      }
    }
  }
}

// generic fallbacks (E not a multiple of 4, or the block does not fit LDS)
template <typename T>
__global__ __launch_bounds__(256) void pair_dot_fwd_generic(const T* __restrict__ x, T* __restrict__ out, int64_t B,
                                                            int N, int E) {
  const int P = N * (N - 1) / 2;
  const int64_t total = B * P, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / P;
    int rem = (int)(t - b * P), i = 0;
    while (rem >= N - 1 - i) { rem -= N - 1 - i; ++i; }
    const int j = i + 1 + rem;
    const T* xi = x + (b * N + i) * E;
    const T* xj = x + (b * N + j) * E;
    float acc = 0.f;
    for (int e = 0; e < E; ++e) acc = fmaf(to_f32(xi[e]), to_f32(xj[e]), acc);
    out[t] = from_f32<T>(acc);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void pair_dot_bwd_generic(const T* __restrict__ x, const T* __restrict__ g,
                                                            T* __restrict__ dx, int64_t B, int N, int E) {
  const int P = N * (N - 1) / 2;
  const int64_t total = B * N * E, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / ((int64_t)N * E);
    const int rem = (int)(t - b * N * E);
    const int i = rem / E, e = rem - i * E;
    float acc = 0.f;
    for (int j = 0; j < N; ++j) {
      if (j == i) continue;
      const int p = i < j ? pair_index(i, j, N) : pair_index(j, i, N);
      acc = fmaf(to_f32(g[b * P + p]), to_f32(x[(b * N + j) * E + e]), acc);
    }
    dx[t] = from_f32<T>(acc);
  }
}

// --------------------------------------------------------------------------------------------
// IPN on the matrix cores (bf16):  G = X X^T per sample (N x E block), strict upper triangle stored.
// Both MFMA operands are ROWS of X: A[m = field i][k = 8 consecutive e] and B[k = 8 consecutive e][n = field j]
// are the same 16-byte loads straight from the (B,N,E) block -- no LDS, no transposes.  One wave = one sample,
// NT = ceil(N/16) row tiles x KS = E/32 k-steps of fragments in registers (24 VGPRs at N=39, E=64), then
// NT(NT+1)/2 tile pairs x KS v_mfma_f32_16x16x32_bf16.  D[m = i][n = j] lands with j on lanes: each register
// is a run of 16 consecutive pair slots of one i.
typedef __attribute__((ext_vector_type(8))) __bf16 pd_bf16x8;
typedef __attribute__((ext_vector_type(4))) float pd_f32x4;

// GATHER: the rows come straight from the embedding table (K7 fused with K1: row id = idx[b,n] + offsets[n], an
// out-of-range id reads as a zero row and raises err_flag), and -- when ``emb`` is given -- the looked-up block is written
// on the way (the fragments a lane holds are exactly its 16-byte pieces of the rows), so the (B,N,E) block is neither
// written first and read back nor, at inference, written at all.
template <int NT, int KS, bool GATHER = false, typename IdxT = int64_t>
__global__ __launch_bounds__(256) void pair_dot_fwd_mfma_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                                int64_t B, int N, int E,
                                                                const IdxT* __restrict__ idx = nullptr,
                                                                const int64_t* __restrict__ offsets = nullptr,
                                                                int64_t V = 0, bf16_t* __restrict__ emb = nullptr,
                                                                int32_t* __restrict__ err_flag = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  const int P = N * (N - 1) / 2;
  // per-wave staging of one sample's pair slots: the D tiles scatter 2-byte results (issue-bound as direct
  // global stores: 70 % of wave cycles in issue stalls); from LDS the row leaves as P/64 coalesced stores
  unsigned short* O = reinterpret_cast<unsigned short*>(smem) + (size_t)wave * ((P + 7) & ~7);
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = wid; b < B; b += nwaves) {
    uint4 F[NT][KS];
    int64_t rid[NT];
    if (GATHER) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int row = 16 * t + r;
        rid[t] = -1;
        if (row < N) {
          rid[t] = load_row_id(idx, offsets, b * N + row, row);
          if (rid[t] < 0 || rid[t] >= V) {
            if (err_flag != nullptr) *err_flag = 1;
            rid[t] = -1;
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int row = 16 * t + r;
        F[t][ks] = make_uint4(0, 0, 0, 0);
        if (GATHER) {
          if (rid[t] >= 0) F[t][ks] = *reinterpret_cast<const uint4*>(x + rid[t] * (int64_t)E + 32 * ks + 8 * q);
        } else if (row < N) {
          F[t][ks] = *reinterpret_cast<const uint4*>(x + (b * N + row) * (int64_t)E + 32 * ks + 8 * q);
        }
      }
    if (GATHER && emb != nullptr) {
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int row = 16 * t + r;
          if (row < N) {
            const u32x4 w = {F[t][ks].x, F[t][ks].y, F[t][ks].z, F[t][ks].w};
            __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(emb + (b * N + row) * (int64_t)E + 32 * ks + 8 * q));
          }
        }
    }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = ti; tj < NT; ++tj) {
        pd_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pd_bf16x8, F[ti][ks]),
                                                        __builtin_bit_cast(pd_bf16x8, F[tj][ks]), acc, 0, 0, 0);
        const int j = 16 * tj + r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = 16 * ti + 4 * q + k;
          if (i < j && j < N) O[pair_index(i, j, N)] = from_f32<bf16_t>(acc[k]).v;
        }
      }
    __builtin_amdgcn_wave_barrier();
    unsigned short* o = reinterpret_cast<unsigned short*>(out + b * P);
    for (int k = lane; k < P; k += 64) o[k] = O[k];
    __builtin_amdgcn_wave_barrier();
  }
}

// backward on the matrix cores: dX = Gs X, Gs = symmetric zero-diagonal matrix of the pair gradients, computed
// TRANSPOSED so that both operands are 16-byte LDS reads and the result leaves as 16-byte stores:
//   D^T[m = e][n = i] = sum_j A[m = e][k = j] B[k = j][n = i]
// A = X^T (an LDS copy of the sample written transposed, [e][j], 2-byte stores; rows fed in a permuted order so
// that a lane's outputs for field i are runs of 8 consecutive e), B = Gs (symmetric: row i, 8 consecutive j).
// The first version (D = Gs X) needed 8 two-byte LDS reads per B fragment and stored dx as scattered 2-byte
// global stores (48 per lane and sample): it ran at 34 % of the HBM roofline.
template <int NT, int KE /* E/16 row tiles of D^T */>
__global__ __launch_bounds__(256) void pair_dot_bwd_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ g,
                                                                bf16_t* __restrict__ dx, int64_t B, int N, int E) {
  constexpr int NP = 16 * NT;            // padded field count
  constexpr int KJ = (NP + 31) / 32;     // k-steps over j
  constexpr int NPK = 32 * KJ;
  constexpr int GS = NPK + 8;            // Gs row stride (bf16 elements): 16-byte aligned, conflict-spreading pad
  constexpr int EC = 16 * KE;
  constexpr int XTS = NPK + 8;           // X^T row stride
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  const int P = N * (N - 1) / 2;
  const int PL = (P + 7) & ~7;
  unsigned short* lut = reinterpret_cast<unsigned short*>(smem);   // pair -> (i << 8 | j), shared by the block
  unsigned short* Gs = lut + PL + (size_t)wave * (NP * GS + EC * XTS);
  unsigned short* XT = Gs + NP * GS;
  for (int i = threadIdx.x; i < N - 1; i += blockDim.x) {
    const int base = pair_index(i, i + 1, N);
    for (int j = i + 1; j < N; ++j) lut[base + (j - i - 1)] = (unsigned short)((i << 8) | j);
  }
  // Gs / XT belong to this wave alone.  Zero them once: the diagonal of Gs, its padding and the X^T columns >= N
  // are never written afterwards, every other entry is overwritten for each sample.
  for (int v = lane; v < (NP * GS) / 8; v += 64) reinterpret_cast<uint4*>(Gs)[v] = make_uint4(0, 0, 0, 0);
  for (int v = lane; v < (EC * XTS) / 8; v += 64) reinterpret_cast<uint4*>(XT)[v] = make_uint4(0, 0, 0, 0);
  __syncthreads();   // lut complete (block-wide), zero fill visible
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // A wave works through its samples one at a time, and a sample is a chain of dependent latencies (its gradient row
  // and its x block from HBM, the LDS images, the fragments): the raw data of the NEXT sample is requested while this
  // one is processed -- hand-written loads (the compiler moves ordinary ones to their first use), waited for after
  // the MFMAs and before the stores of this sample, so that the wait covers the prefetch only (236 -> 164 us at the
  // BASELINE shape).  Every lane issues every load (addresses clamped): a branch around an asm load makes the compiler
  // copy its destination register while the load is still in flight.
  constexpr int GR = (NP * (NP - 1) / 2 + 63) / 64;      // gradient values per lane (upper bound from the padded N)
  constexpr int XR = (NP * (EC / 8) + 63) / 64;          // 16-byte x vectors per lane
  typedef __attribute__((ext_vector_type(4))) unsigned pd_u32x4;
  unsigned gcur[GR], gnxt[GR];
  pd_u32x4 xcur[XR], xnxt[XR];
  const int nxv = N * (EC / 8);
#define TRS_PD_FETCH(gd, xd, bb)                                                                        \
  _Pragma("unroll") for (int u = 0; u < GR; ++u) {                                                      \
    const int pp = lane + 64 * u;                                                                       \
    const bf16_t* a_ = g + (bb) * P + (pp < P ? pp : P - 1);                                            \
    asm volatile("global_load_ushort %0, %1, off" : "=v"(gd[u]) : "v"(a_));                             \
  }                                                                                                     \
  _Pragma("unroll") for (int u = 0; u < XR; ++u) {                                                      \
    const int v = lane + 64 * u;                                                                        \
    const int vv = v < nxv ? v : nxv - 1;                                                               \
    const bf16_t* a_ = x + ((bb) * N) * (int64_t)E + 8 * vv;       /* the sample's block is contiguous */ \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xd[u]) : "v"(a_));                            \
  }
#define TRS_PD_COMMIT(gd, xd)                                                     \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                \
  _Pragma("unroll") for (int u = 0; u < GR; ++u) asm volatile("" : "+v"(gd[u]));  \
  _Pragma("unroll") for (int u = 0; u < XR; ++u) asm volatile("" : "+v"(xd[u]));
  if (wid < B) {
    TRS_PD_FETCH(gcur, xcur, wid)
    TRS_PD_COMMIT(gcur, xcur)
  }
  for (int64_t b = wid; b < B; b += nwaves) {
    const int64_t bn = b + nwaves < B ? b + nwaves : b;
    TRS_PD_FETCH(gnxt, xnxt, bn)
    __builtin_amdgcn_sched_barrier(0);
    {
#pragma unroll
      for (int u = 0; u < GR; ++u) {
        const int pp = lane + 64 * u;
        if (pp < P) {
          const unsigned ij = lut[pp];
          const int i = ij >> 8, j = ij & 255;
          Gs[i * GS + j] = (unsigned short)gcur[u];
          Gs[j * GS + i] = (unsigned short)gcur[u];
        }
      }
#pragma unroll
      for (int u = 0; u < XR; ++u) {
        const int v = lane + 64 * u;
        if (v < nxv) {
          const int row = v / (EC / 8), c8 = v - row * (EC / 8);
          const unsigned w[4] = {xcur[u][0], xcur[u][1], xcur[u][2], xcur[u][3]};
#pragma unroll
          for (int k = 0; k < 8; ++k) XT[(8 * c8 + k) * XTS + row] = (unsigned short)(w[k >> 1] >> (16 * (k & 1)));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    uint4 res[KE / 2][NT];             // dx of the sample, held until the prefetch has been waited for
    {
      uint4 Bg[NT][KJ];                  // Gs fragments: field i = 16 ti + r, j run 32 kj + 8 q
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int kj = 0; kj < KJ; ++kj)
          Bg[ti][kj] = *reinterpret_cast<const uint4*>(Gs + (16 * ti + r) * GS + 32 * kj + 8 * q);
#pragma unroll
      for (int u = 0; u < KE / 2; ++u) {
        uint4 Ax[2][KJ];                 // X^T fragments of the tile pair u: row e = 32 u + 8 (r>>2) + 4 h + (r&3)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int kj = 0; kj < KJ; ++kj)
            Ax[h][kj] = *reinterpret_cast<const uint4*>(XT + (32 * u + 8 * (r >> 2) + 4 * h + (r & 3)) * XTS + 32 * kj + 8 * q);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
          pd_f32x4 acc[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            acc[h] = pd_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kj = 0; kj < KJ; ++kj)
              acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pd_bf16x8, Ax[h][kj]),
                                                               __builtin_bit_cast(pd_bf16x8, Bg[ti][kj]), acc[h], 0, 0, 0);
          }
          const float run[8] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
          res[u][ti] = Vec16<bf16_t>::pack(run);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    TRS_PD_COMMIT(gnxt, xnxt)
#pragma unroll
    for (int u = 0; u < GR; ++u) gcur[u] = gnxt[u];
#pragma unroll
    for (int u = 0; u < XR; ++u) xcur[u] = xnxt[u];
#pragma unroll
    for (int u = 0; u < KE / 2; ++u)
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        const int i = 16 * ti + r;
        if (i < N) *reinterpret_cast<uint4*>(dx + (b * N + i) * (int64_t)E + 32 * u + 8 * q) = res[u][ti];
      }
    __builtin_amdgcn_wave_barrier();   // the next sample overwrites Gs / XT
  }
#undef TRS_PD_FETCH
#undef TRS_PD_COMMIT
}

static bool pair_mfma_ok(int N, int E) { return N >= 2 && N <= 64 && (E == 32 || E == 64 || E == 128); }

static int pair_dot_fwd_mfma(const void* x, void* out, int64_t B, int N, int E, hipStream_t s) {
  const int NT = (N + 15) / 16, KS = E / 32;
  const int grid = (int)std::min<int64_t>((B + 3) / 4, 256 * 8);
  const size_t lds_f = (size_t)4 * (((N * (N - 1) / 2) + 7) & ~7) * 2;
#define TRS_PF(NT_, KS_)                                                                                      \
  hipLaunchKernelGGL((pair_dot_fwd_mfma_kernel<NT_, KS_>), dim3(grid), dim3(256), lds_f, s, (const bf16_t*)x,  \
                     (bf16_t*)out, B, N, E)
#define TRS_PF_K(NT_)               \
  do {                              \
    if (KS == 1) TRS_PF(NT_, 1);    \
    else if (KS == 2) TRS_PF(NT_, 2); \
    else TRS_PF(NT_, 4);            \
  } while (0)
  switch (NT) {
    case 1: TRS_PF_K(1); break;
    case 2: TRS_PF_K(2); break;
    case 3: TRS_PF_K(3); break;
    default: TRS_PF_K(4); break;
  }
#undef TRS_PF_K
#undef TRS_PF
  return check_launch("pair_dot_fwd(mfma)");
}

template <typename IdxT>
static int embed_pair_dot_mfma(const void* table, const IdxT* idx, const int64_t* offsets, int64_t V, void* emb, void* out,
                               int32_t* err_flag, int64_t B, int N, int E, hipStream_t s) {
  const int NT = (N + 15) / 16, KS = E / 32;
  const int grid = (int)std::min<int64_t>((B + 3) / 4, 256 * 8);
  const size_t lds_f = (size_t)4 * (((N * (N - 1) / 2) + 7) & ~7) * 2;
#define TRS_PG(NT_, KS_)                                                                                           \
  hipLaunchKernelGGL((pair_dot_fwd_mfma_kernel<NT_, KS_, true, IdxT>), dim3(grid), dim3(256), lds_f, s,             \
                     (const bf16_t*)table, (bf16_t*)out, B, N, E, idx, offsets, V, (bf16_t*)emb, err_flag)
#define TRS_PG_K(NT_)               \
  do {                              \
    if (KS == 1) TRS_PG(NT_, 1);    \
    else if (KS == 2) TRS_PG(NT_, 2); \
    else TRS_PG(NT_, 4);            \
  } while (0)
  switch (NT) {
    case 1: TRS_PG_K(1); break;
    case 2: TRS_PG_K(2); break;
    case 3: TRS_PG_K(3); break;
    default: TRS_PG_K(4); break;
  }
#undef TRS_PG_K
#undef TRS_PG
  return check_launch("embed_pair_dot");
}

static int pair_dot_bwd_mfma(const void* x, const void* g, void* dx, int64_t B, int N, int E, hipStream_t s) {
  const int NT = (N + 15) / 16, KE = E / 16;
  const int NP = 16 * NT, NPK = 32 * ((NP + 31) / 32);
  const int P = N * (N - 1) / 2;
  const size_t lds = (size_t)((P + 7) & ~7) * 2 + (size_t)4 * (NP * (NPK + 8) + E * (NPK + 8)) * 2;
  const int grid = (int)std::min<int64_t>((B + 3) / 4, 256 * 8);
#define TRS_PB(NT_, KE_)                                                                                      \
  do {                                                                                                        \
    auto kern = pair_dot_bwd_mfma_kernel<NT_, KE_>;                                                           \
    static bool attr_set = false;                                                                             \
    if (!attr_set && lds > 64 * 1024) {                                                                       \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=     \
          hipSuccess)                                                                                         \
        return check_launch("pair_dot_bwd(mfma): LDS attribute");                                             \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)g, (bf16_t*)dx,  \
                       B, N, E);                                                                              \
  } while (0)
#define TRS_PB_K(NT_)               \
  do {                              \
    if (KE == 2) TRS_PB(NT_, 2);    \
    else if (KE == 4) TRS_PB(NT_, 4); \
    else TRS_PB(NT_, 8);            \
  } while (0)
  switch (NT) {
    case 1: TRS_PB_K(1); break;
    case 2: TRS_PB_K(2); break;
    case 3: TRS_PB_K(3); break;
    default: TRS_PB_K(4); break;
  }
#undef TRS_PB_K
#undef TRS_PB
  return check_launch("pair_dot_bwd(mfma)");
}

constexpr size_t LDS_BUDGET = 80 * 1024;  // per block: 2 blocks per CU out of 160 KiB

template <typename T>
static int pair_dot_fwd_launch(const void* x, void* out, int64_t B, int N, int E, hipStream_t s) {
  const int TN = (N + 3) / 4, NP = TN * 4, ES = E + 4, P = N * (N - 1) / 2;
  const size_t lds = (size_t)((TN * (TN + 1) / 2 * 4 + 15) & ~15) + 4 * (size_t)(NP * ES + ((P + 3) & ~3)) * 4;
  if (E % 4 == 0 && lds <= LDS_BUDGET && N >= 2) {
    const int grid = (int)std::min<int64_t>((B + 3) / 4, 256 * 8);
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)pair_dot_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)LDS_BUDGET) != hipSuccess)
        return check_launch("pair_dot_fwd: LDS attribute");
      attr_set = true;
    }
    hipLaunchKernelGGL((pair_dot_fwd_kernel<T>), dim3(grid), dim3(256), lds, s, (const T*)x, (T*)out, B, N, E);
  } else {
    hipLaunchKernelGGL((pair_dot_fwd_generic<T>), dim3(stream_grid(B * P, 256, 8192)), dim3(256), 0, s, (const T*)x,
                       (T*)out, B, N, E);
  }
  return check_launch("pair_dot_fwd");
}
template <typename T>
static int pair_dot_bwd_launch(const void* x, const void* g, void* dx, int64_t B, int N, int E, hipStream_t s) {
  const int TN = (N + 3) / 4, NP = TN * 4, ES = E + 4, GS = NP + 4;
  const size_t lds = 4 * (size_t)(NP * ES + NP * GS) * 4;
  if (E % 4 == 0 && lds <= LDS_BUDGET && N >= 2) {
    const int grid = (int)std::min<int64_t>((B + 3) / 4, 256 * 8);
    static bool attr_set = false;   // more than 64 KiB of dynamic LDS needs the attribute
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)pair_dot_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)LDS_BUDGET) != hipSuccess)
        return check_launch("pair_dot_bwd: LDS attribute");
      attr_set = true;
    }
    hipLaunchKernelGGL((pair_dot_bwd_kernel<T>), dim3(grid), dim3(256), lds, s, (const T*)x, (const T*)g, (T*)dx, B,
                       N, E);
  } else {
    hipLaunchKernelGGL((pair_dot_bwd_generic<T>), dim3(stream_grid(B * N * E, 256, 8192)), dim3(256), 0, s,
                       (const T*)x, (const T*)g, (T*)dx, B, N, E);
  }
  return check_launch("pair_dot_bwd");
}

// --------------------------------------------------------------------------------------------
// FFM on a materialised (B, N*N, E) block.  UNIT = uint4 (16-byte vectors) or T (elements).
template <typename T, typename UNIT>
__device__ __forceinline__ UNIT mul_unit(const UNIT& a, const UNIT& b);
template <>
__device__ __forceinline__ uint4 mul_unit<float, uint4>(const uint4& a, const uint4& b) {
  float x[4], y[4];
  Vec16<float>::unpack(a, x);
  Vec16<float>::unpack(b, y);
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] *= y[k];
  return Vec16<float>::pack(x);
}
template <>
__device__ __forceinline__ uint4 mul_unit<bf16_t, uint4>(const uint4& a, const uint4& b) {
  float x[8], y[8];
  Vec16<bf16_t>::unpack(a, x);
  Vec16<bf16_t>::unpack(b, y);
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] *= y[k];
  return Vec16<bf16_t>::pack(x);
}
template <>
__device__ __forceinline__ float mul_unit<float, float>(const float& a, const float& b) { return a * b; }
template <>
__device__ __forceinline__ bf16_t mul_unit<bf16_t, bf16_t>(const bf16_t& a, const bf16_t& b) {
  return from_f32<bf16_t>(to_f32(a) * to_f32(b));
}
template <typename UNIT>
__device__ __forceinline__ UNIT zero_unit();
template <>
__device__ __forceinline__ uint4 zero_unit<uint4>() { return make_uint4(0, 0, 0, 0); }
template <>
__device__ __forceinline__ float zero_unit<float>() { return 0.f; }
template <>
__device__ __forceinline__ bf16_t zero_unit<bf16_t>() { return bf16_t{0}; }

// forward: items walk the full N x N square per sample, only i < j produce output
template <typename T, typename UNIT>
__global__ __launch_bounds__(256) void ffm_fwd_kernel(const UNIT* __restrict__ x, UNIT* __restrict__ out, int64_t B,
                                                      int N, int upr /* units per row */) {
  const int P = N * (N - 1) / 2;
  const int per_b = N * N * upr;
  const int64_t total = B * per_b, stride = (int64_t)gridDim.x * blockDim.x;
  const bool f32 = total < ((int64_t)1 << 32);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = udiv_fast(t, per_b, f32);
    unsigned rem = (unsigned)(t - b * per_b);
    const int i = (int)(rem / (unsigned)(N * upr));
    rem -= i * N * upr;
    const int j = (int)(rem / (unsigned)upr), lv = (int)rem - j * upr;
    if (i < j) {
      const UNIT a = x[(b * N * N + (int64_t)i * N + j) * upr + lv];
      const UNIT c = x[(b * N * N + (int64_t)j * N + i) * upr + lv];
      out[(b * P + pair_index(i, j, N)) * upr + lv] = mul_unit<T, UNIT>(a, c);
    }
  }
}
// backward: dx[b,i,j] = g[b,pair(i,j)] * x[b,j,i] (i != j), 0 on the diagonal
template <typename T, typename UNIT>
__global__ __launch_bounds__(256) void ffm_bwd_kernel(const UNIT* __restrict__ x, const UNIT* __restrict__ g,
                                                      UNIT* __restrict__ dx, int64_t B, int N, int upr) {
  const int P = N * (N - 1) / 2;
  const int per_b = N * N * upr;
  const int64_t total = B * per_b, stride = (int64_t)gridDim.x * blockDim.x;
  const bool f32 = total < ((int64_t)1 << 32);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = udiv_fast(t, per_b, f32);
    unsigned rem = (unsigned)(t - b * per_b);
    const int i = (int)(rem / (unsigned)(N * upr));
    rem -= i * N * upr;
    const int j = (int)(rem / (unsigned)upr), lv = (int)rem - j * upr;
    UNIT r = zero_unit<UNIT>();
    if (i != j) {
      const int p = i < j ? pair_index(i, j, N) : pair_index(j, i, N);
      r = mul_unit<T, UNIT>(g[(b * P + p) * upr + lv], x[(b * N * N + (int64_t)j * N + i) * upr + lv]);
    }
    dx[t] = r;
  }
}
// fused: gather straight from the N tables
template <typename T, typename UNIT, typename IdxT>
__global__ __launch_bounds__(256) void ffm_fused_fwd_kernel(const UNIT* const* __restrict__ tables,
                                                            const IdxT* __restrict__ idx,
                                                            const int64_t* __restrict__ offsets,
                                                            UNIT* __restrict__ out, int64_t B, int N, int upr,
                                                            int64_t V, int32_t* __restrict__ err_flag) {
  const int P = N * (N - 1) / 2;
  const int64_t per_b = (int64_t)N * N * upr;
  const int64_t total = B * per_b, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / per_b;
    int rem = (int)(t - b * per_b);
    const int i = rem / (N * upr);
    rem -= i * N * upr;
    const int j = rem / upr, lv = rem - j * upr;
    if (i < j) {
      const int64_t ri = load_row_id(idx, offsets, b * N + i, i);
      const int64_t rj = load_row_id(idx, offsets, b * N + j, j);
      UNIT r = zero_unit<UNIT>();
      if (err_flag != nullptr && (ri < 0 || ri >= V || rj < 0 || rj >= V)) {
        *err_flag = 1;
      } else {
        r = mul_unit<T, UNIT>(tables[i][rj * upr + lv], tables[j][ri * upr + lv]);
      }
      out[(b * P + pair_index(i, j, N)) * upr + lv] = r;
    }
  }
}

// The same, round 5 (rows of whole 16-byte units): a workgroup takes a SAMPLE at a time -- its N row ids are read once into
// LDS (the element kernel above re-reads two of them per 16 bytes of output and divides a 64-bit item index three times)
// -- and its lane groups (UPR lanes = one row) walk the P pairs, PIPE of them in flight per group: 2 PIPE independent row
// loads, then PIPE products and streaming stores.  Consecutive groups take consecutive pairs: a wave writes whole
// contiguous KBs of the output.  Rows come from N tables of V rows each (5 GB at the BASELINE shape: nothing is reused
// before it is evicted), so they are fetched with streaming loads when the tables are larger than the caches, and the
// 6.2 GB product is written past them.  The (i, j) of every pair comes from a table in LDS built once per workgroup.
template <typename T, typename IdxT, int LOG2U, bool STREAM>
__global__ __launch_bounds__(256) void ffm_fused_fwd_rows_kernel(const uint4* const* __restrict__ tables,
                                                                 const IdxT* __restrict__ idx,
                                                                 const int64_t* __restrict__ offsets,
                                                                 uint4* __restrict__ out, int64_t B, int N, int64_t V,
                                                                 int32_t* __restrict__ err_flag) {
  constexpr int U = 1 << LOG2U, G = 256 >> LOG2U, PIPE = 4;
  extern __shared__ __attribute__((aligned(16))) char ffm_lds[];
  const int P = N * (N - 1) / 2;
  unsigned short* pij = reinterpret_cast<unsigned short*>(ffm_lds);                       // [P]: i << 8 | j
  int64_t* rid = reinterpret_cast<int64_t*>(ffm_lds + ((2 * P + 15) / 16) * 16);          // [N] row ids of the sample
  for (int i = threadIdx.x; i < N; i += 256)
    for (int j = i + 1; j < N; ++j) pij[pair_index(i, j, N)] = (unsigned short)(i << 8 | j);
  const int lv = threadIdx.x & (U - 1), grp = threadIdx.x >> LOG2U;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();                                      // the previous sample's ids are no longer read
    for (int n = threadIdx.x; n < N; n += 256) {
      int64_t r = load_row_id(idx, offsets, b * N + n, n);
      if (r < 0 || r >= V) {
        if (err_flag != nullptr) *err_flag = 1;
        r = -1;
      }
      rid[n] = r;
    }
    __syncthreads();
    uint4* ob = out + b * P * U;
    for (int p0 = grp; p0 < P; p0 += G * PIPE) {
      uint4 a[PIPE], c[PIPE];
#pragma unroll
      for (int k = 0; k < PIPE; ++k) {
        const int p = p0 + k * G;
        a[k] = make_uint4(0, 0, 0, 0);
        c[k] = make_uint4(0, 0, 0, 0);
        if (p < P) {
          const int i = pij[p] >> 8, j = pij[p] & 255;
          const int64_t ri = rid[i], rj = rid[j];
          if (ri >= 0 && rj >= 0) {
            a[k] = STREAM ? load_stream(&tables[i][rj * U + lv]) : tables[i][rj * U + lv];
            c[k] = STREAM ? load_stream(&tables[j][ri * U + lv]) : tables[j][ri * U + lv];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < PIPE; ++k) {
        const int p = p0 + k * G;
        if (p < P) store_stream(&ob[p * U + lv], mul_unit<T, uint4>(a[k], c[k]));
      }
    }
  }
}

// fused FFM backward: for table i and row r,
//   grad_i[r,:] = sum over lookups p=(b,j) of row r with j != i of  gout[b, pair(i,j), :] * tables[j][g(b,i), :]
// one UNIT-lane group per (table, row) walks the row's bucket of the shared CSR; every gradient row is
// written exactly once (zeros for rows nobody looked up) -- dense gradients like nn.Embedding's default.
template <typename T, typename UNIT, typename IdxT>
__global__ __launch_bounds__(256) void ffm_fused_bwd_kernel(const UNIT* const* __restrict__ tables,
                                                            const IdxT* __restrict__ idx,
                                                            const int64_t* __restrict__ offsets,
                                                            const UNIT* __restrict__ gout,
                                                            const int32_t* __restrict__ row_start,
                                                            const int32_t* __restrict__ perm, int N, int upr, int64_t V,
                                                            UNIT* const* __restrict__ grads) {
  const int P = N * (N - 1) / 2;
  const int64_t per_table = V * upr;
  const int64_t total = (int64_t)N * per_table, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int i = (int)(t / per_table);
    const int64_t rem = t - (int64_t)i * per_table;
    const int64_t r = rem / upr;
    const int lv = (int)(rem - r * upr);
    const int beg = row_start[r], end = row_start[r + 1];
    float acc[sizeof(UNIT) / sizeof(T)];
#pragma unroll
    for (int k = 0; k < (int)(sizeof(UNIT) / sizeof(T)); ++k) acc[k] = 0.f;
    for (int q = beg; q < end; ++q) {
      const int p = perm[q];
      const int64_t b = p / N;
      const int j = (int)(p - b * N);
      if (j == i) continue;
      const int pidx = i < j ? pair_index(i, j, N) : pair_index(j, i, N);
      const int64_t ri = load_row_id(idx, offsets, b * N + i, i);
      const UNIT gv = gout[(b * P + pidx) * upr + lv];
      const UNIT tv = tables[j][ri * upr + lv];
      if constexpr (sizeof(UNIT) == 16) {
        float gf[Vec16<T>::VE], tf[Vec16<T>::VE];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(&gv), gf);
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(&tv), tf);
#pragma unroll
        for (int k = 0; k < Vec16<T>::VE; ++k) acc[k] = fmaf(gf[k], tf[k], acc[k]);
      } else {
        acc[0] = fmaf(to_f32(*reinterpret_cast<const T*>(&gv)), to_f32(*reinterpret_cast<const T*>(&tv)), acc[0]);
      }
    }
    if constexpr (sizeof(UNIT) == 16) {
      const uint4 o = Vec16<T>::pack(acc);
      grads[i][rem] = *reinterpret_cast<const UNIT*>(&o);
    } else {
      const T o = from_f32<T>(acc[0]);
      grads[i][rem] = *reinterpret_cast<const UNIT*>(&o);
    }
  }
}

// The same walk for rows of whole 16-byte units, round 5: blockIdx.y = table, one UPR-lane group per row, CH bucket
// entries in flight (their positions, then their row ids, then 2 CH independent row loads: the element kernel above
// serialises three dependent loads per entry), 32-bit arithmetic with a reciprocal for p / N, the next row's bucket bounds
// fetched beside the current walk; the output gradient (6.2 GB, each row read by two tables' walks) and -- when the
// tables are larger than the caches -- the table rows come in by streaming loads, the gradient rows leave by streaming
// stores.
template <typename T, typename IdxT, int LOG2U, bool STREAM>
__global__ __launch_bounds__(256) void ffm_fused_bwd_rows_kernel(const uint4* const* __restrict__ tables,
                                                                 const IdxT* __restrict__ idx,
                                                                 const int64_t* __restrict__ offsets,
                                                                 const uint4* __restrict__ gout,
                                                                 const int32_t* __restrict__ row_start,
                                                                 const int32_t* __restrict__ perm, unsigned N,
                                                                 unsigned rcpN, int64_t V, uint4* const* __restrict__ grads,
                                                                 const int32_t* __restrict__ rid_t, int64_t B) {
  constexpr int U = 1 << LOG2U, VE = Vec16<T>::VE, CH = 4;
  const int i = blockIdx.y;
  const int P = (int)(N * (N - 1) / 2);
  const int lv = threadIdx.x & (U - 1);
  const int64_t groups = ((int64_t)gridDim.x * 256) >> LOG2U;
  int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> LOG2U;
  const int64_t off_i = offsets != nullptr ? offsets[i] : 0;
  uint4* gi = grads[i];
  int nbeg = 0, nend = 0;
  if (r < V) { nbeg = row_start[r]; nend = row_start[r + 1]; }
  for (; r < V; r += groups) {
    const int beg = nbeg, end = nend;
    if (r + groups < V) { nbeg = row_start[r + groups]; nend = row_start[r + groups + 1]; }
    float acc[VE];
#pragma unroll
    for (int k = 0; k < VE; ++k) acc[k] = 0.f;
    for (int q = beg; q < end; q += CH) {
      int p[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) p[c] = q + c < end ? perm[q + c] : -1;
      int64_t ri[CH];
      unsigned bb[CH], jj[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        ri[c] = -1;
        bb[c] = jj[c] = 0;
        if (p[c] >= 0) {
          bb[c] = __umulhi((unsigned)p[c], rcpN);      // p / N for p < 2^31, N < 2^16 (host-checked)
          jj[c] = (unsigned)p[c] - bb[c] * N;
          if (jj[c] != (unsigned)i) {
            const int64_t v = rid_t != nullptr ? (int64_t)rid_t[(int64_t)i * B + bb[c]]
                                               : (int64_t)idx[(int64_t)bb[c] * N + i] + off_i;
            ri[c] = (v >= 0 && v < V) ? v : -1;
          }
        }
      }
      uint4 gv[CH], tv[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        gv[c] = make_uint4(0, 0, 0, 0);
        tv[c] = make_uint4(0, 0, 0, 0);
        if (ri[c] >= 0) {
          const int j = (int)jj[c];
          const int pidx = i < j ? pair_index(i, j, (int)N) : pair_index(j, i, (int)N);
          gv[c] = load_stream(&gout[((int64_t)bb[c] * P + pidx) * U + lv]);
          tv[c] = STREAM ? load_stream(&tables[j][ri[c] * U + lv]) : tables[j][ri[c] * U + lv];
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        float gf[VE], tf[VE];
        Vec16<T>::unpack(gv[c], gf);
        Vec16<T>::unpack(tv[c], tf);
#pragma unroll
        for (int k = 0; k < VE; ++k) acc[k] = fmaf(gf[k], tf[k], acc[k]);
      }
    }
    store_stream(&gi[r * U + lv], Vec16<T>::pack(acc));
  }
}

}  // namespace trs

using namespace trs;

#define TRS_CHECK_BNE(name)                                                                                  \
  TRS_REQUIRE(E > 0 && B >= 0 && N > 0, TRS_EINVAL, name ": bad size B=%lld N=%d E=%d", (long long)B, N, E); \
  TRS_REQUIRE(dtype == TRS_F32 || dtype == TRS_BF16, TRS_EDTYPE, name ": dtype %d", dtype)

extern "C" int trs_pair_dot_fwd(const void* x, int64_t B, int32_t N, int32_t E, int32_t dtype, void* out,
                                trs_stream_t stream) {
  if (B == 0 || N < 2) return TRS_OK;  // empty batch / no pairs: nothing to do (pointers may be NULL)
  TRS_REQUIRE(x && out, TRS_EINVAL, "pair_dot_fwd: NULL pointer");
  TRS_CHECK_BNE("pair_dot_fwd");
  if (B == 0 || N < 2) return TRS_OK;
  if (dtype == TRS_F32) return pair_dot_fwd_launch<float>(x, out, B, N, E, (hipStream_t)stream);
  if (pair_mfma_ok(N, E) && aligned16(x)) return pair_dot_fwd_mfma(x, out, B, N, E, (hipStream_t)stream);
  return pair_dot_fwd_launch<bf16_t>(x, out, B, N, E, (hipStream_t)stream);
}

/* see include/trs_abi.h: the inner-product network straight from the embedding table (K7 fused with K1) */
extern "C" int trs_embed_pair_dot(const void* table, int64_t V, int32_t E, int32_t dtype, const void* idx,
                                  int32_t idx_dtype, const int64_t* offsets, int64_t B, int32_t N, void* emb, void* out,
                                  int32_t* err_flag, trs_stream_t stream) {
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(table && idx && (out || N < 2), TRS_EINVAL, "embed_pair_dot: NULL pointer");
  TRS_REQUIRE(V > 0 && E > 0 && N > 0, TRS_EINVAL, "embed_pair_dot: bad size");
  TRS_REQUIRE(idx_dtype == TRS_I64 || idx_dtype == TRS_I32, TRS_EDTYPE, "embed_pair_dot: idx dtype %d", idx_dtype);
  if (dtype != TRS_BF16 || !pair_mfma_ok(N, E) || N < 2 || !aligned16(table) || !aligned16(emb))
    return fail(TRS_ESHAPE, "embed_pair_dot: needs bf16 rows the matrix-core path covers (use gather_rows + pair_dot_fwd)");
  if (idx_dtype == TRS_I64)
    return embed_pair_dot_mfma<int64_t>(table, (const int64_t*)idx, offsets, V, emb, out, err_flag, B, N, E,
                                        (hipStream_t)stream);
  return embed_pair_dot_mfma<int32_t>(table, (const int32_t*)idx, offsets, V, emb, out, err_flag, B, N, E,
                                      (hipStream_t)stream);
}

extern "C" int trs_pair_dot_bwd(const void* x, const void* g, int64_t B, int32_t N, int32_t E, int32_t dtype,
                                void* dx, trs_stream_t stream) {
  if (B == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  TRS_REQUIRE(x && dx && (g || N < 2), TRS_EINVAL, "pair_dot_bwd: NULL pointer");
  TRS_CHECK_BNE("pair_dot_bwd");
  if (N < 2) {
    return zero_bytes(dx, (size_t)B * N * E * dtype_size(dtype), (hipStream_t)stream);
  }
  if (dtype == TRS_F32) return pair_dot_bwd_launch<float>(x, g, dx, B, N, E, (hipStream_t)stream);
  if (pair_mfma_ok(N, E) && aligned16(x)) return pair_dot_bwd_mfma(x, g, dx, B, N, E, (hipStream_t)stream);
  return pair_dot_bwd_launch<bf16_t>(x, g, dx, B, N, E, (hipStream_t)stream);
}

extern "C" int trs_ffm_fwd(const void* x, int64_t B, int32_t N, int32_t E, int32_t dtype, void* out,
                           trs_stream_t stream) {
  if (B == 0 || N < 2) return TRS_OK;  // empty batch / no pairs: nothing to do (pointers may be NULL)
  TRS_REQUIRE(x && out, TRS_EINVAL, "ffm_fwd: NULL pointer");
  TRS_CHECK_BNE("ffm_fwd");
  if (B == 0 || N < 2) return TRS_OK;
  hipStream_t s = (hipStream_t)stream;
  const int rb = E * dtype_size(dtype);
  const bool vec = rb % 16 == 0 && aligned16(x) && aligned16(out);
  const int upr = vec ? rb / 16 : E;
  const int grid = stream_grid(B * N * N * upr, 256, 256 * 32);
  if (vec && dtype == TRS_F32)
    hipLaunchKernelGGL((ffm_fwd_kernel<float, uint4>), dim3(grid), dim3(256), 0, s, (const uint4*)x, (uint4*)out, B, N, upr);
  else if (vec)
    hipLaunchKernelGGL((ffm_fwd_kernel<bf16_t, uint4>), dim3(grid), dim3(256), 0, s, (const uint4*)x, (uint4*)out, B, N, upr);
  else if (dtype == TRS_F32)
    hipLaunchKernelGGL((ffm_fwd_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)out, B, N, upr);
  else
    hipLaunchKernelGGL((ffm_fwd_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, B, N, upr);
  return check_launch("ffm_fwd");
}

extern "C" int trs_ffm_bwd(const void* x, const void* g, int64_t B, int32_t N, int32_t E, int32_t dtype, void* dx,
                           trs_stream_t stream) {
  if (B == 0) return TRS_OK;  // empty batch: nothing to do (pointers may be NULL)
  TRS_REQUIRE(x && dx && (g || N < 2), TRS_EINVAL, "ffm_bwd: NULL pointer");
  TRS_CHECK_BNE("ffm_bwd");
  if (B == 0) return TRS_OK;
  hipStream_t s = (hipStream_t)stream;
  const int rb = E * dtype_size(dtype);
  const bool vec = rb % 16 == 0 && aligned16(x) && aligned16(g) && aligned16(dx);
  const int upr = vec ? rb / 16 : E;
  const int grid = stream_grid(B * N * N * upr, 256, 256 * 32);
  if (vec && dtype == TRS_F32)
    hipLaunchKernelGGL((ffm_bwd_kernel<float, uint4>), dim3(grid), dim3(256), 0, s, (const uint4*)x, (const uint4*)g, (uint4*)dx, B, N, upr);
  else if (vec)
    hipLaunchKernelGGL((ffm_bwd_kernel<bf16_t, uint4>), dim3(grid), dim3(256), 0, s, (const uint4*)x, (const uint4*)g, (uint4*)dx, B, N, upr);
  else if (dtype == TRS_F32)
    hipLaunchKernelGGL((ffm_bwd_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)x, (const float*)g, (float*)dx, B, N, upr);
  else
    hipLaunchKernelGGL((ffm_bwd_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)g, (bf16_t*)dx, B, N, upr);
  return check_launch("ffm_bwd");
}

template <typename IdxT>
static int ffm_fused_dispatch(const void* const* tables, int64_t V, int E, int dtype, const IdxT* idx,
                              const int64_t* offsets, int64_t B, int N, void* out, int32_t* err_flag, hipStream_t s) {
  const int rb = E * dtype_size(dtype);
  const bool vec = rb % 16 == 0 && aligned16(out);
  const int upr = vec ? rb / 16 : E;
  if (vec && is_pow2(upr) && upr <= 64 && N <= 255) {      // whole 16-byte units, a power of two of them per row: the sample-wise kernel
    int lg = 0;
    while ((1 << lg) < upr) ++lg;
    const int P = N * (N - 1) / 2;
    const size_t lds = (size_t)((2 * P + 15) / 16) * 16 + (size_t)N * 8;
    const bool stream = (size_t)N * V * rb > ((size_t)512 << 20);
    const int grid = (int)std::min<int64_t>(B, 256 * 8);
#define TRS_FFM_ROWS(TT, LG)                                                                                              \
  if (stream)                                                                                                             \
    hipLaunchKernelGGL((ffm_fused_fwd_rows_kernel<TT, IdxT, LG, true>), dim3(grid), dim3(256), lds, s,                     \
                       (const uint4* const*)tables, idx, offsets, (uint4*)out, B, N, V, err_flag);                        \
  else                                                                                                                    \
    hipLaunchKernelGGL((ffm_fused_fwd_rows_kernel<TT, IdxT, LG, false>), dim3(grid), dim3(256), lds, s,                    \
                       (const uint4* const*)tables, idx, offsets, (uint4*)out, B, N, V, err_flag)
#define TRS_FFM_ROWS_T(TT)                    \
  switch (lg) {                               \
    case 0: TRS_FFM_ROWS(TT, 0); break;       \
    case 1: TRS_FFM_ROWS(TT, 1); break;       \
    case 2: TRS_FFM_ROWS(TT, 2); break;       \
    case 3: TRS_FFM_ROWS(TT, 3); break;       \
    case 4: TRS_FFM_ROWS(TT, 4); break;       \
    case 5: TRS_FFM_ROWS(TT, 5); break;       \
    default: TRS_FFM_ROWS(TT, 6); break;      \
  }
    if (dtype == TRS_F32) {
      TRS_FFM_ROWS_T(float)
    } else {
      TRS_FFM_ROWS_T(bf16_t)
    }
#undef TRS_FFM_ROWS_T
#undef TRS_FFM_ROWS
    return check_launch("ffm_fused_fwd");
  }
  const int grid = stream_grid(B * N * N * upr, 256, 256 * 32);
  if (vec && dtype == TRS_F32)
    hipLaunchKernelGGL((ffm_fused_fwd_kernel<float, uint4, IdxT>), dim3(grid), dim3(256), 0, s, (const uint4* const*)tables, idx, offsets, (uint4*)out, B, N, upr, V, err_flag);
  else if (vec)
    hipLaunchKernelGGL((ffm_fused_fwd_kernel<bf16_t, uint4, IdxT>), dim3(grid), dim3(256), 0, s, (const uint4* const*)tables, idx, offsets, (uint4*)out, B, N, upr, V, err_flag);
  else if (dtype == TRS_F32)
    hipLaunchKernelGGL((ffm_fused_fwd_kernel<float, float, IdxT>), dim3(grid), dim3(256), 0, s, (const float* const*)tables, idx, offsets, (float*)out, B, N, upr, V, err_flag);
  else
    hipLaunchKernelGGL((ffm_fused_fwd_kernel<bf16_t, bf16_t, IdxT>), dim3(grid), dim3(256), 0, s, (const bf16_t* const*)tables, idx, offsets, (bf16_t*)out, B, N, upr, V, err_flag);
  return check_launch("ffm_fused_fwd");
}

extern "C" int trs_ffm_fused_fwd(const void* const* tables, int64_t V, int32_t E, int32_t dtype, const void* idx,
                                 int32_t idx_dtype, const int64_t* offsets, int64_t B, int32_t N, void* out,
                                 int32_t* err_flag, trs_stream_t stream) {
  if (B == 0 || N < 2) return TRS_OK;  // empty batch / no pairs: nothing to do (pointers may be NULL)
  TRS_REQUIRE(tables && idx && out, TRS_EINVAL, "ffm_fused_fwd: NULL pointer");
  TRS_REQUIRE(V > 0, TRS_EINVAL, "ffm_fused_fwd: bad V");
  TRS_CHECK_BNE("ffm_fused_fwd");
  TRS_REQUIRE(idx_dtype == TRS_I64 || idx_dtype == TRS_I32, TRS_EDTYPE, "ffm_fused_fwd: idx dtype %d", idx_dtype);
  if (B == 0 || N < 2) return TRS_OK;
  if (idx_dtype == TRS_I64)
    return ffm_fused_dispatch<int64_t>(tables, V, E, dtype, (const int64_t*)idx, offsets, B, N, out, err_flag, (hipStream_t)stream);
  return ffm_fused_dispatch<int32_t>(tables, V, E, dtype, (const int32_t*)idx, offsets, B, N, out, err_flag, (hipStream_t)stream);
}

template <typename IdxT>
static int ffm_fused_bwd_dispatch(const void* const* tables, int64_t V, int E, int dtype, const IdxT* idx,
                                  const int64_t* offsets, const void* gout, const int32_t* row_start,
                                  const int32_t* perm, int N, void* const* grads, hipStream_t s, int64_t BN,
                                  const int32_t* rid_t) {
  const int rb = E * dtype_size(dtype);
  const bool vec = rb % 16 == 0 && aligned16(gout);
  const int upr = vec ? rb / 16 : E;
  if (vec && is_pow2(upr) && upr <= 64 && N < 65536 && BN < ((int64_t)1 << 31)) {
    int lg = 0;
    while ((1 << lg) < upr) ++lg;
    const bool stream = (size_t)N * V * rb > ((size_t)512 << 20);
    const unsigned rcpN = (unsigned)((((uint64_t)1 << 32) + N - 1) / N);
    const dim3 grid((unsigned)stream_grid(V * upr, 256, 256 * 4), (unsigned)N);
#define TRS_FFMB(TT, LG)                                                                                                 \
  if (stream)                                                                                                           \
    hipLaunchKernelGGL((ffm_fused_bwd_rows_kernel<TT, IdxT, LG, true>), grid, dim3(256), 0, s, (const uint4* const*)tables, \
                       idx, offsets, (const uint4*)gout, row_start, perm, (unsigned)N, rcpN, V, (uint4* const*)grads,    \
                       rid_t, BN / N);                                                                                  \
  else                                                                                                                  \
    hipLaunchKernelGGL((ffm_fused_bwd_rows_kernel<TT, IdxT, LG, false>), grid, dim3(256), 0, s, (const uint4* const*)tables, \
                       idx, offsets, (const uint4*)gout, row_start, perm, (unsigned)N, rcpN, V, (uint4* const*)grads,    \
                       rid_t, BN / N)
#define TRS_FFMB_T(TT)                   \
  switch (lg) {                          \
    case 0: TRS_FFMB(TT, 0); break;      \
    case 1: TRS_FFMB(TT, 1); break;      \
    case 2: TRS_FFMB(TT, 2); break;      \
    case 3: TRS_FFMB(TT, 3); break;      \
    case 4: TRS_FFMB(TT, 4); break;      \
    case 5: TRS_FFMB(TT, 5); break;      \
    default: TRS_FFMB(TT, 6); break;     \
  }
    if (dtype == TRS_F32) {
      TRS_FFMB_T(float)
    } else {
      TRS_FFMB_T(bf16_t)
    }
#undef TRS_FFMB_T
#undef TRS_FFMB
    return check_launch("ffm_fused_bwd");
  }
  const int grid = stream_grid((int64_t)N * V * upr, 256, 256 * 32);
  if (vec && dtype == TRS_F32)
    hipLaunchKernelGGL((ffm_fused_bwd_kernel<float, uint4, IdxT>), dim3(grid), dim3(256), 0, s, (const uint4* const*)tables, idx, offsets, (const uint4*)gout, row_start, perm, N, upr, V, (uint4* const*)grads);
  else if (vec)
    hipLaunchKernelGGL((ffm_fused_bwd_kernel<bf16_t, uint4, IdxT>), dim3(grid), dim3(256), 0, s, (const uint4* const*)tables, idx, offsets, (const uint4*)gout, row_start, perm, N, upr, V, (uint4* const*)grads);
  else if (dtype == TRS_F32)
    hipLaunchKernelGGL((ffm_fused_bwd_kernel<float, float, IdxT>), dim3(grid), dim3(256), 0, s, (const float* const*)tables, idx, offsets, (const float*)gout, row_start, perm, N, upr, V, (float* const*)grads);
  else
    hipLaunchKernelGGL((ffm_fused_bwd_kernel<bf16_t, bf16_t, IdxT>), dim3(grid), dim3(256), 0, s, (const bf16_t* const*)tables, idx, offsets, (const bf16_t*)gout, row_start, perm, N, upr, V, (bf16_t* const*)grads);
  return check_launch("ffm_fused_bwd");
}

extern "C" int trs_ffm_fused_bwd(const void* const* tables, int64_t V, int32_t E, int32_t dtype, const void* idx,
                                 int32_t idx_dtype, const int64_t* offsets, const void* gout,
                                 const int32_t* row_start, const int32_t* perm, int64_t B, int32_t N,
                                 void* const* grad_tables, const int32_t* row_ids_t, trs_stream_t stream) {
  TRS_REQUIRE(tables && grad_tables && row_start, TRS_EINVAL, "ffm_fused_bwd: NULL pointer");
  TRS_REQUIRE(V > 0, TRS_EINVAL, "ffm_fused_bwd: bad V");
  TRS_CHECK_BNE("ffm_fused_bwd");
  TRS_REQUIRE(idx_dtype == TRS_I64 || idx_dtype == TRS_I32, TRS_EDTYPE, "ffm_fused_bwd: idx dtype %d", idx_dtype);
  TRS_REQUIRE(B == 0 || (idx && gout && perm), TRS_EINVAL, "ffm_fused_bwd: NULL pointer");
  if (idx_dtype == TRS_I64)
    return ffm_fused_bwd_dispatch<int64_t>(tables, V, E, dtype, (const int64_t*)idx, offsets, gout, row_start, perm, N, grad_tables, (hipStream_t)stream, B * (int64_t)N, row_ids_t);
  return ffm_fused_bwd_dispatch<int32_t>(tables, V, E, dtype, (const int32_t*)idx, offsets, gout, row_start, perm, N, grad_tables, (hipStream_t)stream, B * (int64_t)N, row_ids_t);
}
