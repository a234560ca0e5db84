// K5 on the matrix cores (bf16, channels-last tensors).
//
//   y[b,c,e] = bias[c] + sum_{n,h} Wc[c, n*H+h] * x0[b,n,e] * xk[b,h,e]
//            = bias[c] + sum_n x0[b,n,e] * ( sum_h Wc[(c,n),h] * xk[b,h,e] )
//
// The inner sum is a plain GEMM  T_n = W_n (C x H) * xk^T (H x pixels)  -- pixels = (b,e) pairs -- so the
// outer product Z = x0 (x) xk of the reference (25-84 GB at the BASELINE shape) is never formed, not even as an
// MFMA operand: the x0[n] factor is applied to the MFMA RESULT (4 FMAs per lane per 4 MFMAs), which keeps the
// VALU off the critical path.  With channels-last activations xk^T (B,E,H) the B operand of
// v_mfma_f32_16x16x32_bf16 (lane = pixel, 8 consecutive h) is one 16-byte load per lane, and with the output
// channels fed in the permuted order c(ct,q,i) = 32*(ct>>1) + 8q + 4*(ct&1) + i a lane ends up with 8
// consecutive channels of its pixel: y^T (B,E,C) is written with 16-byte stores and IS the next layer's
// B-operand layout.
//
// Work decomposition: wave = P pixel tiles (16*P pixels of one sample; E = 64 -> P = 4 = the whole sample),
// workgroup = 4 waves; the W fragments of one (channel-pair-tile j, field n) step (2*KS KiB) are staged in LDS
// (double buffered, one barrier per step) and shared by the 4 waves; every fragment read from LDS feeds P MFMAs.
#include <type_traits>

#include "trs_common.hpp"

namespace trs {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__host__ __device__ __forceinline__ int cin_chan_of_slot(int j, int ct2, int m) {
  return 32 * j + 8 * (m >> 2) + 4 * ct2 + (m & 3);
}

// Wp[((j*NP + n)*2 + ct2)*KS + ks][lane][8] = Wc[chan(j,ct2,lane&15)][n*H + 32*ks + 8*(lane>>4) + 0..7]
// (0 past H and for the field slots N <= n < NP that pad the last pipeline step)
__global__ __launch_bounds__(256) void cin_prepack_fwd_kernel(const bf16_t* __restrict__ Wc, bf16_t* __restrict__ Wp,
                                                              int C, int N, int NP, int H, int KS) {
  const int64_t total = (int64_t)(C / 32) * NP * 2 * KS * 64;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(t & 63);
    int64_t f = t >> 6;
    const int ks = (int)(f % KS); f /= KS;
    const int ct2 = (int)(f & 1); f >>= 1;
    const int n = (int)(f % NP);
    const int j = (int)(f / NP);
    const int c = cin_chan_of_slot(j, ct2, lane & 15);
    const int h0 = 32 * ks + 8 * (lane >> 4);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int h = h0 + jj;
      Wp[t * 8 + jj] = (h < H && n < N) ? Wc[(size_t)c * N * H + (size_t)n * H + h] : bf16_t{0};
    }
  }
}

// Pipeline step = NS fields of one channel-pair tile j = G = 2*NS groups (field, 16-channel tile) of KS*P MFMAs each.
// Inside a step everything is software-pipelined at group granularity: the KS fragments of group g+1 are read from LDS
// before the MFMAs of group g are issued (two register sets), and the x0 scaling of group g-1's result (4 FMAs per
// pixel tile) is placed between the MFMAs of group g (two result sets), so a wave keeps the matrix pipe fed without
// leaning on the other wave of its SIMD.  The fragment stream is continuous over the items a workgroup walks (the
// step after an item's last is the next item's first), fetched one step ahead with unconditional, clamped loads.
// Fields past N in the last step have zero fragments and zero x0 rows.
template <int KS, int P, int NS, bool TRI>
__global__ __launch_bounds__(256, 2) void cin_cl_fwd_kernel(const bf16_t* __restrict__ x0T, int ld0,
                                                            const bf16_t* __restrict__ xkT, int ldk,
                                                            const uint4* __restrict__ Wp, const float* __restrict__ bias,
                                                            bf16_t* __restrict__ yT, int64_t B, int N, int C, int E) {
  constexpr int PIX = 16 * P;
  constexpr int G = 2 * NS;
  constexpr int FR1 = 2 * KS * 64;                     // uint4 per (j,n)
  constexpr int FR = NS * FR1;                         // uint4 per step
  constexpr int NPF = (FR + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* Abuf = reinterpret_cast<uint4*>(smem);                          // [2][FR]
  float* bs = reinterpret_cast<float*>(smem + 2 * FR * 16);              // [C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  const int items_per_b = E / PIX;
  const int64_t nitems = B * items_per_b;
  const int nj = C / 32;
  const int npairs = (N + NS - 1) / NS;
  const int nsteps = nj * npairs;
  const int NP = npairs * NS;                                            // field slots incl. the zero ones
  unsigned short* x0w = reinterpret_cast<unsigned short*>(bs + C) + (size_t)wave * NP * PIX;   // [NP][16][P]
  for (int i = threadIdx.x; i < C; i += 256) bs[i] = bias ? bias[i] : 0.f;
  for (int v = lane; v < (NP - N) * PIX; v += 64) x0w[N * PIX + v] = 0;
  // A step's fragments are FR consecutive vectors of Wp.  The loads are issued by hand: the optimiser moves ordinary
  // loads of read-only memory next to their use (the LDS store at the end of the step), which would expose the whole
  // L2 round trip every step; an asm load stays where it is written and is waited for by the s_waitcnt of the commit.
#define TRS_CIN_FETCH(dst, step)                                                              \
  _Pragma("unroll") for (int k = 0; k < NPF; ++k) {                                           \
    const int i_ = threadIdx.x + 256 * k;                                                     \
    const uint4* p_ = Wp + (size_t)(step) * FR + (FR % 256 == 0 || i_ < FR ? i_ : FR - 1);    \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[k]) : "v"(p_));                 \
  }
#define TRS_CIN_COMMIT(par_, src)                                                             \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
  _Pragma("unroll") for (int k = 0; k < NPF; ++k) {                                           \
    asm volatile("" : "+v"(src[k]));                                                          \
    const int i_ = threadIdx.x + 256 * k;                                                     \
    if (FR % 256 == 0 || i_ < FR) Abuf[(par_) * FR + i_] = __builtin_bit_cast(uint4, src[k]); \
  }
  {
    u32x4 first[NPF];
    TRS_CIN_FETCH(first, 0)
    TRS_CIN_COMMIT(0, first)
  }
  int par = 0;
  __syncthreads();
  for (int64_t it0 = (int64_t)blockIdx.x * 4; it0 < nitems; it0 += (int64_t)gridDim.x * 4) {
    const int64_t it = it0 + wave;
    const bool live = it < nitems;
    const int64_t b = live ? it / items_per_b : 0;
    const int e0 = live ? (int)(it - b * items_per_b) * PIX : 0;
    const int64_t pix0 = b * E + e0;                                     // first pixel row of this wave
    uint4 Bf[P][KS];
#pragma unroll
    for (int t = 0; t < P; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 v = *reinterpret_cast<const uint4*>(xkT + (pix0 + 16 * t + r) * ldk + 32 * ks + 8 * q);
        Bf[t][ks] = live ? v : make_uint4(0, 0, 0, 0);
      }
    // x0 of this wave's pixels, [field][r][t]: the scale factors of a lane's P pixel tiles are one LDS read.  The
    // array is the wave's own (LDS operations of one wave complete in order: no barrier)
    for (int v = lane; v < PIX * ((N + 7) / 8); v += 64) {
      const int p = v % PIX, ch = v / PIX;
      const uint4 u = *reinterpret_cast<const uint4*>(x0T + (pix0 + p) * ld0 + 8 * ch);
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int n = 8 * ch + jj;
        if (n < N) x0w[(n * 16 + (p & 15)) * P + (p >> 4)] = (unsigned short)(jj & 1 ? w[jj >> 1] >> 16 : w[jj >> 1] & 0xffffu);
      }
    }
    // the xk fragments are waited for here, once: inside the loop the only loads in flight are the next step's
#pragma unroll
    for (int t = 0; t < P; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(Bf[t][ks].x), "v"(Bf[t][ks].y), "v"(Bf[t][ks].z), "v"(Bf[t][ks].w));
    f32x4 acc[P][2];
    int step = 0;
    for (int j = 0; j < nj; ++j) {
#pragma unroll
      for (int ct2 = 0; ct2 < 2; ++ct2) {
        const float4 bv = *reinterpret_cast<const float4*>(bs + 32 * j + 8 * q + 4 * ct2);
#pragma unroll
        for (int t = 0; t < P; ++t) acc[t][ct2] = f32x4{bv.x, bv.y, bv.z, bv.w};
      }
      // one basic block per step: the next step's loads stay at its top (nothing to sink them into)
      for (int n0 = 0; n0 < NP; n0 += NS) {
      const uint4* A = Abuf + par * FR;
      u32x4 nxt[NPF];
      step = step + 1 < nsteps ? step + 1 : 0;
      TRS_CIN_FETCH(nxt, step)
      __builtin_amdgcn_sched_barrier(0);
      uint4 Af[2][KS];
      f32x4 T[2][P];
      unsigned xraw[2][(P + 1) / 2];            // x0 of group g's field at this lane's P pixel tiles, bf16 pairs
      auto loadA = [&](int g) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) Af[g & 1][ks] = A[(g * KS + ks) * 64 + lane];
      };
      auto loadX = [&](int g) {                 // read while group g's MFMAs run, used between those of group g+1
        const unsigned short* xp = x0w + ((n0 + (g >> 1)) * 16 + r) * P;
        if constexpr (P == 4) {
          const uint2 u = *reinterpret_cast<const uint2*>(xp);
          xraw[g & 1][0] = u.x; xraw[g & 1][1] = u.y;
        } else if constexpr (P == 2) {
          xraw[g & 1][0] = *reinterpret_cast<const unsigned*>(xp);
        } else {
          xraw[g & 1][0] = *xp;
        }
      };
      auto scale = [&](int g, int t) {         // acc[t][ct2] += x0[n, pixel tile t] * T of group g = (field n0 + g/2, tile g&1)
        const unsigned w = xraw[g & 1][t >> 1];
        const float xv = __uint_as_float(t & 1 ? w & 0xffff0000u : w << 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][g & 1][i] = fmaf(xv, T[g & 1][t][i], acc[t][g & 1][i]);
      };
      loadA(0);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g + 1 < G) loadA(g + 1);
        loadX(g);
        // lower-triangular weights (tri): field n has no weight on h > n, so its k-steps past n / 32 are all zero
        const int ks_end = TRI ? (n0 + (g >> 1)) / 32 + 1 : KS;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (!TRI || ks == 0 || ks < ks_end) {
#pragma unroll
            for (int t = 0; t < P; ++t)
              T[g & 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8, Af[g & 1][ks]), __builtin_bit_cast(bf16x8, Bf[t][ks]),
                  ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : T[g & 1][t], 0, 0, 0);
          }
          if (g > 0) {
#pragma unroll
            for (int t = 0; t < P; ++t)
              if ((t * KS) / P == ks) scale(g - 1, t);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int t = 0; t < P; ++t) scale(G - 1, t);
      TRS_CIN_COMMIT(par ^ 1, nxt)
      par ^= 1;
      __syncthreads();
      }
      if (live) {
#pragma unroll
        for (int t = 0; t < P; ++t) {
          float f[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) { f[i] = acc[t][0][i]; f[4 + i] = acc[t][1][i]; }
          *reinterpret_cast<uint4*>(yT + (pix0 + 16 * t + r) * (int64_t)C + 32 * j + 8 * q) = Vec16<bf16_t>::pack(f);
        }
      }
    }
  }
}

#undef TRS_CIN_FETCH
#undef TRS_CIN_COMMIT

static size_t cin_fwd_frag_bytes(int NP, int KS, int C) { return (size_t)(C / 32) * NP * 2 * KS * 64 * 16; }

size_t cin_mfma_fwd_workspace_bytes(int N, int H, int C) {
  const int KS = (H + 31) / 32;
  return cin_fwd_frag_bytes(N + 2, KS, C) + (size_t)C * 4 + 512;   // up to two padding field slots
}

__global__ void cin_bias_to_f32_kernel(const bf16_t* __restrict__ b, float* __restrict__ o, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) o[i] = to_f32(b[i]);
}

static bool cin_cl_covers(int N, int H, int C, int E) {
  const int KS = (H + 31) / 32;
  return C % 32 == 0 && E % 16 == 0 && (KS == 1 || KS == 2 || KS == 4 || KS == 8) && N >= 1;
}

// x0T: (B,E,ld0) with ld0 % 8 == 0 and zeros past N; xkT: rows (B*E) of stride ldk >= 32*ceil(H/32), zeros past H.
int cin_cl_fwd(const void* x0T, int ld0, const void* xkT, int ldk, const void* Wc, const void* bias, int64_t B, int N,
               int H, int C, int E, int tri, void* yT, void* workspace, size_t ws_bytes, hipStream_t s) {
  if (!cin_cl_covers(N, H, C, E)) return 1;
  const int KS = (H + 31) / 32;
  if (ld0 % 8 != 0 || ld0 < ((N + 7) / 8) * 8 || ldk % 8 != 0 || ldk < 32 * KS || !aligned16(x0T) || !aligned16(xkT) ||
      !aligned16(yT) || workspace == nullptr)
    return 1;
  if (ws_bytes < cin_mfma_fwd_workspace_bytes(N, H, C)) return fail(TRS_EWORKSPACE, "cin_cl_fwd: workspace too small");
  int P = E % 64 == 0 ? 4 : (E % 32 == 0 ? 2 : 1);
  if (KS == 8 && P == 4) P = 2;                        // the xk fragments (P*KS*4 registers) must leave room for the rest
  // fields per pipeline step: the one that pads N least, three when both do (fewer barriers) and the stage fits
  auto lds_for = [&](int NS_) {
    const int np = (N + NS_ - 1) / NS_ * NS_;
    return (size_t)2 * NS_ * 2 * KS * 64 * 16 + (size_t)C * 4 + (size_t)4 * np * 16 * P * 2;
  };
  int NS = 2;
  while (P > 1 && lds_for(2) > 78 * 1024) P >>= 1;     // many fields: the per-wave x0 array (N x 16 P x 2 bytes) must fit
  if (KS <= 4 && (N + 2) / 3 * 3 <= (N + 1) / 2 * 2 && lds_for(3) <= 78 * 1024) NS = 3;
  const size_t lds = lds_for(NS);
  if (lds > 78 * 1024) return 1;                       // two workgroups per CU
  const int NP = (N + NS - 1) / NS * NS;
  bf16_t* Wp = (bf16_t*)workspace;
  float* bf = (float*)((char*)workspace + cin_fwd_frag_bytes(NP, KS, C));
  const int64_t total = (int64_t)(C / 32) * NP * 2 * KS * 64;
  hipLaunchKernelGGL(cin_prepack_fwd_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, s,
                     (const bf16_t*)Wc, Wp, C, N, NP, H, KS);
  if (bias) hipLaunchKernelGGL(cin_bias_to_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const bf16_t*)bias, bf, C);
  const int64_t nitems = B * (E / (16 * P));
  const int grid = (int)std::min<int64_t>((nitems + 3) / 4, 256 * 2);
#define TRS_CINF(KS_, P_, NS_) \
  do {                         \
    if (tri && KS_ == 2) TRS_CINF_T(KS_, P_, NS_, (KS_ == 2)); /* KS = 4 (97..128 fields): the skipping form spills, */ \
    else TRS_CINF_T(KS_, P_, NS_, false);                      /* the plain kernel multiplies the zero k-steps too    */ \
  } while (0)
#define TRS_CINF_T(KS_, P_, NS_, TRI_)                                                                              \
  do {                                                                                                              \
    auto kern = cin_cl_fwd_kernel<KS_, P_, NS_, TRI_>;                                                              \
    static size_t attr_lds = 0;                                                                                     \
    if (lds > 64 * 1024 && lds > attr_lds) {                                                                        \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return check_launch("cin_cl_fwd: LDS attribute");                                                           \
      attr_lds = lds;                                                                                               \
    }                                                                                                               \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, (const bf16_t*)x0T, ld0, (const bf16_t*)xkT, ldk,       \
                       (const uint4*)Wp, bias ? bf : (const float*)nullptr, (bf16_t*)yT, B, N, C, E);               \
  } while (0)
#define TRS_CINF_NS(KS_, P_)         \
  do {                               \
    if (NS == 3) TRS_CINF(KS_, P_, 3); \
    else TRS_CINF(KS_, P_, 2);       \
  } while (0)
#define TRS_CINF_P(KS_)              \
  do {                               \
    if (P == 4) TRS_CINF_NS(KS_, 4); \
    else if (P == 2) TRS_CINF_NS(KS_, 2); \
    else TRS_CINF_NS(KS_, 1);        \
  } while (0)
  switch (KS) {
    case 1: TRS_CINF_P(1); break;
    case 2: TRS_CINF_P(2); break;
    case 4: TRS_CINF_P(4); break;
    default:
      if (P == 2) TRS_CINF(8, 2, 2);
      else TRS_CINF(8, 1, 2);
      break;
  }
#undef TRS_CINF_P
#undef TRS_CINF_NS
#undef TRS_CINF
#undef TRS_CINF_T
  return check_launch("cin_cl_fwd");
}

// =============================================================================================
// backward, data gradients (channels-last):  S_n[h,pix] = sum_c Wc[c,(n,h)] * gy[c,pix]   (MFMA, K = c)
//     dxk[h,pix] = sum_n x0[n,pix] * S_n[h,pix]          dx0[n,pix] = sum_h xk[h,pix] * S_n[h,pix]
// Same pipeline as the forward (wave = P pixel tiles of one sample, the W^T fragments of NS fields of one 32-h tile
// staged per step, groups of KC*P MFMAs software-pipelined against the VALU work of the previous group).  Both
// results are linear in S, so the contraction over c may be cut into passes of 32*KC channels whose partial S go
// through the same epilogue: with C = 256 the gy fragments of a pass are 64 registers instead of 128, which is what
// lets a wave keep four pixel tiles (an LDS fragment then feeds four MFMAs) at two waves per SIMD.
// dxk accumulates in registers (h on the D rows, permuted so a lane owns 8 consecutive h -> 16-byte stores).
// dx0[n] is a reduction over h = over the D rows: 8 FMAs per lane and tile, then the four tiles' partials are summed
// over the four 16-lane rows with three v_permlane swaps (every row ends up with one tile's total) and all 64 lanes
// add into the wave's fp32 LDS array [n][pixel] with one ds_add_f32.
// WpT[((((jh*npass + pass)*NP + n)*2 + ct2)*KC + kc][lane][8]
//     = Wc[32*(pass*KC + kc) + 8*(lane>>4) + 0..7][n*H + hslot(jh,ct2,lane&15)]      (0 for n >= N, h >= H)
__global__ __launch_bounds__(256) void cin_prepack_bwd_kernel(const bf16_t* __restrict__ Wc, bf16_t* __restrict__ WpT,
                                                              int C, int N, int NP, int H, int KSH, int KC, int npass) {
  const int64_t total = (int64_t)KSH * npass * NP * 2 * KC * 64;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(t & 63);
    int64_t f = t >> 6;
    const int kc = (int)(f % KC); f /= KC;
    const int ct2 = (int)(f & 1); f >>= 1;
    const int n = (int)(f % NP); f /= NP;
    const int pass = (int)(f % npass);
    const int jh = (int)(f / npass);
    const int h = cin_chan_of_slot(jh, ct2, lane & 15);
    const int c0 = 32 * (pass * KC + kc) + 8 * (lane >> 4);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
      WpT[t * 8 + jj] = (h < H && n < N) ? Wc[(size_t)(c0 + jj) * N * H + (size_t)n * H + h] : bf16_t{0};
  }
}

// sum the P per-lane partials over the four 16-lane rows; returns the total of tile cin_red_tile(q) in every lane of row q
template <int P>
__device__ __forceinline__ float cin_row_reduce(const float (&part)[P]) {
  if constexpr (P == 4) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[0]), __float_as_uint(part[1]), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[2]), __float_as_uint(part[3]), false, false);
    const float x = __uint_as_float(a[0]) + __uint_as_float(a[1]);     // rows 0,1: tile 0 (two partials); rows 2,3: tile 1
    const float y = __uint_as_float(b[0]) + __uint_as_float(b[1]);     // the same for tiles 2, 3
    const auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(c[0]) + __uint_as_float(c[1]);              // rows: tile 0, 2, 1, 3
  } else if constexpr (P == 2) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[0]), __float_as_uint(part[1]), false, false);
    const float x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(c[0]) + __uint_as_float(c[1]);              // rows 0,1: tile 0; rows 2,3: tile 1
  } else {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(part[0]), __float_as_uint(part[0]), false, false);
    const float x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(c[0]) + __uint_as_float(c[1]);              // every row: tile 0
  }
}
template <int P>
__device__ __forceinline__ int cin_red_tile(int q) { return P == 4 ? ((q & 1) * 2 + (q >> 1)) : (P == 2 ? (q >> 1) : 0); }
template <int P>
__device__ __forceinline__ bool cin_red_owner(int q) { return P == 4 ? true : (P == 2 ? (q & 1) == 0 : q == 0); }

template <int KC, int P, int NS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void cin_cl_bwd_data_kernel(
    const bf16_t* __restrict__ x0T, int ld0, const bf16_t* __restrict__ xkT, int ldk, const bf16_t* __restrict__ gyT,
    const uint4* __restrict__ WpT, bf16_t* __restrict__ dx0T, bf16_t* __restrict__ dxkT, int ldo, int64_t B, int N, int H,
    int C, int E, int npass, int tri, int ldg /* channels between two pixels of gyT (>= C: only the first C are read) */) {
  constexpr int NT = 64 * WAVES;
  constexpr int PIX = 16 * P;
  constexpr int G = 2 * NS;
  constexpr int FR1 = 2 * KC * 64;   // uint4 per (jh,pass,n)
  constexpr int FR = NS * FR1;       // uint4 per step
  constexpr int NPF = (FR + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* Abuf = reinterpret_cast<uint4*>(smem);                                   // [2][FR]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  const int items_per_b = E / PIX;
  const int64_t nitems = B * items_per_b;
  const int KSH = (H + 31) / 32;
  const int npairs = (N + NS - 1) / NS;
  const int NP = npairs * NS;
  float* dx0s = reinterpret_cast<float*>(smem + 2 * FR * 16) + (size_t)wave * NP * PIX;                // [NP][PIX] fp32
  unsigned short* x0w = reinterpret_cast<unsigned short*>(smem + 2 * FR * 16 + (size_t)WAVES * NP * PIX * 4) +
                        (size_t)wave * NP * PIX;                                                       // [NP][16][P]
  for (int v = lane; v < (NP - N) * PIX; v += 64) x0w[N * PIX + v] = 0;
#define TRS_CIN_FETCH(dst, step)                                                              \
  _Pragma("unroll") for (int k = 0; k < NPF; ++k) {                                           \
    const int i_ = threadIdx.x + NT * k;                                                      \
    const uint4* p_ = WpT + (size_t)(step) * FR + (FR % NT == 0 || i_ < FR ? i_ : FR - 1);    \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[k]) : "v"(p_));                 \
  }
#define TRS_CIN_COMMIT(par_, src)                                                             \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
  _Pragma("unroll") for (int k = 0; k < NPF; ++k) {                                           \
    asm volatile("" : "+v"(src[k]));                                                          \
    const int i_ = threadIdx.x + NT * k;                                                      \
    if (FR % NT == 0 || i_ < FR) Abuf[(par_) * FR + i_] = __builtin_bit_cast(uint4, src[k]);  \
  }
  {
    u32x4 first[NPF];
    TRS_CIN_FETCH(first, 0)
    TRS_CIN_COMMIT(0, first)
  }
  int par = 0;
  __syncthreads();
  for (int64_t it0 = (int64_t)blockIdx.x * WAVES; it0 < nitems; it0 += (int64_t)gridDim.x * WAVES) {
    const int64_t it = it0 + wave;
    const bool live = it < nitems;
    const int64_t b = live ? it / items_per_b : 0;
    const int e0 = live ? (int)(it - b * items_per_b) * PIX : 0;
    const int64_t pix0 = b * E + e0;
    for (int v = lane; v < N * PIX; v += 64) dx0s[v] = 0.f;
    for (int v = lane; v < PIX * ((N + 7) / 8); v += 64) {
      const int p = v % PIX, ch = v / PIX;
      const uint4 u = *reinterpret_cast<const uint4*>(x0T + (pix0 + p) * ld0 + 8 * ch);
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int n = 8 * ch + jj;
        if (n < N) x0w[(n * 16 + (p & 15)) * P + (p >> 4)] = (unsigned short)(jj & 1 ? w[jj >> 1] >> 16 : w[jj >> 1] & 0xffffu);
      }
    }
    uint4 Bg[P][KC];
    // lower-triangular weights (tri): the 32-h tile jh only gets contributions from the fields n >= 32 jh, so the steps
    // below the one that holds field 32 jh are left out of the walk (and of the fragment stream)
    auto first_n0 = [&](int jh_) { return tri ? (32 * jh_ / NS) * NS : 0; };
    for (int jh = 0; jh < KSH; ++jh) {
      f32x4 acc[P][2];
      float xkd[P][2][4];
#pragma unroll
      for (int t = 0; t < P; ++t) {
        acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint4 u = *reinterpret_cast<const uint4*>(xkT + (pix0 + 16 * t + r) * ldk + 32 * jh + 8 * q);
        float f[8];
        Vec16<bf16_t>::unpack(u, f);
#pragma unroll
        for (int i = 0; i < 4; ++i) { xkd[t][0][i] = live ? f[i] : 0.f; xkd[t][1][i] = live ? f[4 + i] : 0.f; }
      }
      for (int pass = 0; pass < npass; ++pass) {
        if (npass > 1 || jh == 0) {
#pragma unroll
          for (int t = 0; t < P; ++t)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
              const uint4 v = *reinterpret_cast<const uint4*>(gyT + (pix0 + 16 * t + r) * (int64_t)ldg + 32 * (pass * KC + kc) + 8 * q);
              Bg[t][kc] = live ? v : make_uint4(0, 0, 0, 0);
            }
        }
        // everything loaded so far is waited for here: inside the step loop only the next step's fragments are in flight
#pragma unroll
        for (int t = 0; t < P; ++t) {
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) asm volatile("" ::"v"(Bg[t][kc].x), "v"(Bg[t][kc].y), "v"(Bg[t][kc].z), "v"(Bg[t][kc].w));
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(xkd[t][0][i]), "v"(xkd[t][1][i]));
        }
        for (int n0 = first_n0(jh); n0 < NP; n0 += NS) {      // one basic block per step
          const uint4* A = Abuf + par * FR;
          u32x4 nxt[NPF];
          int step;                                  // the step after this one (the next item's first after the last)
          if (n0 + NS < NP) {
            step = (jh * npass + pass) * npairs + (n0 + NS) / NS;
          } else {
            const int jn = pass + 1 < npass ? jh : (jh + 1 < KSH ? jh + 1 : 0);
            const int pn = pass + 1 < npass ? pass + 1 : 0;
            step = (jn * npass + pn) * npairs + first_n0(jn) / NS;
          }
          TRS_CIN_FETCH(nxt, step)
          __builtin_amdgcn_sched_barrier(0);
          uint4 Af[2][KC];
          f32x4 T[2][P];
          unsigned xraw[2][(P + 1) / 2];
          float part[P];
          auto loadA = [&](int g) {
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) Af[g & 1][kc] = A[(g * KC + kc) * 64 + lane];
          };
          auto loadX = [&](int g) {
            const unsigned short* xp = x0w + ((n0 + (g >> 1)) * 16 + r) * P;
            if constexpr (P == 4) {
              const uint2 u = *reinterpret_cast<const uint2*>(xp);
              xraw[g & 1][0] = u.x; xraw[g & 1][1] = u.y;
            } else if constexpr (P == 2) {
              xraw[g & 1][0] = *reinterpret_cast<const unsigned*>(xp);
            } else {
              xraw[g & 1][0] = *xp;
            }
          };
          auto scale = [&](int g, int t) {       // group g = (field n0 + g/2, h half-tile g&1)
            const unsigned w = xraw[g & 1][t >> 1];
            const float xv = __uint_as_float(t & 1 ? w & 0xffff0000u : w << 16);
            float pt = (g & 1) ? part[t] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              acc[t][g & 1][i] = fmaf(xv, T[g & 1][t][i], acc[t][g & 1][i]);
              pt = fmaf(xkd[t][g & 1][i], T[g & 1][t][i], pt);
            }
            part[t] = pt;
          };
          // dx0s[field][pixel] += total: a plain read-modify-write (the array is this wave's; an LDS float atomic costs
          // ~150 LDS cycles per wave instruction).  The old value is read one group ahead of the add.
          float dold = 0.f;
          auto dx0_slot = [&](int g) { return dx0s + (n0 + (g >> 1)) * PIX + 16 * cin_red_tile<P>(q) + r; };
          auto reduce = [&](int g) {             // after both half-tiles of field n0 + g/2 went through scale()
            const float z = cin_row_reduce<P>(part);
            if (cin_red_owner<P>(q)) *dx0_slot(g) = dold + z;
          };
          loadA(0);
#pragma unroll
          for (int g = 0; g < G; ++g) {
            if (g + 1 < G) loadA(g + 1);
            loadX(g);
            if (g & 1) dold = *dx0_slot(g);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
              for (int t = 0; t < P; ++t)
                T[g & 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, Af[g & 1][kc]), __builtin_bit_cast(bf16x8, Bg[t][kc]),
                    kc == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : T[g & 1][t], 0, 0, 0);
              if (g > 0) {
#pragma unroll
                for (int t = 0; t < P; ++t)
                  if ((t * KC) / P == kc) scale(g - 1, t);
                if (kc == KC - 1 && ((g - 1) & 1)) reduce(g - 1);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#pragma unroll
          for (int t = 0; t < P; ++t) scale(G - 1, t);
          reduce(G - 1);
          TRS_CIN_COMMIT(par ^ 1, nxt)
          par ^= 1;
          __syncthreads();
        }
      }
      if (tri) {
        // xk IS x0: dxk[h] and dx0[h] are gradients of the same values -- summed here in fp32 (the wave's own array)
        // and rounded once, instead of two bf16 tensors that the caller adds
#pragma unroll
        for (int t = 0; t < P; ++t)
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int h = 32 * jh + 8 * q + k;
            if (h < N) dx0s[h * PIX + 16 * t + r] += acc[t][k >> 2][k & 3];
          }
      } else if (live) {
#pragma unroll
        for (int t = 0; t < P; ++t) {
          float f[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) { f[i] = acc[t][0][i]; f[4 + i] = acc[t][1][i]; }
          *reinterpret_cast<uint4*>(dxkT + (pix0 + 16 * t + r) * (int64_t)ldo + 32 * jh + 8 * q) = Vec16<bf16_t>::pack(f);
        }
      }
    }
    if (live) {
      for (int v = lane; v < PIX * (ld0 / 8); v += 64) {
        const int p = v % PIX, ch = v / PIX;
        float f[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int n = 8 * ch + jj;
          f[jj] = n < N ? dx0s[n * PIX + p] : 0.f;
        }
        *reinterpret_cast<uint4*>(dx0T + (pix0 + p) * ld0 + 8 * ch) = Vec16<bf16_t>::pack(f);
      }
    }
  }
}
#undef TRS_CIN_FETCH
#undef TRS_CIN_COMMIT

static size_t cin_bwd_frag_bytes(int KSH, int NP, int KCT) { return (size_t)KSH * NP * 2 * KCT * 64 * 16; }

size_t cin_mfma_bwd_data_workspace_bytes(int N, int H, int C) {
  const int KSH = (H + 31) / 32, KCT = C / 32;
  return cin_bwd_frag_bytes(KSH, N + 2, KCT) + 256;
}

int cin_cl_bwd_data(const void* x0T, int ld0, const void* xkT, int ldk, const void* gyT, const void* Wc, int64_t B, int N,
                    int H, int C, int E, int tri, void* dx0T, void* dxkT, int ldo, void* workspace, size_t ws_bytes,
                    hipStream_t s, int ldg = 0) {
  const int KSH = (H + 31) / 32, KCT = C / 32;
  if (ldg == 0) ldg = C;
  if (ldg < C || ldg % 8 != 0) return 1;
  if (C % 32 != 0 || E % 16 != 0 || !(KCT == 1 || KCT == 2 || KCT == 4 || KCT == 8) || ld0 % 8 != 0 || ldk % 8 != 0 ||
      ldk < 32 * KSH || ldo < 32 * KSH || ldo % 8 != 0 || workspace == nullptr)
    return 1;
  if (ws_bytes < cin_mfma_bwd_data_workspace_bytes(N, H, C)) return fail(TRS_EWORKSPACE, "cin_cl_bwd_data: workspace");
  int P = E % 64 == 0 ? 4 : (E % 32 == 0 ? 2 : 1);
  // C = 256: two passes of 128 channels (one pass with 512-register waves and four waves per workgroup measured 1.06 vs
  // 0.93 ms: the accumulators end up in AGPRs and are copied out for every VALU use)
  const int KC = KCT == 8 ? 4 : KCT;
  const int npass = KCT / KC;
  const int WAVES = 8;
  auto lds_for = [&](int NS_) {
    const int np = (N + NS_ - 1) / NS_ * NS_;
    return (size_t)2 * NS_ * 2 * KC * 64 * 16 + (size_t)WAVES * np * 16 * P * (4 + 2);
  };
  int NS = 1;
  const size_t cap = 156 * 1024;
  while (P > 1 && lds_for(1) > cap) P >>= 1;     // many fields: the per-wave x0 / dx0 arrays (N x 16 P x 6 bytes) must fit
  // (three fields per step at 64 channels per pass, P = 4 needs 17 registers more than a wave has)
  if ((N + 2) / 3 * 3 <= (N + 1) / 2 * 2 && lds_for(3) <= cap && KC <= 4 && !(KC == 2 && P == 4)) NS = 3;
  else if (lds_for(2) <= cap) NS = 2;
  const size_t lds = lds_for(NS);
  if (lds > cap) return 1;
  const int NP = (N + NS - 1) / NS * NS;
  bf16_t* WpT = (bf16_t*)workspace;
  const int64_t total = (int64_t)KSH * npass * NP * 2 * KC * 64;
  hipLaunchKernelGGL(cin_prepack_bwd_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, s,
                     (const bf16_t*)Wc, WpT, C, N, NP, H, KSH, KC, npass);
  const int64_t nitems = B * (E / (16 * P));
  const int grid = (int)std::min<int64_t>((nitems + WAVES - 1) / WAVES, 256);
  // (generic lambda: only the combinations the choice of NS above can produce are instantiated)
  int rc_launch = 0;
  auto launch = [&](auto kc_c, auto p_c, auto ns_c) {
    constexpr int KC_ = decltype(kc_c)::value, P_ = decltype(p_c)::value, NS_ = decltype(ns_c)::value;
    if constexpr (!(NS_ == 3 && KC_ == 2 && P_ == 4)) {
      auto kern = cin_cl_bwd_data_kernel<KC_, P_, NS_, 8>;
      static size_t attr_lds = 0;
      if (lds > 64 * 1024 && lds > attr_lds) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
          rc_launch = check_launch("cin_cl_bwd_data: LDS attribute");
          return;
        }
        attr_lds = lds;
      }
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * 8), lds, s, (const bf16_t*)x0T, ld0, (const bf16_t*)xkT, ldk,
                         (const bf16_t*)gyT, (const uint4*)WpT, (bf16_t*)dx0T, (bf16_t*)dxkT, ldo, B, N, H, C, E, npass, tri,
                         ldg);
    }
  };
  auto launch_ns = [&](auto kc_c, auto p_c) {
    if (NS == 3) launch(kc_c, p_c, std::integral_constant<int, 3>{});
    else if (NS == 2) launch(kc_c, p_c, std::integral_constant<int, 2>{});
    else launch(kc_c, p_c, std::integral_constant<int, 1>{});
  };
  auto launch_p = [&](auto kc_c) {
    if (P == 4) launch_ns(kc_c, std::integral_constant<int, 4>{});
    else if (P == 2) launch_ns(kc_c, std::integral_constant<int, 2>{});
    else launch_ns(kc_c, std::integral_constant<int, 1>{});
  };
  switch (KC) {
    case 1: launch_p(std::integral_constant<int, 1>{}); break;
    case 2: launch_p(std::integral_constant<int, 2>{}); break;
    default: launch_p(std::integral_constant<int, 4>{}); break;
  }
  if (rc_launch != 0) return rc_launch;
  return check_launch("cin_cl_bwd_data");
}

// =============================================================================================
// backward, weight gradient (channels-FIRST operands: the contraction runs over pixels, so pixels must be
// the contiguous K axis):   dW[c,(n,h)] = sum_{b,e} gy[b,c,e] * x0[b,n,e] * xk[b,h,e]
// GEMM view per field n:  dW_n (C x H) = (gy .* x0_n) (C x pixels) * xk^T (pixels x H).
// Workgroup = 8 waves owns a (NG = 8 fields) x (CB channels) x (16*HW*WH h) block of dW and streams a range of
// samples.  Per sample the gy tile (CB x E), the xk tile and the 8 x0 rows are staged once in LDS (rows padded to
// E*2+16 bytes: conflict-free ds_read_b128).  Wave w = (field quad w&1, channel pair (w>>1)%CW, h part) owns
// 4 fields x 2 c-tiles x HW h-tiles = up to 32 accumulator tiles: a gy fragment read from LDS is scaled by four
// x0 rows (one A operand per field) and every A operand is used for HW MFMAs, so a wave reads 2+HW fragments
// from LDS per 8*HW MFMAs and the block pulls (CB + 16*HW*WH + 8) rows from L2 per 64*HW MFMAs per wave --
// half the L2 bytes per flop of a 2-field x all-C tiling, which is what bounds this kernel.
// Partial sums per sample-split go to a workspace and are reduced by a second kernel (no float atomics).
constexpr int DW_NG = 8;      // fields per workgroup
constexpr int DW_WAVES = 8;

// NW fields per wave (DW_NG / NW field groups), WH wave groups along h, the other waves along c (32 channels each):
// every gy fragment is unpacked once and scaled by NW x0 rows, and every scaled fragment feeds HW MFMAs -- the VALU cost
// per MFMA is (16 + 8/NW) / HW instructions, so wide layers run NW = 2, HW = 8 (all of H = 128 in one wave: 2.5 per
// MFMA; the NW = 4, HW = 4 split of the same workgroup tile needs 4.5 and is VALU-bound).
template <int NW, int WH, int HW /* h tiles per wave */, int KE /* E/32 */>
__global__ __launch_bounds__(512) void cin_dw_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x0,
                                                     const bf16_t* __restrict__ xk, float* __restrict__ dWpart,
                                                     int64_t B, int N, int H, int C, int nsplit,
                                                     int gy_bs /* elements between two samples of gy (>= C*E) */) {
  constexpr int FG = DW_NG / NW;            // field groups (waves along the fields)
  constexpr int CW = DW_WAVES / (FG * WH);  // waves along c
  constexpr int CB = 32 * CW;               // channels per block
  constexpr int HBK = 16 * HW * WH;         // h per block
  constexpr int E = 32 * KE;
  constexpr int RS = E * 2 + 16;            // padded row stride in bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [2 buffers] x { gy_s [CB][RS] | xk_s [HBK][RS] | x0_s [DW_NG][RS] }
  constexpr int ROWS = CB + HBK + DW_NG;
  constexpr int BUF = ROWS * RS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  const int ng = (N + DW_NG - 1) / DW_NG;
  const int cbc = C / CB;
  const int hb_count = (H + HBK - 1) / HBK;
  int bid = blockIdx.x;
  const int g = bid % ng; bid /= ng;                 // field group fastest: co-scheduled blocks share samples in L2
  const int cb = bid % cbc; bid /= cbc;
  const int hb = bid % hb_count; bid /= hb_count;
  const int split = bid;
  const int nq = wave % FG, cw = (wave / FG) % CW, hw = (wave / FG) / CW;
  const int64_t per = (B + nsplit - 1) / nsplit;
  const int64_t b_lo = split * per, b_hi = b_lo + per < B ? b_lo + per : B;
  f32x4 acc[NW][2][HW];
#pragma unroll
  for (int nn = 0; nn < NW; ++nn)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int ht = 0; ht < HW; ++ht) acc[nn][ct][ht] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int VPR = E / 8;                          // 16-byte vectors per row
  constexpr int TOTV = ROWS * VPR;                    // vectors per sample stage
  constexpr int NV = (TOTV + 511) / 512;
  // per-thread source pointers advance by a fixed per-sample stride: precompute both
  const bf16_t* src[NV];
  int sstride[NV];
  int doff[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = threadIdx.x + 512 * k;
    src[k] = nullptr; sstride[k] = 0; doff[k] = 0;
    if (v < TOTV) {
      const int row = v / VPR, col = v - row * VPR;
      doff[k] = row * RS + col * 16;
      if (row < CB) {
        sstride[k] = gy_bs;
        src[k] = gy + b_lo * sstride[k] + (int64_t)(CB * cb + row) * E + col * 8;
      } else if (row < CB + HBK) {
        const int h = HBK * hb + (row - CB);
        sstride[k] = H * E;
        if (h < H) src[k] = xk + b_lo * sstride[k] + (int64_t)h * E + col * 8;
      } else {
        const int nn = g * DW_NG + (row - CB - HBK);
        sstride[k] = N * E;
        if (nn < N) src[k] = x0 + b_lo * sstride[k] + (int64_t)nn * E + col * 8;
      }
    }
  }
  uint4 stage[NV];
  auto fetch = [&](int64_t b) {     // called with consecutive b starting at b_lo
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      stage[k] = make_uint4(0, 0, 0, 0);
      if (src[k] != nullptr && b < b_hi) {
        stage[k] = *reinterpret_cast<const uint4*>(src[k]);
        src[k] += sstride[k];
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (threadIdx.x + 512 * k < TOTV) *reinterpret_cast<uint4*>(smem + (size_t)buf * BUF + doff[k]) = stage[k];
  };
  if (b_lo < b_hi) {
    fetch(b_lo);
    commit(0);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t b = b_lo; b < b_hi; ++b) {
    fetch(b + 1);
    const char* base = smem + (size_t)cur * BUF;
    const char* gy_s = base + (32 * cw) * RS;
    const char* xk_s = base + (CB + 16 * HW * hw) * RS;
    const char* x0_s = base + (CB + HBK + NW * nq) * RS;
#pragma unroll
    for (int ke = 0; ke < KE; ++ke) {
      const int eoff = (32 * ke + 8 * q) * 2;
      float xf[NW][8];
#pragma unroll
      for (int nn = 0; nn < NW; ++nn) Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(x0_s + nn * RS + eoff), xf[nn]);
      uint4 Bx[HW];
#pragma unroll
      for (int ht = 0; ht < HW; ++ht) Bx[ht] = *reinterpret_cast<const uint4*>(xk_s + (16 * ht + r) * RS + eoff);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float gf[8];
        Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(gy_s + (16 * ct + r) * RS + eoff), gf);
#pragma unroll
        for (int nn = 0; nn < NW; ++nn) {
          float p[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) p[k] = gf[k] * xf[nn][k];
          const uint4 a = Vec16<bf16_t>::pack(p);
#pragma unroll
          for (int ht = 0; ht < HW; ++ht)
            acc[nn][ct][ht] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, Bx[ht]), acc[nn][ct][ht], 0, 0, 0);
        }
      }
    }
    commit(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  float* out = dWpart + (size_t)split * C * N * H;
#pragma unroll
  for (int nn = 0; nn < NW; ++nn) {
    const int n = g * DW_NG + NW * nq + nn;
    if (n >= N) continue;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int ht = 0; ht < HW; ++ht) {
        const int h = HBK * hb + 16 * (HW * hw + ht) + r;
        if (h < H) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c = CB * cb + 32 * cw + 16 * ct + 4 * q + i;
            out[(size_t)c * N * H + (size_t)n * H + h] = acc[nn][ct][ht][i];
          }
        }
      }
  }
}

__global__ __launch_bounds__(256) void cin_reduce_partials_kernel(const float* __restrict__ part, int nparts, int64_t n,
                                                                  float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
    out[i] += s;
  }
}

// First layer (xk IS x0, weights folded onto h <= n): G[c,n,h] = sum_pix gy[c,pix] * x0[n,pix] * x0[h,pix] is
// symmetric in (n,h), so only the 16-h blocks hb <= n/16 of every field are computed and the reduce kernel mirrors
// them.  The roles of the operands are swapped relative to cin_dw_kernel: the A operand is gy itself (no VALU work,
// read from LDS once per k-step for all the fields of the wave) and the B operand is the product row
// Z[(n,h),pix] = x0[n,pix] * x0[h,pix] (8 multiplies + 4 packs per fragment, one fragment per field), so a fragment's
// VALU cost is spread over the CT = 8 channel tiles it is multiplied with: (8 + 20 PT) / (PT CT) = 2.8 VALU
// instructions per MFMA where the general kernel needs 6 at three h tiles.
// Wave = task (h block hb, PT = 3 consecutive fields >= 16 hb) x 128 channels; workgroup = 8 tasks of one channel
// block, sharing the staged gy tile and x0 rows of a sample.
template <int KE, int NV>
__global__ __launch_bounds__(512) void cin_dw_tri_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x0,
                                                         float* __restrict__ dWpart, int64_t B, int N, int C, int nsplit,
                                                         int wg_per_cb) {
  constexpr int PT = 3, CT = 8, CB = 16 * CT;
  constexpr int E = 32 * KE;
  constexpr int RS = E * 2 + 16;
  constexpr int VPR = E / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  const int NB = (N + 15) / 16;
  const int XR = 16 * NB + 1;                         // x0 rows staged: N real ones, zero rows up to 16 NB, one more zero row
  const int ROWS = CB + XR;
  const int BUF = ROWS * RS;
  const int TOTV = ROWS * VPR;
  int bid = blockIdx.x;
  const int part = bid % wg_per_cb; bid /= wg_per_cb;
  const int cbc = C / CB;
  const int cb = bid % cbc; bid /= cbc;
  const int split = bid;
  int task = part * 8 + wave, hb = 0, n_first = -1;
  for (; hb < NB; ++hb) {
    const int cnt = (N - 16 * hb + PT - 1) / PT;
    if (task < cnt) { n_first = 16 * hb + PT * task; break; }
    task -= cnt;
  }
  const bool active = n_first >= 0;
  if (!active) { hb = 0; n_first = 0; }
  const int64_t per = (B + nsplit - 1) / nsplit;
  const int64_t b_lo = split * per, b_hi = b_lo + per < B ? b_lo + per : B;
  f32x4 acc[PT][CT];
#pragma unroll
  for (int i = 0; i < PT; ++i)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[i][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16_t* src[NV];
  int sstride[NV];
  int doff[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = threadIdx.x + 512 * k;
    src[k] = nullptr; sstride[k] = 0; doff[k] = 0;
    if (v < TOTV) {
      const int row = v / VPR, col = v - row * VPR;
      doff[k] = row * RS + col * 16;
      if (row < CB) {
        sstride[k] = C * E;
        src[k] = gy + b_lo * sstride[k] + (int64_t)(CB * cb + row) * E + col * 8;
      } else if (row - CB < N) {
        sstride[k] = N * E;
        src[k] = x0 + b_lo * sstride[k] + (int64_t)(row - CB) * E + col * 8;
      }
    }
  }
  uint4 stage[NV];
  auto fetch = [&](int64_t b) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      stage[k] = make_uint4(0, 0, 0, 0);
      if (src[k] != nullptr && b < b_hi) {
        stage[k] = *reinterpret_cast<const uint4*>(src[k]);
        src[k] += sstride[k];
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (threadIdx.x + 512 * k < TOTV) *reinterpret_cast<uint4*>(smem + (size_t)buf * BUF + doff[k]) = stage[k];
  };
  if (b_lo < b_hi) {
    fetch(b_lo);
    commit(0);
  }
  __syncthreads();
  int nrow[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) nrow[i] = n_first + i < N ? n_first + i : XR - 1;     // past N: the zero row
  int cur = 0;
  for (int64_t b = b_lo; b < b_hi; ++b) {
    fetch(b + 1);
    const char* gy_s = smem + (size_t)cur * BUF;
    const char* x0_s = gy_s + CB * RS;
    if (active) {
#pragma unroll
      for (int ke = 0; ke < KE; ++ke) {
        const int eoff = (32 * ke + 8 * q) * 2;
        float hf[8];
        Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(x0_s + (16 * hb + r) * RS + eoff), hf);
        uint4 Bz[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) {
          float nf[8];
          Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(x0_s + nrow[i] * RS + eoff), nf);
#pragma unroll
          for (int k = 0; k < 8; ++k) nf[k] *= hf[k];
          Bz[i] = Vec16<bf16_t>::pack(nf);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const uint4 a = *reinterpret_cast<const uint4*>(gy_s + (16 * ct + r) * RS + eoff);
#pragma unroll
          for (int i = 0; i < PT; ++i)
            acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                                 __builtin_bit_cast(bf16x8, Bz[i]), acc[i][ct], 0, 0, 0);
        }
      }
    }
    commit(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (!active) return;
  float* out = dWpart + (size_t)split * C * N * N;
  const int h = 16 * hb + r;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int n = n_first + i;
    if (n >= N || h >= N) continue;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = CB * cb + 16 * ct + 4 * q + j;
        out[(size_t)c * N * N + (size_t)n * N + h] = acc[i][ct][j];
      }
  }
}

// out[c,n,h] += sum_p part[p][c][max(n,h)][min(n,h)]: every such entry lies in a computed block (min/16 <= max/16)
__global__ __launch_bounds__(256) void cin_reduce_partials_tri_kernel(const float* __restrict__ part, int nparts, int C,
                                                                      int N, float* __restrict__ out) {
  const int64_t n_all = (int64_t)C * N * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += (int64_t)gridDim.x * blockDim.x) {
    const int h = (int)(i % N), n = (int)((i / N) % N);
    const int64_t c = i / ((int64_t)N * N);
    const int64_t j = (c * N + (n > h ? n : h)) * N + (n > h ? h : n);
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n_all + j];
    out[i] += s;
  }
}

size_t cin_dw_workspace_bytes(int64_t B, int N, int H, int C) {
  return (size_t)64 * C * N * H * 4 + 256;     // up to 32 sample splits (64: the first-layer kernel)
}

static int cin_dw_tri(const void* gy, const void* x0, int64_t B, int N, int C, int E, float* dW, float* part, int NV,
                      hipStream_t s) {
  const int KE = E / 32;
  const int NB = (N + 15) / 16;
  int ntasks = 0;
  for (int hb = 0; hb < NB; ++hb) ntasks += (N - 16 * hb + 2) / 3;
  const int wg_per_cb = (ntasks + 7) / 8, cbc = C / 128;
  int nsplit = (int)std::max<int64_t>(1, 256 / (wg_per_cb * cbc));              // one workgroup per CU
  nsplit = (int)std::min<int64_t>(std::min(nsplit, 64), std::max<int64_t>(1, B / 8));
  const size_t lds = (size_t)2 * (128 + 16 * NB + 1) * (E * 2 + 16);
#define TRS_DWT(KE_, NV_)                                                                                        \
  do {                                                                                                           \
    auto kern = cin_dw_tri_kernel<KE_, NV_>;                                                                     \
    static size_t attr_lds = 0;                                                                                  \
    if (lds > 64 * 1024 && lds > attr_lds) {                                                                     \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return check_launch("cin_dw: LDS attribute");                                                            \
      attr_lds = lds;                                                                                            \
    }                                                                                                            \
    hipLaunchKernelGGL(kern, dim3(wg_per_cb * cbc * nsplit), dim3(512), lds, s, (const bf16_t*)gy,               \
                       (const bf16_t*)x0, part, B, N, C, nsplit, wg_per_cb);                                     \
  } while (0)
#define TRS_DWT_NV(KE_)               \
  do {                                \
    if (NV == 2) TRS_DWT(KE_, 2);     \
    else if (NV == 3) TRS_DWT(KE_, 3); \
    else TRS_DWT(KE_, 4);             \
  } while (0)
  if (KE == 1) TRS_DWT_NV(1);
  else TRS_DWT_NV(2);
#undef TRS_DWT_NV
#undef TRS_DWT
  const int64_t n = (int64_t)C * N * N;
  hipLaunchKernelGGL(cin_reduce_partials_tri_kernel, dim3((int)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s,
                     part, nsplit, C, N, dW);
  return check_launch("cin_dw (first layer)");
}

// gy (B,C,E), x0 (B,N,E), xk (B,H,E) contiguous bf16; dW (C, N*H) fp32 accumulated into.
// tri: xk is x0 (same pointer, H == N): the symmetric first-layer form, cin_dw_tri_kernel where it covers the shape
int cin_dw(const void* gy, const void* x0, const void* xk, int64_t B, int N, int H, int C, int E, int tri, float* dW,
           void* workspace, size_t ws_bytes, hipStream_t s, int64_t gy_bs = 0) {
  if (!(C == 64 || C == 128 || C == 256) || !(E == 32 || E == 64 || E == 128) || workspace == nullptr ||
      !aligned16(gy) || !aligned16(x0) || !aligned16(xk))
    return 1;
  if (gy_bs == 0) gy_bs = (int64_t)C * E;
  if (gy_bs < (int64_t)C * E || gy_bs % 8 != 0 || gy_bs >= ((int64_t)1 << 31) || (tri && gy_bs != (int64_t)C * E)) return 1;
  if (ws_bytes < cin_dw_workspace_bytes(B, N, H, C)) return fail(TRS_EWORKSPACE, "cin_dw: workspace too small");
  const int KE = E / 32;
  if (tri && xk == x0 && H == N && C % 128 == 0 && KE <= 2) {
    const int nv = ((128 + 16 * ((N + 15) / 16) + 1) * (E / 8) + 511) / 512;
    if (nv <= 4) return cin_dw_tri(gy, x0, B, N, C, E, dW, (float*)workspace, std::max(nv, 2), s);
  }
  const int tiles = (H + 15) / 16;
  // wide H (5..8 tiles per 128-h block): two fields x 32 channels x 8 h-tiles per wave; narrow H: four fields per wave,
  // all waves share the h range and split 128 channels; otherwise two wave groups split h
  const int hw_cap = KE == 4 ? 2 : 4;    // E = 128 stages more vectors per thread: keep the accumulators small
  int NW = 4, WH, HW;
  if (KE <= 2 && (tiles + 7) / 8 * 8 - tiles <= 3) {
    NW = 2; WH = 1; HW = 8;
  } else if (tiles <= hw_cap && C % 128 == 0) {
    WH = 1; HW = tiles;
  } else {
    WH = 2; HW = 1;
    int best = 1 << 30;
    for (int cand = hw_cap; cand >= 1; --cand) {
      const int cover = (tiles + 2 * cand - 1) / (2 * cand) * (2 * cand);
      if (cover < best) { best = cover; HW = cand; }
    }
  }
  const int CB = 32 * (DW_WAVES / ((DW_NG / NW) * WH)), HBK = 16 * HW * WH;
  const int ng = (N + DW_NG - 1) / DW_NG, cbc = C / CB, hbc = (H + HBK - 1) / HBK;
  int nsplit = (int)std::max<int64_t>(1, 256 / std::max(1, ng * cbc * hbc));   // one workgroup per CU
  nsplit = (int)std::min<int64_t>(nsplit, std::max<int64_t>(1, B / 8));
  nsplit = std::min(nsplit, 32);
  const int grid = ng * cbc * hbc * nsplit;
  const size_t lds = (size_t)2 * (CB + HBK + DW_NG) * (E * 2 + 16);
  float* part = (float*)workspace;
#define TRS_DW(NW_, WH_, HW_, KE_)                                                                               \
  do {                                                                                                           \
    auto kern = cin_dw_kernel<NW_, WH_, HW_, KE_>;                                                               \
    static size_t attr_lds = 0;                                                                                  \
    if (lds > 64 * 1024 && lds > attr_lds) {                                                                     \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return check_launch("cin_dw: LDS attribute");                                                            \
      attr_lds = lds;                                                                                            \
    }                                                                                                            \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, (const bf16_t*)gy, (const bf16_t*)x0, (const bf16_t*)xk, \
                       part, B, N, H, C, nsplit, (int)gy_bs);                                                    \
  } while (0)
#define TRS_DW_KE(WH_, HW_)                                                              \
  do {                                                                                   \
    if (KE == 1) TRS_DW(4, WH_, HW_, 1);                                                 \
    else if (KE == 2) TRS_DW(4, WH_, HW_, 2);                                            \
    else if constexpr (HW_ <= 2) TRS_DW(4, WH_, HW_, 4);      /* E = 128: hw_cap = 2 */  \
    else return fail(TRS_ESHAPE, "cin_dw: E = 128 runs at most 2 h tiles per wave");     \
  } while (0)
#define TRS_DW_HW(WH_)                         \
  do {                                         \
    if (HW == 1) TRS_DW_KE(WH_, 1);            \
    else if (HW == 2) TRS_DW_KE(WH_, 2);       \
    else if (HW == 3) TRS_DW_KE(WH_, 3);       \
    else TRS_DW_KE(WH_, 4);                    \
  } while (0)
  if (NW == 2) {
    if (KE == 1) TRS_DW(2, 1, 8, 1);
    else TRS_DW(2, 1, 8, 2);
  } else if (WH == 1) TRS_DW_HW(1);
  else TRS_DW_HW(2);
#undef TRS_DW_HW
#undef TRS_DW_KE
#undef TRS_DW
  const int64_t n = (int64_t)C * N * H;
  hipLaunchKernelGGL(cin_reduce_partials_kernel, dim3((int)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s,
                     part, nsplit, n, dW);
  return check_launch("cin_dw");
}

// channels-first entry points keep using the generic kernels for now
int cin_mfma_fwd(const void*, const void*, const void*, const void*, int64_t, int, int, int, int, void*, float*,
                 hipStream_t) { return 1; }
int cin_mfma_bwd(const void*, const void*, const void*, const void*, int64_t, int, int, int, int, float*, void*, void*,
                 int, hipStream_t) { return 1; }

}  // namespace trs

using namespace trs;

extern "C" size_t trs_cin_cl_workspace_bytes(int32_t N, int32_t H, int32_t C) {
  if (N <= 0 || H <= 0 || C <= 0) return 0;
  return cin_mfma_fwd_workspace_bytes(N, H, C);
}

extern "C" int trs_cin_cl_fwd(const void* x0T, int32_t ld0, const void* xkT, int32_t ldk, const void* Wc,
                              const void* bias, int64_t B, int32_t N, int32_t H, int32_t C, int32_t E, int32_t dtype,
                              int32_t tri, void* yT, void* workspace, size_t ws_bytes, trs_stream_t stream) {
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(x0T && xkT && Wc && yT, TRS_EINVAL, "cin_cl_fwd: NULL pointer");
  TRS_REQUIRE(B > 0 && N > 0 && H > 0 && C > 0 && E > 0, TRS_EINVAL, "cin_cl_fwd: bad size");
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "cin_cl_fwd: bf16 only (dtype %d)", dtype);
  TRS_REQUIRE(!tri || (N == H && x0T == xkT && ld0 == ldk), TRS_EINVAL, "cin_cl_fwd: tri needs xkT to be x0T (N %d, H %d)",
              N, H);
  const int rc = cin_cl_fwd(x0T, ld0, xkT, ldk, Wc, bias, B, N, H, C, E, tri != 0, yT, workspace, ws_bytes,
                            (hipStream_t)stream);
  if (rc == 1) return fail(TRS_ESHAPE, "cin_cl_fwd: shape not covered (need C%%32==0, E%%16==0, H<=256, padded rows)");
  return rc;
}

extern "C" size_t trs_cin_cl_bwd_data_workspace_bytes(int32_t N, int32_t H, int32_t C) {
  if (N <= 0 || H <= 0 || C <= 0) return 0;
  return cin_mfma_bwd_data_workspace_bytes(N, H, C);
}

extern "C" int trs_cin_cl_bwd_data(const void* x0T, int32_t ld0, const void* xkT, int32_t ldk, const void* gyT,
                                   const void* Wc, int64_t B, int32_t N, int32_t H, int32_t C, int32_t E,
                                   int32_t dtype, int32_t tri, void* dx0T, void* dxkT, int32_t ldo, void* workspace,
                                   size_t ws_bytes, trs_stream_t stream) {
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(x0T && xkT && gyT && Wc && dx0T && (dxkT || tri), TRS_EINVAL, "cin_cl_bwd_data: NULL pointer");
  TRS_REQUIRE(B > 0 && N > 0 && H > 0 && C > 0 && E > 0, TRS_EINVAL, "cin_cl_bwd_data: bad size");
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "cin_cl_bwd_data: bf16 only (dtype %d)", dtype);
  TRS_REQUIRE(!tri || (N == H && x0T == xkT && ld0 == ldk), TRS_EINVAL,
              "cin_cl_bwd_data: tri needs xkT to be x0T (N %d, H %d)", N, H);
  const int rc = cin_cl_bwd_data(x0T, ld0, xkT, ldk, gyT, Wc, B, N, H, C, E, tri != 0, dx0T, dxkT, ldo, workspace,
                                 ws_bytes, (hipStream_t)stream);
  if (rc == 1) return fail(TRS_ESHAPE, "cin_cl_bwd_data: shape not covered (C in {32,64,128,256}, E%%16==0)");
  return rc;
}

extern "C" size_t trs_cin_dw_workspace_bytes(int64_t B, int32_t N, int32_t H, int32_t C) {
  if (N <= 0 || H <= 0 || C <= 0) return 0;
  return cin_dw_workspace_bytes(B, N, H, C);
}

extern "C" int trs_cin_dw(const void* gy, const void* x0, const void* xk, int64_t B, int32_t N, int32_t H, int32_t C,
                          int32_t E, int32_t dtype, int32_t tri, float* dW, void* workspace, size_t ws_bytes,
                          trs_stream_t stream) {
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(gy && x0 && xk && dW, TRS_EINVAL, "cin_dw: NULL pointer");
  TRS_REQUIRE(B > 0 && N > 0 && H > 0 && C > 0 && E > 0, TRS_EINVAL, "cin_dw: bad size");
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "cin_dw: bf16 only (dtype %d)", dtype);
  TRS_REQUIRE(!tri || (xk == x0 && N == H), TRS_EINVAL, "cin_dw: tri needs xk to be x0 (N %d, H %d)", N, H);
  const int rc = cin_dw(gy, x0, xk, B, N, H, C, E, tri != 0, dW, workspace, ws_bytes, (hipStream_t)stream);
  if (rc == 1) return fail(TRS_ESHAPE, "cin_dw: shape not covered (C in {64,128,256}, E in {32,64,128})");
  return rc;
}

/* see include/trs_abi.h: the two gradients of the contraction when only the first C of the layer's Ct output channels can
 * carry a gradient */
extern "C" int trs_cin_cl_bwd_data_live(const void* x0T, int32_t ld0, const void* xkT, int32_t ldk, const void* gyT,
                                        int32_t ldg, const void* Wc, int64_t B, int32_t N, int32_t H, int32_t C, int32_t E,
                                        int32_t dtype, void* dx0T, void* dxkT, int32_t ldo, void* workspace,
                                        size_t ws_bytes, trs_stream_t stream) {
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(x0T && xkT && gyT && Wc && dx0T && dxkT, TRS_EINVAL, "cin_cl_bwd_data_live: NULL pointer");
  TRS_REQUIRE(B > 0 && N > 0 && H > 0 && C > 0 && E > 0 && ldg >= C, TRS_EINVAL, "cin_cl_bwd_data_live: bad size");
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "cin_cl_bwd_data_live: bf16 only (dtype %d)", dtype);
  const int rc = cin_cl_bwd_data(x0T, ld0, xkT, ldk, gyT, Wc, B, N, H, C, E, false, dx0T, dxkT, ldo, workspace, ws_bytes,
                                 (hipStream_t)stream, ldg);
  if (rc == 1) return fail(TRS_ESHAPE, "cin_cl_bwd_data_live: shape not covered (C in {32,64,128,256}, E%%16==0, ldg%%8==0)");
  return rc;
}

extern "C" int trs_cin_dw_live(const void* gy, int64_t gy_batch_stride, const void* x0, const void* xk, int64_t B, int32_t N,
                               int32_t H, int32_t C, int32_t E, int32_t dtype, float* dW, void* workspace, size_t ws_bytes,
                               trs_stream_t stream) {
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(gy && x0 && xk && dW, TRS_EINVAL, "cin_dw_live: NULL pointer");
  TRS_REQUIRE(B > 0 && N > 0 && H > 0 && C > 0 && E > 0 && gy_batch_stride >= (int64_t)C * E, TRS_EINVAL,
              "cin_dw_live: bad size");
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "cin_dw_live: bf16 only (dtype %d)", dtype);
  const int rc = cin_dw(gy, x0, xk, B, N, H, C, E, false, dW, workspace, ws_bytes, (hipStream_t)stream, gy_batch_stride);
  if (rc == 1) return fail(TRS_ESHAPE, "cin_dw_live: shape not covered (C in {64,128,256}, E in {32,64,128})");
  return rc;
}
