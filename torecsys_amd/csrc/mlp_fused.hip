// the per-field MLP of DeepAndCrossNetwork (models/ctr/deep_and_cross_network.py:71-87 applies
// MultilayerPerceptionLayer, layers/ctr/multilayer_perceptron.py:63-84, to every (sample, field) row of the (B,N,E)
// block: B*N = 2.5 M rows of 64 -> 400 -> 400 -> 400 -> 64) as ONE kernel per direction, bf16 on the matrix cores.
//
// A workgroup (8 waves) owns 128 rows at a time and walks the whole layer stack with the activations of those rows in
// LDS: a layer's input is read from LDS as MFMA B operands (16-byte reads of [row][k]), its output goes back to LDS
// from the accumulators as 16-byte row pieces (the W rows are fed in the permuted slot order of cross_mfma.hip, so a
// lane's 8 outputs of a column pair are 8 consecutive columns of its row) and becomes the next layer's input.  The
// weights (0.74 MB for the DCN stack: they do not fit in LDS) are pre-packed once per call into MFMA fragment order
// and streamed from L2 straight into registers: the output columns of a layer are split over the 8 waves, so every
// weight byte is loaded once per workgroup and no weight passes through LDS.  HBM sees the rows once per tensor:
//   forward : x read; every hidden activation written once (the weight-gradient GEMMs need them) + 1 ReLU sign bit each;
//   backward: the output gradient read; d(pre-activation) of every layer written once (again for the weight
//             gradients), the sign bits read (64 bytes per row and layer instead of a 832-byte activation row), column
//             sums (bias gradients) accumulated on chip, dx written.
// The unfused pipeline (hipBLASLt GEMMs + trs_relu_bwd_bias) moves ~16 GB of hidden activations per step and pads the
// 400-wide layers to 512 columns; here the widths are padded to 32 (416).  The weight gradients stay GEMMs with
// K = rows (torch.bmm split-K + trs_wgrad_finish) on the tensors this kernel writes.
#include <algorithm>

#include "trs_common.hpp"

namespace trs {

typedef __attribute__((ext_vector_type(8))) __bf16 mf_bf16x8;
typedef __attribute__((ext_vector_type(4))) float mf_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned mf_u32x4;

#ifndef TRS_MF_ROWS
#define TRS_MF_ROWS 128
#endif
#ifndef TRS_ROWS_GEMM_PLAIN_STORES
#define TRS_ROWS_GEMM_PLAIN_STORES 0      // 1: the wide input gradient written with cache-allocating stores (experiment)
#endif
#ifndef TRS_MF_GRID
#define TRS_MF_GRID 256
#endif
#ifndef TRS_MF_MINW
#define TRS_MF_MINW 2      // waves per SIMD the kernels are compiled for (2 = one 8-wave workgroup per CU)
#endif
constexpr int MF_ROWS = TRS_MF_ROWS;  // rows per workgroup pass
constexpr int MF_MT = MF_ROWS / 16;   // 16-row tiles
constexpr int MF_WAVES = 8;
constexpr int MF_MAXP = 2;            // 32-column pairs per wave: widths up to 8 * 2 * 32 = 512
constexpr int MF_MAXL = 8;
constexpr int MF_MASK_TILE = 64 * MF_WAVES * 16;   // mask bytes per 128-row pass and layer: 16 per lane (see mlp_fused_fwd_kernel)

// column of D-row slot m (= 4*q + i) of 16-column tile mt: the two tiles of a pair interleave 4-column groups, so that
// a lane's (tile 2p, tile 2p+1) outputs are the 8 consecutive columns 32p + 8q .. +7 of its row
__host__ __device__ __forceinline__ int mf_col_of_slot(int mt, int m) {
  return 32 * (mt >> 1) + 8 * (m >> 2) + 4 * (mt & 1) + (m & 3);
}

// Fragment order: frag (mt, ks) = 64 lanes x 8 bf16; lane (m = lane&15, q = lane>>4) holds
//   forward : W[col(mt, m)][32*ks + 8*q + j]        (out = W's rows, contraction over W's columns)
//   backward: W[32*ks + 8*q + j][col(mt, m)]        (out = W's columns, contraction over W's rows)
// zero outside the logical (out_f, in_f) matrix.  bias -> fp32, zero padded.
__global__ __launch_bounds__(256) void mlp_prepack_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ b,
                                                          int out_f, int in_f, int transpose, int NTt /* out tiles */,
                                                          int KS, bf16_t* __restrict__ Wf, float* __restrict__ bf,
                                                          int bias_n) {
  const int total = NTt * KS * 64;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int lane = t & 63;
    const int f = t >> 6;
    const int ks = f % KS, mt = f / KS;
    const int oc = mf_col_of_slot(mt, lane & 15);
    const int k0 = 32 * ks + 8 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      bf16_t v{0};
      if (!transpose) {
        if (oc < out_f && k < in_f) v = W[(size_t)oc * in_f + k];
      } else {
        if (oc < in_f && k < out_f) v = W[(size_t)k * in_f + oc];
      }
      Wf[(size_t)t * 8 + j] = v;
    }
  }
  if (bf != nullptr)
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < bias_n; t += gridDim.x * blockDim.x)
      bf[t] = (b != nullptr && t < out_f) ? to_f32(b[t]) : 0.f;
}

// all layers of a stack in one launch (blockIdx.y = layer): six 8-us launches per step were 100 us of a 1.5 ms DeepFM step
struct MlpPackJob {
  const bf16_t* W;
  const bf16_t* b;
  bf16_t* Wf;
  float* bf;
  int out_f, in_f, NTt, KS, bias_n;
  int transpose;
};
constexpr int MF_MAXJOBS = 2 * MF_MAXL + 1;      // a stack's forward and backward copies + the layer in front of it (trs_mlp_pack_branch)
struct MlpPackArgs {
  MlpPackJob job[MF_MAXJOBS];
};
__global__ __launch_bounds__(256) void mlp_prepack_many_kernel(MlpPackArgs a) {
  const MlpPackJob& j = a.job[blockIdx.y];
  const int total = j.NTt * j.KS * 64;
  const int transpose = j.transpose;
  const bool vec = !transpose && (j.in_f & 7) == 0 && (reinterpret_cast<uintptr_t>(j.W) & 15u) == 0;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int lane = t & 63;
    const int f = t >> 6;
    const int ks = f % j.KS, mt = f / j.KS;
    const int oc = mf_col_of_slot(mt, lane & 15);
    const int k0 = 32 * ks + 8 * (lane >> 4);
    uint4* dst = reinterpret_cast<uint4*>(j.Wf + (size_t)t * 8);
    if (vec) {       // 8 consecutive k of one weight row: one 16-byte load, one 16-byte store
      *dst = (oc < j.out_f && k0 + 8 <= j.in_f) ? *reinterpret_cast<const uint4*>(j.W + (size_t)oc * j.in_f + k0)
                                                 : make_uint4(0, 0, 0, 0);
      continue;
    }
    uint16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + e;
      v[e] = 0;
      if (!transpose) {
        if (oc < j.out_f && k < j.in_f) v[e] = j.W[(size_t)oc * j.in_f + k].v;
      } else {
        if (oc < j.in_f && k < j.out_f) v[e] = j.W[(size_t)k * j.in_f + oc].v;
      }
    }
    *dst = make_uint4(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16), v[4] | ((uint32_t)v[5] << 16),
                      v[6] | ((uint32_t)v[7] << 16));
  }
  if (j.bf != nullptr)
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < j.bias_n; t += gridDim.x * blockDim.x)
      j.bf[t] = (j.b != nullptr && t < j.out_f) ? to_f32(j.b[t]) : 0.f;
}

struct MlpStep {
  const uint4* wf;    // fragment-order weights of this step
  const float* bias;  // fp32, padded (forward) or null
  void* out;          // global output of the step (rows x out_stride elements), may be null
  uint8_t* mask;      // forward: ReLU sign bits written for this step's output; backward: applied to this step's output
  float* colsum;      // backward: partial column sums of this step's INPUT, [gridDim.x][K] (may be null)
  int K, N;           // padded contraction / output widths (multiples of 32)
  int out_stride;     // elements per global output row
  int relu;           // forward: ReLU on the output
};
struct MlpArgs {
  MlpStep step[MF_MAXL];
  int nsteps;
  const void* in;     // (rows x in_stride) bf16
  int in_stride;      // elements per input row (= its logical width, a multiple of 8)
  int nbias;          // forward: total padded bias entries (copied to LDS)
  int64_t rows;
  int act_str;        // bytes per LDS activation row
  uint8_t* mask_in;   // sign bits [input > 0] of the stack's INPUT rows (an upstream ReLU's output), in the item order of
                      // the backward's last step: written by the forward, applied by the backward; may be null
  float* colsum_in;   // backward, with mask_in: partial column sums of the masked input gradient, [gridDim.x][step N]
  int ro_masks;       // backward: the sign bits are in the ROW-OWNER kernels' layout (mlp_ro.hpp), see mlp_fused_bwd_kernel
};

// tiles of the step owned by this wave: wide outputs split the 32-column pairs over the 8 waves (all 8 row tiles
// each); narrow outputs (<= 4 pairs) also split the row tiles so that no wave idles
struct MlpShare {
  int npw;              // pairs of this wave (0..MF_MAXP)
  int pair[MF_MAXP];
  int mt0, mcnt;        // row tiles mt0 .. mt0+mcnt-1
  int mt1, mc1;         // mc1 > 0: the wave's LAST pair covers only row tiles mt1 .. mt1+mc1-1 (a pair shared by 4 or 2 waves)
};
__device__ __forceinline__ MlpShare mlp_share(int N, int wave) {
  MlpShare s;
  s.mt1 = 0;
  s.mc1 = 0;
  const int npairs = N >> 5;
  if (npairs > 4) {
    s.mt0 = 0;
    s.mcnt = MF_MT;
    s.npw = 0;
#pragma unroll
    for (int i = 0; i < MF_MAXP; ++i) {
      s.pair[i] = wave + MF_WAVES * i;
      if (s.pair[i] < npairs) s.npw = i + 1;
    }
    // Waves w and w+4 share a SIMD: pairs 8, 9, ... given whole to waves 0, 1, ... load the SIMDs evenly only in groups of
    // four.  One or two pairs left over (13 pairs = 416 columns: pair 12) would sit on one or two SIMDs -- 4 pairs against
    // 3 on the others, the wave at the barrier for a third of the GEMM -- so the four waves of that group share them by
    // row tiles instead: 2 of the 8 row tiles each (one pair left over) or 4 each (two).
    const int extra = npairs - MF_WAVES, left = extra & 3, g0 = extra & ~3;
    if (MF_MT == 8 && extra > 0 && extra < MF_WAVES && (left == 1 || left == 2) && wave >= g0 && wave < g0 + 4) {
      const int c = wave - g0;
      s.npw = 2;
      s.pair[1] = MF_WAVES + g0 + (left == 1 ? 0 : (c >> 1));
      s.mc1 = left == 1 ? 2 : 4;
      s.mt1 = left == 1 ? 2 * c : 4 * (c & 1);
    }
  } else {
    const int ng = npairs <= 1 ? 1 : (npairs == 2 ? 2 : 4);     // pair groups; MF_WAVES / ng row groups
    const int mg = MF_WAVES / ng < MF_MT ? MF_WAVES / ng : MF_MT;
    s.mcnt = MF_MT / mg;
    s.mt0 = (wave / ng) * s.mcnt;
    s.pair[0] = wave % ng;
    s.pair[1] = 0;
    s.npw = (s.pair[0] < npairs && wave / ng < mg) ? 1 : 0;
  }
  return s;
}

// acc[mi][2*pi + h] (+)= sum_k W-frag(pair pi, half h, ks) x act rows of tile mt0+mi, for MCNT row tiles and NPW column
// pairs known at compile time (runtime bounds put a scalar branch in front of every MFMA).
// The weight fragments come from L2: a k-step's fragments are requested at the top of the k-step before it and waited
// for at that k-step's bottom; the activations are read from LDS per k-step in two halves (16 registers of B operands
// instead of 64).
// MC1 > 0: the last pair covers MC1 row tiles from sh.mt1 (its accumulators are acc[0 .. MC1-1][2 * (NPW-1) + h]).
template <int MCNT, int NPW, int MC1 = 0>
__device__ __forceinline__ void mlp_gemm_t(const char* act, int act_str, const uint4* __restrict__ wf, int K,
                                           const MlpShare& sh, int lane, mf_f32x4 (&acc)[MF_MT][2 * MF_MAXP]) {
  const int KS = K >> 5;
  const int r = lane & 15, q = lane >> 4;
  const char* arow = act + (sh.mt0 * 16 + r) * act_str + q * 16;
  const char* arow1 = act + (sh.mt1 * 16 + r) * act_str + q * 16;
  constexpr int TF = 2 * (NPW - (MC1 > 0 ? 1 : 0));      // fragments of the pairs that cover all MCNT row tiles
  const uint4* wbase[NPW];
#pragma unroll
  for (int pi = 0; pi < NPW; ++pi) wbase[pi] = wf + ((size_t)(2 * sh.pair[pi]) * KS) * 64 + lane;
  mf_u32x4 A[2 * NPW], An[2 * NPW];
#define TRS_MF_FETCH(dst, ks_)                                                                                      \
  _Pragma("unroll") for (int pi = 0; pi < NPW; ++pi) {                                                             \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[2 * pi]) : "v"(wbase[pi] + (size_t)(ks_) * 64));     \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[2 * pi + 1]) : "v"(wbase[pi] + (size_t)(KS + (ks_)) * 64)); \
  }
#define TRS_MF_COMMIT(dst)                                \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        \
  _Pragma("unroll") for (int t = 0; t < 2 * NPW; ++t) asm volatile("" : "+v"(dst[t]));
  TRS_MF_FETCH(A, 0)
  TRS_MF_COMMIT(A)
  {
    constexpr int HALF = MCNT >= 2 ? MCNT / 2 : 1;
    // The fragment loads are issued by hand at the top of a k-step and waited for at its bottom: written as ordinary
    // loads the compiler places them at the END of the previous k-step's body and waits vmcnt(0) at the top of the
    // next one, i.e. every k-step starts with a full L2 round trip (ISA of the first version of this loop).
    // two k-steps per iteration with the two fragment sets trading places: no register copies between the steps
    // (16 v_mov per k-step beside 32 MFMAs: the matrix pipe and the VALU share the SIMD's issue port)
    auto kstep = [&](int ks, mf_u32x4 (&cur)[2 * NPW], mf_u32x4 (&nxt)[2 * NPW]) {
      const int kn = ks + 1 < KS ? ks + 1 : ks;
      TRS_MF_FETCH(nxt, kn)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m0 = 0; m0 < MCNT; m0 += HALF) {
        uint4 Bh[HALF];
#pragma unroll
        for (int mi = 0; mi < HALF; ++mi) Bh[mi] = *reinterpret_cast<const uint4*>(arow + (m0 + mi) * 16 * act_str + ks * 64);
#pragma unroll
        for (int t = 0; t < TF; ++t)
#pragma unroll
          for (int mi = 0; mi < HALF; ++mi)
            acc[m0 + mi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(mf_bf16x8, cur[t]), __builtin_bit_cast(mf_bf16x8, Bh[mi]), acc[m0 + mi][t], 0, 0, 0);
      }
      if constexpr (MC1 > 0) {
        uint4 Bx[MC1];
#pragma unroll
        for (int mi = 0; mi < MC1; ++mi) Bx[mi] = *reinterpret_cast<const uint4*>(arow1 + mi * 16 * act_str + ks * 64);
#pragma unroll
        for (int t = TF; t < TF + 2; ++t)
#pragma unroll
          for (int mi = 0; mi < MC1; ++mi)
            acc[mi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(mf_bf16x8, cur[t]), __builtin_bit_cast(mf_bf16x8, Bx[mi]), acc[mi][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      TRS_MF_COMMIT(nxt)
    };
    int ks = 0;
    for (; ks + 1 < KS; ks += 2) {
      kstep(ks, A, An);
      kstep(ks + 1, An, A);
    }
    if (ks < KS) kstep(ks, A, An);
  }
#undef TRS_MF_FETCH
#undef TRS_MF_COMMIT
}

// The same k-loop with the number of k-steps known at compile time: everything is unrolled, so every fragment set is its
// own value (no loop-carried register roles, nothing for the compiler to copy while a load is in flight) and the loads
// can run TWO k-steps ahead -- a k-step (32 MFMAs of this wave, 64 with its SIMD neighbour) is about one L2 round trip,
// one step of distance leaves the wave waiting at most steps.  Only fragments that will be multiplied are requested.
template <int MCNT, int NPW, int KS, int MC1 = 0>
__device__ __forceinline__ void mlp_gemm_static(const char* act, int act_str, const uint4* __restrict__ wf,
                                                const MlpShare& sh, int lane, mf_f32x4 (&acc)[MF_MT][2 * MF_MAXP]) {
  const int r = lane & 15, q = lane >> 4;
  const char* arow = act + (sh.mt0 * 16 + r) * act_str + q * 16;
  const char* arow1 = act + (sh.mt1 * 16 + r) * act_str + q * 16;
  constexpr int TF = 2 * (NPW - (MC1 > 0 ? 1 : 0));      // fragments of the pairs that cover all MCNT row tiles
  const uint4* wbase[NPW];
#pragma unroll
  for (int pi = 0; pi < NPW; ++pi) wbase[pi] = wf + ((size_t)(2 * sh.pair[pi]) * KS) * 64 + lane;
  mf_u32x4 F[KS][2 * NPW];
  auto fetch = [&](int ks) {
#pragma unroll
    for (int pi = 0; pi < NPW; ++pi) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(F[ks][2 * pi]) : "v"(wbase[pi] + (size_t)ks * 64));
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(F[ks][2 * pi + 1]) : "v"(wbase[pi] + (size_t)(KS + ks) * 64));
    }
  };
  // the activation fragments are read two row tiles at a time, the NEXT two (of this k-step or of the next one) while the
  // current two are multiplied: the rows-GEMM's counters showed half the wave cycles waiting for an issuable instruction
  constexpr int RT = MCNT >= 2 ? 2 : 1;               // row tiles per read group
  constexpr int NG = MCNT / RT;                        // groups per k-step
  uint4 B[KS * NG][RT];
  auto read_group = [&](int g) {                       // g = ks * NG + group
#pragma unroll
    for (int mi = 0; mi < RT; ++mi)
      B[g][mi] = *reinterpret_cast<const uint4*>(arow + ((g % NG) * RT + mi) * 16 * act_str + (g / NG) * 64);
  };
  fetch(0);
  if (KS > 1) fetch(1);
  read_group(0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");      // the next step's set may stay in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2 * NPW; ++t) asm volatile("" : "+v"(F[ks][t]));
    if (ks + 2 < KS) fetch(ks + 2);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int g = ks * NG + gi;
      if (g + 1 < KS * NG) read_group(g + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TF; ++t)
#pragma unroll
        for (int mi = 0; mi < RT; ++mi)
          acc[gi * RT + mi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(mf_bf16x8, F[ks][t]), __builtin_bit_cast(mf_bf16x8, B[g][mi]), acc[gi * RT + mi][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MC1 > 0) {      // the pair shared by row tiles: MC1 row tiles from sh.mt1
      uint4 Bx[MC1];
#pragma unroll
      for (int mi = 0; mi < MC1; ++mi) Bx[mi] = *reinterpret_cast<const uint4*>(arow1 + mi * 16 * act_str + ks * 64);
#pragma unroll
      for (int t = TF; t < TF + 2; ++t)
#pragma unroll
        for (int mi = 0; mi < MC1; ++mi)
          acc[mi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(mf_bf16x8, F[ks][t]), __builtin_bit_cast(mf_bf16x8, Bx[mi]), acc[mi][t], 0, 0, 0);
    }
  }
}

// run ``body.template operator()<MCNT, NPW>()`` for this wave's share (wave-uniform dispatch)
template <typename F>
__device__ __forceinline__ void mlp_dispatch(const MlpShare& sh, F&& body) {
  if (sh.npw == 0) return;
  if (sh.mcnt == MF_MT) {
    if (sh.npw == 2) {
      if (sh.mc1 == 0) body.template operator()<MF_MT, 2, 0>();
      else if (sh.mc1 == 2) body.template operator()<MF_MT, 2, 2>();
      else body.template operator()<MF_MT, 2, 4>();
    } else {
      body.template operator()<MF_MT, 1, 0>();
    }
  } else if (sh.mcnt == 4) {
    body.template operator()<4, 1, 0>();
  } else if (sh.mcnt == 2) {
    body.template operator()<2, 1, 0>();
  } else {
    body.template operator()<1, 1, 0>();
  }
}

// global rows (row0 + r, 0 .. in_stride) -> LDS rows, zero-filled up to ``ncols`` columns and past the last row
__device__ __forceinline__ void mlp_load_in(char* act, int act_str, const void* in, int in_stride, int ncols, int64_t row0,
                                            int64_t rows) {
  const int cpr = ncols >> 3, cin = in_stride >> 3;
  for (int t = threadIdx.x; t < MF_ROWS * cpr; t += blockDim.x) {
    const int r = t / cpr, c = t - r * cpr;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row0 + r < rows && c < cin)
      v = *(reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(in) + (row0 + r) * in_stride) + c);
    *reinterpret_cast<uint4*>(act + r * act_str + c * 16) = v;
  }
}

// ------------------------------------------------------------------------------------------------ forward
// workgroup barrier for the LDS hand-offs inside a pass: __syncthreads() also waits vmcnt(0), i.e. for the global stores
// in flight, which nothing here depends on
// Sign bits of the 8 values of an item as one byte: value j sits at bit (j >> 1) + 4 * (j & 1) -- the order in which the
// packed [x > 0] words of the forward (bit 0 / bit 16 of word k = values 2k / 2k+1, shifted left by k and ORed) fold into it.
__device__ __forceinline__ unsigned mf_mask_byte(unsigned m) { return (m & 0xFu) | ((m >> 12) & 0xF0u); }
__device__ __forceinline__ int mf_mask_bit(int j) { return (j >> 1) + 4 * (j & 1); }
#define MF_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__global__ __launch_bounds__(64 * MF_WAVES, TRS_MF_MINW) void mlp_fused_fwd_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act = smem;                                          // [MF_ROWS][act_str]
  float* bias_s = reinterpret_cast<float*>(act + MF_ROWS * a.act_str);      // all layers' padded biases, back to back
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  for (int i = threadIdx.x; i < a.nbias; i += blockDim.x) bias_s[i] = a.step[0].bias[i];
  const int64_t ntiles = (a.rows + MF_ROWS - 1) / MF_ROWS;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * MF_ROWS;
    mlp_load_in(act, a.act_str, a.in, a.in_stride, a.step[0].K, row0, a.rows);
    MF_BAR();
    if (a.mask_in != nullptr) {
      // the input rows are the output of a ReLU in front of this stack: their sign bits, taken from LDS in the item order
      // in which the backward's last step (output width = this input width) hands the same lane its 8 columns
      const MlpShare shi = mlp_share(a.step[0].K, wave);
      unsigned mb[4] = {0u, 0u, 0u, 0u};
      const char* li0 = act + (shi.mt0 * 16 + r) * a.act_str + 16 * q;
      mlp_dispatch(shi, [&]<int MCNT, int NPW, int MC1>() {
#pragma unroll
        for (int pi = 0; pi < NPW; ++pi)
#pragma unroll
          for (int mi = 0; mi < MCNT; ++mi) {
            const bool part = MC1 > 0 && pi == NPW - 1;      // the pair this wave shares by row tiles (mlp_share)
            if (part && mi >= MC1) continue;
            const int rt = part ? shi.mt1 - shi.mt0 + mi : mi;
            const uint4 u = *reinterpret_cast<const uint4*>(li0 + rt * 16 * a.act_str + shi.pair[pi] * 64);
            const unsigned w[4] = {u.x, u.y, u.z, u.w};
            unsigned m = 0;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
              unsigned x, t;
              asm("v_pk_max_i16 %0, %1, 0" : "=v"(x) : "v"(w[k2]));
              asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(x), "s"(0x00010001u));
              m |= t << k2;
            }
            const unsigned bits = mf_mask_byte(m);
            mb[(pi * MCNT + mi) >> 2] |= bits << (8 * ((pi * MCNT + mi) & 3));
          }
      });
      store_stream(reinterpret_cast<uint4*>(a.mask_in + tile * MF_MASK_TILE) + threadIdx.x,
                   make_uint4(mb[0], mb[1], mb[2], mb[3]));
    }
    int boff = 0;
    for (int l = 0; l < a.nsteps; ++l) {
      const MlpStep st = a.step[l];
      const MlpShare sh = mlp_share(st.N, wave);
      mf_f32x4 acc[MF_MT][2 * MF_MAXP];
#pragma unroll
      for (int pi = 0; pi < MF_MAXP; ++pi)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          mf_f32x4 init = mf_f32x4{0.f, 0.f, 0.f, 0.f};
          if (pi < sh.npw) {
            const float4 bv = *reinterpret_cast<const float4*>(bias_s + boff + 32 * sh.pair[pi] + 8 * q + 4 * h);
            init = mf_f32x4{bv.x, bv.y, bv.z, bv.w};
          }
#pragma unroll
          for (int mi = 0; mi < MF_MT; ++mi) acc[mi][2 * pi + h] = init;
        }
      mlp_dispatch(sh, [&]<int MCNT, int NPW, int MC1>() {
        if (MCNT == MF_MT && st.K == 416) mlp_gemm_static<MCNT, NPW, 13, MC1>(act, a.act_str, st.wf, sh, lane, acc);
        else mlp_gemm_t<MCNT, NPW, MC1>(act, a.act_str, st.wf, st.K, sh, lane, acc);
      });
      const bool last = l + 1 == a.nsteps;
      const int out_cols = last ? st.out_stride : st.N;
      MF_BAR();      // every wave is done reading the layer's input (after the last layer: LDS is free for the next pass)
      // ReLU sign bits: a lane owns 8 columns of one row in each of its <= 16 (column pair, row tile) items, i.e. one
      // byte per item and 16 bytes per layer.  They go to global memory as that one vector, in (pass, thread) order --
      // the backward kernel gives the same lane the same items.  (Staging them through LDS in (row, column) order cost
      // ~4 k LDS cycles of bank conflicts per layer and pass: byte writes 64 bytes apart.)
      unsigned mbits[4] = {0u, 0u, 0u, 0u};
      // addresses of the lane's items: one base per layer, wave-uniform offsets per item (a 64-bit row * stride product
      // per item was a sixth of the epilogue's instructions); rows past the end of the tensor fail ``16 * mi < left``
      const int lrow = sh.mt0 * 16 + r;
      char* lds0 = act + lrow * a.act_str + 16 * q;
      bf16_t* out0 = reinterpret_cast<bf16_t*>(st.out) + (row0 + lrow) * st.out_stride + 8 * q;
      const int64_t left64 = a.rows - row0 - lrow;
      const int left = st.out == nullptr ? 0 : (left64 > MF_ROWS ? MF_ROWS : (int)left64);
      mlp_dispatch(sh, [&]<int MCNT, int NPW, int MC1>() {
#pragma unroll
        for (int pi = 0; pi < NPW; ++pi) {
          const bool colok = 32 * sh.pair[pi] + 8 * q < out_cols;
#pragma unroll
          for (int mi = 0; mi < MCNT; ++mi) {
            const bool part = MC1 > 0 && pi == NPW - 1;      // the pair this wave shares by row tiles (mlp_share)
            if (part && mi >= MC1) continue;
            const int rt = part ? sh.mt1 - sh.mt0 + mi : mi;
            // round to bf16 first (one convert per two values), then ReLU and the sign bits on the PACKED words: as 16-bit
            // integers negative floats (and -0.0) are negative, so max(x, 0) is ReLU and min(x, 1) is [x > 0], two values per
            // instruction (on the fp32 values it took three instructions per value: 48 VALU per item, now 36)
            unsigned w[4];
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2)
              w[k2] = f32x2_to_bf16x2_bits(acc[mi][2 * pi + (k2 >> 1)][2 * (k2 & 1)], acc[mi][2 * pi + (k2 >> 1)][2 * (k2 & 1) + 1]);
            unsigned bits = 0;
            if (st.relu) {
              unsigned m = 0;
#pragma unroll
              for (int k2 = 0; k2 < 4; ++k2) {
                unsigned t;
                asm("v_pk_max_i16 %0, %1, 0" : "=v"(w[k2]) : "v"(w[k2]));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(w[k2]), "s"(0x00010001u));
                m |= t << k2;
              }
              bits = mf_mask_byte(m);
            }
            const uint4 pk = make_uint4(w[0], w[1], w[2], w[3]);
            if (!last) *reinterpret_cast<uint4*>(lds0 + rt * 16 * a.act_str + sh.pair[pi] * 64) = pk;      // the next layer's input
            // the step's global output (hidden activations kept for the weight gradients / the result) leaves from the
            // registers: 16 bytes per lane, 64 contiguous bytes per row and wave.  (Global stores cost ~64 issue cycles
            // per wave instruction wherever they are placed -- 0.9 of the kernel's 3.3 ms; spreading them over the next
            // layer's k-steps from LDS, one per k-step with a counted vmcnt, measured the same.)
            if (colok && 16 * rt < left)
              store_stream(reinterpret_cast<uint4*>(out0 + (size_t)rt * 16 * st.out_stride + 32 * sh.pair[pi]), pk);
            mbits[(pi * MCNT + mi) >> 2] |= bits << (8 * ((pi * MCNT + mi) & 3));
          }
        }
      });
      if (!last) MF_BAR();
      if (st.mask != nullptr && st.relu)
        store_stream(reinterpret_cast<uint4*>(st.mask + tile * MF_MASK_TILE) + threadIdx.x,
                     make_uint4(mbits[0], mbits[1], mbits[2], mbits[3]));
      boff += st.N;
    }
  }
}

// ------------------------------------------------------------------------------------------------ one wide layer
// y[rows, N] = x[rows, :K] @ W[K, N] for a short K (<= 512: a 128-row tile of x stays in LDS) and a wide N -- the input
// gradient of the first layer of a deep branch (K = 400 hidden units, N = 2496 embedding columns), where the library's
// 256-wide macro tiles pad K to 512 and run at 0.6 PFLOP/s useful.  The same k-loop as the fused kernels (fragments of
// W from L2 one k-step ahead, activation rows from LDS), a wave walks column pairs wave, wave + 8, ... two at a time;
// x is read once, nothing but the bf16 result is written, and there is no barrier between a wave's rounds.
struct RowsGemmArgs {
  const void* in;      // (rows x in_stride) bf16
  const uint4* wf;     // fragment-order W (mlp_prepack_many_kernel, transpose = 1)
  void* out;           // (rows x out_stride) bf16
  int64_t rows;
  int in_stride, K, N, out_cols, out_stride, act_str;
};
__global__ __launch_bounds__(64 * MF_WAVES, TRS_MF_MINW) void mlp_rows_gemm_kernel(RowsGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act = smem;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  const int npairs = a.N >> 5;
  const int64_t ntiles = (a.rows + MF_ROWS - 1) / MF_ROWS;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * MF_ROWS;
    MF_BAR();                                            // the previous tile's rows are no longer read
    mlp_load_in(act, a.act_str, a.in, a.in_stride, a.K, row0, a.rows);
    MF_BAR();
    bf16_t* out0 = reinterpret_cast<bf16_t*>(a.out) + (row0 + r) * a.out_stride + 8 * q;
    const int64_t left64 = a.rows - row0 - r;
    const int left = left64 > MF_ROWS ? MF_ROWS : (int)left64;
    for (int p0 = wave; p0 < npairs; p0 += 2 * MF_WAVES) {
      MlpShare sh;
      sh.mt0 = 0;
      sh.mcnt = MF_MT;
      sh.pair[0] = p0;
      sh.pair[1] = p0 + MF_WAVES;
      sh.npw = sh.pair[1] < npairs ? 2 : 1;
      sh.mt1 = 0;
      sh.mc1 = 0;
      mf_f32x4 acc[MF_MT][2 * MF_MAXP];
#pragma unroll
      for (int mi = 0; mi < MF_MT; ++mi)
#pragma unroll
        for (int t = 0; t < 2 * MF_MAXP; ++t) acc[mi][t] = mf_f32x4{0.f, 0.f, 0.f, 0.f};
      auto body = [&]<int NPW>() {
        if (a.K == 416) mlp_gemm_static<MF_MT, NPW, 13>(act, a.act_str, a.wf, sh, lane, acc);      // 400 hidden units
        else if (a.K == 512) mlp_gemm_static<MF_MT, NPW, 16>(act, a.act_str, a.wf, sh, lane, acc);
        else mlp_gemm_t<MF_MT, NPW>(act, a.act_str, a.wf, a.K, sh, lane, acc);
#pragma unroll
        for (int pi = 0; pi < NPW; ++pi) {
          const bool colok = 32 * sh.pair[pi] + 8 * q < a.out_cols;
#pragma unroll
          for (int mi = 0; mi < MF_MT; ++mi) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              v[i] = acc[mi][2 * pi][i];
              v[4 + i] = acc[mi][2 * pi + 1][i];
            }
            if (colok && 16 * mi < left) {
#if TRS_ROWS_GEMM_PLAIN_STORES
              *reinterpret_cast<uint4*>(out0 + (size_t)mi * 16 * a.out_stride + 32 * sh.pair[pi]) = Vec16<bf16_t>::pack(v);
#else
              store_stream(reinterpret_cast<uint4*>(out0 + (size_t)mi * 16 * a.out_stride + 32 * sh.pair[pi]),
                           Vec16<bf16_t>::pack(v));
#endif
            }
          }
        }
      };
      if (sh.npw == 2) body.template operator()<2>();
      else body.template operator()<1>();
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward (data)
// step s works on layer l = L-1-s: input = d(pre-activation of layer l) (rows x N_l) in LDS, output = d(input of layer l)
// = d(output of layer l-1), masked by layer l-1's ReLU mask into d(pre-activation of layer l-1).
template <bool RO_MASKS>
__global__ __launch_bounds__(64 * MF_WAVES, TRS_MF_MINW) void mlp_fused_bwd_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act = smem;
  float* scratch = reinterpret_cast<float*>(act + MF_ROWS * a.act_str);              // [8 row slices][512]
  float* csum = scratch + 8 * 512;                                                    // [nsteps + 1][512] running column sums
  // sign bits written by the row-owner forward (TRS_MLP_FAMILY_MIXED): a 128-row tile's share of a layer is four 2 KB
  // pieces (one per group of four 32-column chunks) -- one 16-byte load per thread into this 8 KB stage, picked apart per
  // item in the epilogue.  Layout (mlp_ro.hpp): [pass of 256 rows][chunk / 4][slot = 2 (64 (row / 64) + (row % 32) + 32 g)
  // + (row / 32) % 2][chunk % 4] 16-bit words; bit 4 h + k2 = column 32 chunk + 16 h + 8 g + 2 k2, bit 8 + 4 h + k2 the one
  // after it.  This kernel's byte for (row, pair, q) -- bit j/2 + 4 (j & 1) = column 32 pair + 8 q + j -- is the two
  // nibbles h = q / 2 of the word of (row, chunk = pair, g = q & 1).
  // The stage ALIASES the column-sum scratch (a 512-wide stack leaves no 8 KB of LDS free): written once the step's partial
  // sums have been folded, between two extra barriers.
  unsigned short* mstage = reinterpret_cast<unsigned short*>(scratch);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  for (int i = threadIdx.x; i < (a.nsteps + 1) * 512; i += blockDim.x) csum[i] = 0.f;
  const int64_t ntiles = (a.rows + MF_ROWS - 1) / MF_ROWS;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * MF_ROWS;
    mlp_load_in(act, a.act_str, a.in, a.in_stride, a.step[0].K, row0, a.rows);
    MF_BAR();
    for (int s = 0; s < a.nsteps; ++s) {
      const MlpStep st = a.step[s];
      const MlpShare sh = mlp_share(st.N, wave);
      // the ReLU sign bits of this lane's outputs (the forward's vector for this pass and thread), in flight during the GEMM
      uint4 mraw = make_uint4(0, 0, 0, 0);
      if (st.mask != nullptr) {
        if constexpr (RO_MASKS)      // piece threadIdx.x / 128 (= chunk group), 16 bytes of this tile's 2 KB of it
          mraw = *reinterpret_cast<const uint4*>(st.mask + (tile >> 1) * 16384 + (threadIdx.x >> 7) * 4096 + (tile & 1) * 2048 +
                                                 (threadIdx.x & 127) * 16);
        else
          mraw = *(reinterpret_cast<const uint4*>(st.mask + tile * MF_MASK_TILE) + threadIdx.x);
      }
      const unsigned mword[4] = {mraw.x, mraw.y, mraw.z, mraw.w};
      mf_f32x4 acc[MF_MT][2 * MF_MAXP];
#pragma unroll
      for (int mi = 0; mi < MF_MT; ++mi)
#pragma unroll
        for (int t = 0; t < 2 * MF_MAXP; ++t) acc[mi][t] = mf_f32x4{0.f, 0.f, 0.f, 0.f};
      mlp_dispatch(sh, [&]<int MCNT, int NPW, int MC1>() {
        if (MCNT == MF_MT && st.K == 416) mlp_gemm_static<MCNT, NPW, 13, MC1>(act, a.act_str, st.wf, sh, lane, acc);
        else mlp_gemm_t<MCNT, NPW, MC1>(act, a.act_str, st.wf, st.K, sh, lane, acc);
      });
      __builtin_amdgcn_sched_barrier(0);
      // column sums of the step's input (the bias gradient of its layer): 8 row slices x 16-byte column chunks
      if (st.colsum != nullptr) {
        const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
        if (c < (st.K >> 3)) {
          float sum[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) sum[j] = 0.f;
#pragma unroll 4
          for (int rr = 0; rr < MF_ROWS / 8; ++rr) {
            float f[8];
            Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(act + (sl * (MF_ROWS / 8) + rr) * a.act_str + c * 16), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum[j] += f[j];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) scratch[sl * 512 + c * 8 + j] = sum[j];
        }
      }
      const bool last = s + 1 == a.nsteps;
      const int out_cols = last ? st.out_stride : st.N;
      MF_BAR();
      if (st.colsum != nullptr && threadIdx.x < st.K) {
        float t = 0.f;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) t += scratch[sl * 512 + threadIdx.x];
        csum[s * 512 + threadIdx.x] += t;
      }
      if (RO_MASKS && st.mask != nullptr) {
        MF_BAR();                                                        // the partial sums have been read
        reinterpret_cast<uint4*>(mstage)[threadIdx.x] = mraw;
        MF_BAR();
      }
      const int lrow = sh.mt0 * 16 + r;          // item addresses as in the forward
      char* lds0 = act + lrow * a.act_str + 16 * q;
      bf16_t* out0 = reinterpret_cast<bf16_t*>(st.out) + (row0 + lrow) * st.out_stride + 8 * q;
      const int64_t left64 = a.rows - row0 - lrow;
      const int left = st.out == nullptr ? 0 : (left64 > MF_ROWS ? MF_ROWS : (int)left64);
      mlp_dispatch(sh, [&]<int MCNT, int NPW, int MC1>() {
#pragma unroll
        for (int pi = 0; pi < NPW; ++pi) {
          const bool colok = 32 * sh.pair[pi] + 8 * q < out_cols;
#pragma unroll
          for (int mi = 0; mi < MCNT; ++mi) {
            const bool part = MC1 > 0 && pi == NPW - 1;      // the pair this wave shares by row tiles (mlp_share)
            if (part && mi >= MC1) continue;
            const int rt = part ? sh.mt1 - sh.mt0 + mi : mi;
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              v[i] = acc[mi][2 * pi][i];
              v[4 + i] = acc[mi][2 * pi + 1][i];
            }
            if (st.mask != nullptr) {
              int word = (int)mword[(pi * MCNT + mi) >> 2];
              int sh8 = 8 * ((pi * MCNT + mi) & 3);
              if constexpr (RO_MASKS) {
                const int rt_ = lrow + 16 * rt;                              // row inside the 128-row tile
                const int slot = 2 * (64 * (rt_ >> 6) + (rt_ & 31) + 32 * (q & 1)) + ((rt_ >> 5) & 1);
                const unsigned w16 = mstage[(sh.pair[pi] >> 2) * 1024 + slot * 4 + (sh.pair[pi] & 3)];
                const unsigned nib = w16 >> (4 * (q >> 1));
                word = (int)((nib & 0xFu) | ((nib >> 4) & 0xF0u));
                sh8 = 0;
              }
#pragma unroll
              for (int j = 0; j < 8; ++j)      // sign-extended 1-bit field (0 / all ones) ANDed onto the value: 2 instructions
                v[j] = __int_as_float(__float_as_int(v[j]) & __builtin_amdgcn_sbfe(word, sh8 + mf_mask_bit(j), 1));
            }
            const uint4 pk = Vec16<bf16_t>::pack(v);
            if (!last || a.colsum_in != nullptr) *reinterpret_cast<uint4*>(lds0 + rt * 16 * a.act_str + sh.pair[pi] * 64) = pk;
            if (colok && 16 * rt < left)      // as in the forward: from the registers
              store_stream(reinterpret_cast<uint4*>(out0 + (size_t)rt * 16 * st.out_stride + 32 * sh.pair[pi]), pk);
          }
        }
      });
      if (!last) MF_BAR();
    }
    if (a.colsum_in != nullptr) {
      // the stack's input came out of a ReLU (mask_in): the masked input gradient is the gradient of that layer's
      // pre-activation, and its column sums are that layer's bias gradient -- one more pass over the rows in LDS
      const int Kin = a.step[a.nsteps - 1].N;
      MF_BAR();
      const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
      if (c < (Kin >> 3)) {
        float sum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum[j] = 0.f;
#pragma unroll 4
        for (int rr = 0; rr < MF_ROWS / 8; ++rr) {
          float f[8];
          Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(act + (sl * (MF_ROWS / 8) + rr) * a.act_str + c * 16), f);
#pragma unroll
          for (int j = 0; j < 8; ++j) sum[j] += f[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) scratch[sl * 512 + c * 8 + j] = sum[j];
      }
      MF_BAR();
      if (threadIdx.x < Kin) {
        float t = 0.f;
#pragma unroll
        for (int sl2 = 0; sl2 < 8; ++sl2) t += scratch[sl2 * 512 + threadIdx.x];
        csum[a.nsteps * 512 + threadIdx.x] += t;
      }
    }
  }
  for (int s = 0; s < a.nsteps; ++s)
    if (a.step[s].colsum != nullptr && threadIdx.x < a.step[s].K)
      a.step[s].colsum[(size_t)blockIdx.x * a.step[s].K + threadIdx.x] = csum[s * 512 + threadIdx.x];
  if (a.colsum_in != nullptr && threadIdx.x < a.step[a.nsteps - 1].N)
    a.colsum_in[(size_t)blockIdx.x * a.step[a.nsteps - 1].N + threadIdx.x] = csum[a.nsteps * 512 + threadIdx.x];
}

// out[i] = sum_p part[p][i]
// the bias gradients of all layers in one launch (blockIdx.y = layer), 16 waves per 64 columns
struct MlpColsumArgs {
  const float* part[MF_MAXL];
  float* out[MF_MAXL];
  int n[MF_MAXL];
  int nparts;
};
__global__ __launch_bounds__(1024) void mlp_colsum_reduce_many_kernel(MlpColsumArgs a) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = a.n[blockIdx.y];
  const float* part = a.part[blockIdx.y];
  const int i = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (i < n)
    for (int p = ty; p < a.nparts; p += 16) s += part[(size_t)p * n + i];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][tx];
    a.out[blockIdx.y][i] = t;
  }
}

__global__ __launch_bounds__(256) void mlp_colsum_reduce_kernel(const float* __restrict__ part, int nparts, int n,
                                                                float* __restrict__ out) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (i < n)
    for (int p = ty; p < nparts; p += 4) s += part[(size_t)p * n + i];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < n) out[i] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

static inline int pad32(int v) { return (v + 31) / 32 * 32; }
constexpr int MF_GRID = TRS_MF_GRID;

static bool mlp_fused_covers(int L, const int32_t* w) {
  if (L < 1 || L > MF_MAXL) return false;
  for (int i = 0; i <= L; ++i)
    if (w[i] < 8 || w[i] % 8 != 0 || pad32(w[i]) > 32 * MF_MAXP * MF_WAVES) return false;
  return true;
}

// mlp_ro.hip: the row-owner kernels for the stack shapes they are instantiated for
int mlp_resolve_family(int L, const int32_t* widths, int64_t rows, int request);
size_t mlp_ro_mask_bytes(int64_t rows);
int mlp_ro_fwd(const void* x, int64_t rows, int L, const int32_t* widths, const void* const* weights,
               const void* const* biases, void* const* hidden, void* const* masks, void* mask_in, void* y, void* workspace,
               hipStream_t s, int phase, int x_stride);
struct RoColsum {
  const float* part[9];
  float* out[9];
  int n[9];
  int count, nparts;
};
int mlp_ro_bwd(const void* gy, int64_t rows, int L, const int32_t* widths, const void* const* weights,
               const void* const* masks, void* const* gz, float* const* gbias, void* gx, const void* mask_in,
               float* gbias_in, void* workspace, hipStream_t s, RoColsum* cs, int phase);

static int mlp_act_str(int L, const int32_t* w) {
  int mx = 0;
  for (int i = 0; i <= L; ++i) mx = std::max(mx, pad32(w[i]));
  return mx * 2 + 16;
}

static size_t mlp_frag_bytes(int L, const int32_t* w) {
  size_t b = 0;
  for (int l = 0; l < L; ++l) b += (size_t)pad32(w[l]) * pad32(w[l + 1]) * 2;
  return (b + 255) / 256 * 256;
}

// The weight copies of the tile kernels as job lists (the entry points below and trs_mlp_pack_branch build the SAME jobs:
// what a PACK call leaves is what a RUN call reads).  Return the grid width the jobs want.
static int mlp_fwd_pack_jobs(int L, const int32_t* widths, const void* const* weights, const void* const* biases, char* wsp,
                             MlpPackJob* job) {
  float* bias_base = (float*)(wsp + mlp_frag_bytes(L, widths));
  size_t woff = 0, boff = 0;
  int blocks = 1;
  for (int l = 0; l < L; ++l) {
    const int K = pad32(widths[l]), N = pad32(widths[l + 1]);
    job[l] = MlpPackJob{(const bf16_t*)weights[l], (const bf16_t*)biases[l], (bf16_t*)(wsp + woff), bias_base + boff,
                        widths[l + 1], widths[l], N / 16, K / 32, N, 0};
    blocks = std::max(blocks, std::min(256, (N / 16 * (K / 32) * 64 + 255) / 256));
    woff += (size_t)K * N * 2;
    boff += N;
  }
  return blocks;
}
static int mlp_bwd_pack_jobs(int L, const int32_t* widths, const void* const* weights, char* wsp, MlpPackJob* job) {
  size_t woff = 0;
  int blocks = 1;
  for (int sidx = 0; sidx < L; ++sidx) {
    const int l = L - 1 - sidx;
    const int K = pad32(widths[l + 1]), N = pad32(widths[l]);      // contraction over layer l's outputs, output = its inputs
    job[sidx] = MlpPackJob{(const bf16_t*)weights[l], nullptr, (bf16_t*)(wsp + woff), nullptr, widths[l + 1], widths[l],
                           N / 16, K / 32, 0, 1};
    blocks = std::max(blocks, std::min(256, (N / 16 * (K / 32) * 64 + 255) / 256));
    woff += (size_t)K * N * 2;
  }
  return blocks;
}
static int mlp_gemm_pack_job(const void* W, int out_f, int in_f, void* workspace, MlpPackJob* job) {
  const int K = pad32(out_f), N = pad32(in_f);
  job[0] = MlpPackJob{(const bf16_t*)W, nullptr, (bf16_t*)workspace, nullptr, out_f, in_f, N / 16, K / 32, 0, 1};
  return std::min(256, (N / 16 * (K / 32) * 64 + 255) / 256);
}

}  // namespace trs

using namespace trs;

/* workspace: [fragment-order weights][fp32 biases (forward) | column-sum partials (backward)] */
extern "C" size_t trs_mlp_fused_workspace_bytes(int32_t num_layers, const int32_t* widths) {
  if (!mlp_fused_covers(num_layers, widths)) return 0;
  size_t sum = 0;
  for (int l = 0; l <= num_layers; ++l) sum += (size_t)pad32(widths[l]);
  return mlp_frag_bytes(num_layers, widths) + (size_t)MF_GRID * 8 * sum * 4 + 4096;      // (8: per-wave slices of the row-owner kernels)
}

extern "C" int32_t trs_mlp_fused_family(int32_t num_layers, const int32_t* widths, int64_t rows, int32_t request) {
  if (!mlp_fused_covers(num_layers, widths)) return 0;
  return mlp_resolve_family(num_layers, widths, rows, request);
}

extern "C" size_t trs_mlp_fused_mask_bytes(int64_t rows) {      // either kernel family's layout
  return std::max((size_t)((rows + MF_ROWS - 1) / MF_ROWS) * MF_MASK_TILE, mlp_ro_mask_bytes(rows));
}

extern "C" int trs_mlp_fused_supported(int32_t num_layers, const int32_t* widths) {
  if (!mlp_fused_covers(num_layers, widths)) return 0;
  const size_t lds = (size_t)MF_ROWS * mlp_act_str(num_layers, widths) + 8 * 512 * 4 +
                     (size_t)(num_layers + 1) * 512 * 4;
  return lds <= 160 * 1024 ? 1 : 0;
}

extern "C" int trs_mlp_fused_fwd(const void* x, int64_t rows, int32_t num_layers, const int32_t* widths,
                                 const void* const* weights, const void* const* biases, void* const* hidden,
                                 void* const* masks, void* mask_in, void* y, int32_t dtype, int32_t family,
                                 int32_t phase, int32_t x_stride, void* workspace, size_t ws_bytes, trs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "mlp_fused_fwd: bf16 only");
  TRS_REQUIRE(trs_mlp_fused_supported(num_layers, widths), TRS_ESHAPE, "mlp_fused_fwd: unsupported layer widths");
  TRS_REQUIRE(phase == TRS_MLP_PHASE_ALL || phase == TRS_MLP_PHASE_PACK || phase == TRS_MLP_PHASE_RUN, TRS_EINVAL,
              "mlp_fused_fwd: phase %d", phase);
  // (a PACK call reads the parameters only: the input rows may not exist yet)
  TRS_REQUIRE(weights && biases && workspace && (phase == TRS_MLP_PHASE_PACK || (x && y && hidden && masks)), TRS_EINVAL,
              "mlp_fused_fwd: NULL pointer");
  TRS_REQUIRE(ws_bytes >= trs_mlp_fused_workspace_bytes(num_layers, widths), TRS_EWORKSPACE,
              "mlp_fused_fwd: workspace too small");
  const int L = num_layers;
  const int fam = mlp_resolve_family(L, widths, rows, family);
  TRS_REQUIRE(fam != 0, TRS_EINVAL, "mlp_fused_fwd: kernel family %d is not available for this stack (trs_mlp_fused_family)", family);
  if (rows == 0) return TRS_OK;
  TRS_REQUIRE(x_stride == 0 || (x_stride >= widths[0] && x_stride % 8 == 0), TRS_ESHAPE, "mlp_fused_fwd: x_stride %d", x_stride);
  TRS_REQUIRE(x_stride == 0 || x_stride == widths[0] || fam != TRS_MLP_FAMILY_TILE, TRS_ESHAPE,
              "mlp_fused_fwd: rows wider than the stack's input (x_stride %d > %d) are read by the row-owner kernels only", x_stride, widths[0]);
  if (fam == TRS_MLP_FAMILY_ROW_OWNER || fam == TRS_MLP_FAMILY_MIXED)
    return mlp_ro_fwd(x, rows, L, widths, weights, biases, hidden, masks, mask_in, y, workspace, s, phase,
                      x_stride ? x_stride : widths[0]);
  MlpArgs a;
  a.nsteps = L;
  a.in = x;
  a.in_stride = widths[0];
  a.rows = rows;
  a.nbias = 0;
  a.act_str = mlp_act_str(L, widths);
  a.mask_in = (uint8_t*)mask_in;
  a.colsum_in = nullptr;
  char* wsp = (char*)workspace;
  float* bias_base = (float*)(wsp + mlp_frag_bytes(L, widths));
  size_t woff = 0, boff = 0;
  MlpPackArgs pk;
  const int pk_blocks = mlp_fwd_pack_jobs(L, widths, weights, biases, wsp, pk.job);
  for (int l = 0; l < L; ++l) {
    const int K = pad32(widths[l]), N = pad32(widths[l + 1]);
    bf16_t* wf = (bf16_t*)(wsp + woff);
    float* bf = bias_base + boff;
    MlpStep& st = a.step[l];
    st.wf = (const uint4*)wf;
    st.bias = bf;
    st.K = K;
    st.N = N;
    st.relu = l + 1 < L ? 1 : 0;
    st.out = phase == TRS_MLP_PHASE_PACK ? nullptr : (l + 1 < L ? hidden[l] : y);
    st.out_stride = l + 1 < L ? N : widths[L];
    st.mask = (l + 1 < L && phase != TRS_MLP_PHASE_PACK) ? (uint8_t*)masks[l] : nullptr;
    st.colsum = nullptr;
    woff += (size_t)K * N * 2;
    boff += N;
  }
  if (phase != TRS_MLP_PHASE_RUN) hipLaunchKernelGGL(mlp_prepack_many_kernel, dim3(pk_blocks, L), dim3(256), 0, s, pk);
  if (phase == TRS_MLP_PHASE_PACK) return check_launch("mlp_fused_fwd(pack)");
  a.nbias = (int)boff;
  const size_t lds = (size_t)MF_ROWS * a.act_str + boff * 4;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)mlp_fused_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return check_launch("mlp_fused_fwd: LDS attribute");
    attr = true;
  }
  const int64_t ntiles = (rows + MF_ROWS - 1) / MF_ROWS;
  hipLaunchKernelGGL(mlp_fused_fwd_kernel, dim3((int)std::min<int64_t>(ntiles, MF_GRID)), dim3(64 * MF_WAVES), lds, s, a);
  return check_launch("mlp_fused_fwd");
}

/* gz[l] (l = 0..L-2): d(pre-activation of layer l), (rows x pad32(widths[l+1])) bf16 -- d(pre-activation) of the last
 * layer is gy itself.  gbias[l] (l = 0..L-1): pad32(widths[l+1]) fp32 each, written.  gx: (rows x widths[0]). */
extern "C" int trs_mlp_fused_bwd_data(const void* gy, int64_t rows, int32_t num_layers, const int32_t* widths,
                                      const void* const* weights, const void* const* masks, void* const* gz,
                                      float* const* gbias, void* gx, const void* mask_in, float* gbias_in, int32_t dtype,
                                      int32_t family, int32_t phase, void* workspace, size_t ws_bytes,
                                      trs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "mlp_fused_bwd_data: bf16 only");
  TRS_REQUIRE(trs_mlp_fused_supported(num_layers, widths), TRS_ESHAPE, "mlp_fused_bwd_data: unsupported layer widths");
  TRS_REQUIRE(phase == TRS_MLP_PHASE_ALL || phase == TRS_MLP_PHASE_PACK || phase == TRS_MLP_PHASE_RUN, TRS_EINVAL,
              "mlp_fused_bwd_data: phase %d", phase);
  const bool pack_only = phase == TRS_MLP_PHASE_PACK;      // reads the parameters only: no gradient exists yet
  TRS_REQUIRE(weights && workspace && (pack_only || (gy && masks && gz && gbias)), TRS_EINVAL, "mlp_fused_bwd_data: NULL pointer");
  TRS_REQUIRE(ws_bytes >= trs_mlp_fused_workspace_bytes(num_layers, widths), TRS_EWORKSPACE,
              "mlp_fused_bwd_data: workspace too small");
  const int L = num_layers;
  // the family is what the forward of these masks ran (trs_mlp_fused_family): AUTO is refused here because the policy
  // behind it is not the caller's record of what happened
  bool fam_ok = (family == TRS_MLP_FAMILY_TILE || family == TRS_MLP_FAMILY_ROW_OWNER || family == TRS_MLP_FAMILY_MIXED) &&
                mlp_resolve_family(L, widths, rows, family) == family;
  if (!fam_ok && family == TRS_MLP_FAMILY_MIXED && L <= MF_MAXL) {
    // a mixed forward that read the first columns of wider rows (x_stride): this call sees the full row width
    int32_t wn[MF_MAXL + 1];
    for (int l = 0; l <= L; ++l) wn[l] = widths[l];
    for (int w0 = pad32(widths[0]) - 32; w0 >= 32 && !fam_ok; w0 -= 32) {
      wn[0] = w0;
      fam_ok = mlp_resolve_family(L, wn, rows, family) == family;
    }
  }
  TRS_REQUIRE(fam_ok, TRS_EINVAL,
              "mlp_fused_bwd_data: family must be the TILE / ROW_OWNER / MIXED value the forward ran under (got %d)", family);
  TRS_REQUIRE(pack_only || family != TRS_MLP_FAMILY_ROW_OWNER || gx != nullptr, TRS_EINVAL, "mlp_fused_bwd_data: the row-owner kernels need gx");
  TRS_REQUIRE(pack_only || ((mask_in == nullptr) == (gbias_in == nullptr) && (mask_in == nullptr || (L + 1 <= MF_MAXL && gx != nullptr))),
              TRS_EINVAL, "mlp_fused_bwd_data: mask_in and gbias_in come together (and with gx, at most %d layers)", MF_MAXL - 1);
  if (rows == 0 && !pack_only) {
    for (int l = 0; l < L; ++l)
      if (int rc = zero_bytes(gbias[l], (size_t)pad32(widths[l + 1]) * 4, s)) return rc;
    if (gbias_in) return zero_bytes(gbias_in, (size_t)pad32(widths[0]) * 4, s);
    return TRS_OK;
  }
  if (family == TRS_MLP_FAMILY_ROW_OWNER) {
    RoColsum rc;
    const int rcode = mlp_ro_bwd(gy, rows, L, widths, weights, masks, gz, gbias, gx, mask_in, gbias_in, workspace, s, &rc, phase);
    if (rcode != TRS_OK || pack_only) return rcode;
    MlpColsumArgs cs;
    cs.nparts = rc.nparts;
    int kmax = 0;
    for (int i = 0; i < rc.count; ++i) {
      cs.part[i] = rc.part[i];
      cs.out[i] = rc.out[i];
      cs.n[i] = rc.n[i];
      kmax = std::max(kmax, rc.n[i]);
    }
    hipLaunchKernelGGL(mlp_colsum_reduce_many_kernel, dim3((kmax + 63) / 64, rc.count), dim3(1024), 0, s, cs);
    return check_launch("mlp_fused_bwd_data");
  }
  MlpArgs a;
  a.nsteps = L;
  a.in = gy;
  a.in_stride = widths[L];
  a.rows = rows;
  a.nbias = 0;
  a.act_str = mlp_act_str(L, widths);
  const int64_t ntiles = (rows + MF_ROWS - 1) / MF_ROWS;
  const int grid = (int)std::min<int64_t>(ntiles, MF_GRID);
  char* wsp = (char*)workspace;
  float* part_base = (float*)(wsp + mlp_frag_bytes(L, widths));
  size_t woff = 0, poff = 0;
  MlpPackArgs pk;
  const int pk_blocks = mlp_bwd_pack_jobs(L, widths, weights, wsp, pk.job);
  for (int sidx = 0; sidx < L; ++sidx) {
    const int l = L - 1 - sidx;
    const int K = pad32(widths[l + 1]), N = pad32(widths[l]);      // contraction over layer l's outputs, output = its inputs
    bf16_t* wf = (bf16_t*)(wsp + woff);
    MlpStep& st = a.step[sidx];
    st.wf = (const uint4*)wf;
    st.bias = nullptr;
    st.K = K;
    st.N = N;
    st.relu = 0;
    st.out = pack_only ? nullptr : (l > 0 ? gz[l - 1] : gx);
    st.out_stride = l > 0 ? N : widths[0];
    st.mask = pack_only ? nullptr : (l > 0 ? (uint8_t*)masks[l - 1] : (uint8_t*)mask_in);
    st.colsum = part_base + poff;
    woff += (size_t)K * N * 2;
    poff += (size_t)grid * K;
  }
  a.mask_in = nullptr;
  a.colsum_in = mask_in != nullptr ? part_base + poff : nullptr;      // (the workspace counts widths[0] as well)
  a.ro_masks = family == TRS_MLP_FAMILY_MIXED ? 1 : 0;
  const size_t lds = (size_t)MF_ROWS * a.act_str + 8 * 512 * 4 + (size_t)(L + 1) * 512 * 4;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)mlp_fused_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)mlp_fused_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
            hipSuccess)
      return check_launch("mlp_fused_bwd_data: LDS attribute");
    attr = true;
  }
  if (phase != TRS_MLP_PHASE_RUN) hipLaunchKernelGGL(mlp_prepack_many_kernel, dim3(pk_blocks, L), dim3(256), 0, s, pk);
  if (pack_only) return check_launch("mlp_fused_bwd_data(pack)");
  if (a.ro_masks) hipLaunchKernelGGL(mlp_fused_bwd_kernel<true>, dim3(grid), dim3(64 * MF_WAVES), lds, s, a);
  else hipLaunchKernelGGL(mlp_fused_bwd_kernel<false>, dim3(grid), dim3(64 * MF_WAVES), lds, s, a);
  MlpColsumArgs cs;
  cs.nparts = grid;
  int kmax = 0;
  for (int sidx = 0; sidx < L; ++sidx) {
    const int l = L - 1 - sidx;
    cs.part[sidx] = a.step[sidx].colsum;
    cs.out[sidx] = gbias[l];
    cs.n[sidx] = a.step[sidx].K;
    kmax = std::max(kmax, a.step[sidx].K);
  }
  int nsum = L;
  if (a.colsum_in != nullptr) {
    cs.part[L] = a.colsum_in;
    cs.out[L] = gbias_in;
    cs.n[L] = pad32(widths[0]);
    kmax = std::max(kmax, cs.n[L]);
    nsum = L + 1;
  }
  hipLaunchKernelGGL(mlp_colsum_reduce_many_kernel, dim3((kmax + 63) / 64, nsum), dim3(1024), 0, s, cs);
  return check_launch("mlp_fused_bwd_data");
}

/* y (rows x in_f) = x[:, :out_f] (rows x x_stride) @ W (out_f x in_f, row stride in_f), bf16 -- the input gradient of an
 * nn.Linear(in_f, out_f) whose weight is W.  out_f <= 512, in_f % 8 == 0, x_stride % 8 == 0; the columns of x between
 * out_f and pad32(out_f) must be readable (they meet zero weight rows).  workspace: trs_rows_gemm_workspace_bytes. */
extern "C" size_t trs_rows_gemm_workspace_bytes(int32_t out_f, int32_t in_f) {
  if (out_f <= 0 || in_f <= 0) return 256;
  return (size_t)pad32(out_f) * pad32(in_f) * 2 + 256;
}

extern "C" int trs_rows_gemm_supported(int32_t out_f, int32_t in_f, int32_t x_stride) {
  if (out_f < 8 || out_f > 512 || in_f < 32 || in_f % 8 != 0 || x_stride % 8 != 0 || x_stride < pad32(out_f)) return 0;
  const size_t lds = (size_t)MF_ROWS * (pad32(out_f) * 2 + 16);
  return lds <= 160 * 1024 ? 1 : 0;
}

extern "C" int trs_rows_gemm(const void* x, int64_t rows, int32_t x_stride, const void* W, int32_t out_f, int32_t in_f,
                             int32_t dtype, int32_t phase, void* y, void* workspace, size_t ws_bytes, trs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "rows_gemm: bf16 only");
  TRS_REQUIRE(trs_rows_gemm_supported(out_f, in_f, x_stride), TRS_ESHAPE, "rows_gemm: unsupported shape (K %d, N %d)", out_f, in_f);
  TRS_REQUIRE(phase == TRS_MLP_PHASE_ALL || phase == TRS_MLP_PHASE_PACK || phase == TRS_MLP_PHASE_RUN, TRS_EINVAL, "rows_gemm: phase %d", phase);
  const bool pack_only = phase == TRS_MLP_PHASE_PACK;
  TRS_REQUIRE(W && workspace && (pack_only || (x && y)), TRS_EINVAL, "rows_gemm: NULL pointer");
  TRS_REQUIRE(ws_bytes >= trs_rows_gemm_workspace_bytes(out_f, in_f), TRS_EWORKSPACE, "rows_gemm: workspace too small");
  TRS_REQUIRE(aligned16(x) && aligned16(y) && aligned16(workspace), TRS_EALIGN, "rows_gemm: 16-byte alignment");
  if (rows == 0 && !pack_only) return TRS_OK;
  const int K = pad32(out_f), N = pad32(in_f);
  MlpPackArgs pk;
  const int pk_blocks = mlp_gemm_pack_job(W, out_f, in_f, workspace, pk.job);
  if (phase != TRS_MLP_PHASE_RUN) hipLaunchKernelGGL(mlp_prepack_many_kernel, dim3(pk_blocks, 1), dim3(256), 0, s, pk);
  if (pack_only) return check_launch("rows_gemm(pack)");
  RowsGemmArgs a;
  a.in = x;
  a.wf = (const uint4*)workspace;
  a.out = y;
  a.rows = rows;
  a.in_stride = x_stride;
  a.K = K;
  a.N = N;
  a.out_cols = in_f;
  a.out_stride = in_f;
  a.act_str = K * 2 + 16;
  const size_t lds = (size_t)MF_ROWS * a.act_str;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)mlp_rows_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return check_launch("rows_gemm: LDS attribute");
    attr = true;
  }
  const int64_t ntiles = (rows + MF_ROWS - 1) / MF_ROWS;
  hipLaunchKernelGGL(mlp_rows_gemm_kernel, dim3((int)std::min<int64_t>(ntiles, MF_GRID)), dim3(64 * MF_WAVES), lds, s, a);
  return check_launch("rows_gemm");
}

/* Every weight copy of a deep branch in ONE launch: the PACK phases of trs_mlp_fused_fwd (into ws_fwd), of
 * trs_mlp_fused_bwd_data (into ws_bwd; NULL: none) and of the trs_rows_gemm of the layer in front of the stack (gemm_W
 * (gemm_out_f x gemm_in_f) into ws_gemm; NULL: none).  The three RUN calls then find what their own PACK call would have
 * left.  Tile family only (TRS_ESHAPE for a stack that `family` resolves to the row-owner kernels: use the PACK phases). */
extern "C" int trs_mlp_pack_branch(int64_t rows, int32_t num_layers, const int32_t* widths, const void* const* weights,
                                   const void* const* biases, int32_t family, void* ws_fwd, void* ws_bwd, size_t ws_bytes,
                                   const void* gemm_W, int32_t gemm_out_f, int32_t gemm_in_f, void* ws_gemm,
                                   size_t ws_gemm_bytes, trs_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  TRS_REQUIRE(trs_mlp_fused_supported(num_layers, widths), TRS_ESHAPE, "mlp_pack_branch: unsupported layer widths");
  TRS_REQUIRE(weights && biases && ws_fwd, TRS_EINVAL, "mlp_pack_branch: NULL pointer");
  TRS_REQUIRE(ws_bytes >= trs_mlp_fused_workspace_bytes(num_layers, widths), TRS_EWORKSPACE, "mlp_pack_branch: workspace too small");
  TRS_REQUIRE(mlp_resolve_family(num_layers, widths, rows, family) == TRS_MLP_FAMILY_TILE, TRS_ESHAPE,
              "mlp_pack_branch: the tile kernels only (family %d resolves to another one for this stack)", family);
  const int L = num_layers;
  MlpPackArgs pk;
  int njobs = L;
  int blocks = mlp_fwd_pack_jobs(L, widths, weights, biases, (char*)ws_fwd, pk.job);
  if (ws_bwd != nullptr) {
    blocks = std::max(blocks, mlp_bwd_pack_jobs(L, widths, weights, (char*)ws_bwd, pk.job + njobs));
    njobs += L;
  }
  if (gemm_W != nullptr) {
    TRS_REQUIRE(ws_gemm != nullptr && trs_rows_gemm_supported(gemm_out_f, gemm_in_f, pad32(gemm_out_f)), TRS_ESHAPE,
                "mlp_pack_branch: rows_gemm shape (K %d, N %d)", gemm_out_f, gemm_in_f);
    TRS_REQUIRE(ws_gemm_bytes >= trs_rows_gemm_workspace_bytes(gemm_out_f, gemm_in_f), TRS_EWORKSPACE,
                "mlp_pack_branch: rows_gemm workspace too small");
    blocks = std::max(blocks, mlp_gemm_pack_job(gemm_W, gemm_out_f, gemm_in_f, ws_gemm, pk.job + njobs));
    njobs += 1;
  }
  hipLaunchKernelGGL(mlp_prepack_many_kernel, dim3(blocks, njobs), dim3(256), 0, s, pk);
  return check_launch("mlp_pack_branch");
}
