// Per-pair bilinear form on the matrix cores, forward (bf16, E = 32 | 64):
//   T[b,p,:] = x[b,i_p,:] @ W_p        OPN 'mat':  out[b,p] = T . x[b,j_p,:]      Bilinear 'each':  out[b,p,:] = T * x_j + bias_p
// (outer_product_network.py:107-121, bilinear_interaction.py:144-149).  One kernel, nothing of size B*NC2*E besides the
// layer's own output.  Decomposition: a TASK is up to three adjacent pairs (i, j0..j0+2) -- same i, consecutive p; a
// wave owns one task for a whole range of samples and keeps the three W_p^T in registers as MFMA A operands
// (D^T[h][sample] = W_p^T[h][:] . x_i[sample][:]: rows of W^T fed in a permuted order so that a lane's outputs for its
// sample are runs of 8 consecutive h).  Per 16-sample tile the wave loads the x_i fragments once (B operand: 16 bytes
// per lane straight from the (B,N,E) block) and, per pair, the x_j run that is both the epilogue operand and -- by the
// same address pattern -- what a B fragment of x_j would be.  The four waves of a workgroup take adjacent tasks of
// the same sample range, so their x rows hit in L1/L2.
#include <algorithm>

#include "trs_common.hpp"

namespace trs {

typedef __attribute__((ext_vector_type(8))) __bf16 pb_bf16x8;
typedef __attribute__((ext_vector_type(4))) float pb_f32x4;

constexpr int PB_PPT = 3;        // pairs per task

template <int KS /* E/32 */, int MODE /* 0: sum over h, 1: per-h output + bias */>
__global__ __launch_bounds__(256) void pair_bil_fwd_mfma_kernel(const bf16_t* __restrict__ x,
                                                                const bf16_t* __restrict__ Wt /* (P,H,E) */,
                                                                const bf16_t* __restrict__ bias /* (P,E) | null */,
                                                                const int32_t* __restrict__ tasks /* (T,3): i, j0, cnt */,
                                                                int ntasks, int nsplit, int64_t B, int N,
                                                                bf16_t* __restrict__ out) {
  constexpr int E = 32 * KS, MT = 2 * KS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
  const int P = N * (N - 1) / 2;
  const int gw = blockIdx.x * 4 + wave;                 // global wave id: task fastest within a sample split
  const int task = gw % ntasks, split = gw / ntasks;
  if (split >= nsplit) return;
  const int fi = tasks[3 * task], j0 = tasks[3 * task + 1], cnt = tasks[3 * task + 2];
  const int p0 = pair_index_of(fi, j0, N);
  // resident A fragments: row m of tile mt <-> h = 32 (mt>>1) + 8 (m>>2) + 4 (mt&1) + (m&3)
  uint4 Wf[PB_PPT][MT][KS];
  float bv[PB_PPT][KS][8];
#pragma unroll
  for (int c = 0; c < PB_PPT; ++c) {
    const int pc = p0 + (c < cnt ? c : 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int h = 32 * (mt >> 1) + 8 * (n >> 2) + 4 * (mt & 1) + (n & 3);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        Wf[c][mt][ks] = *reinterpret_cast<const uint4*>(Wt + ((size_t)pc * E + h) * E + 32 * ks + 8 * q);
    }
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      if (MODE == 1 && bias != nullptr)
        Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(bias + (size_t)pc * E + 32 * u + 8 * q), bv[c][u]);
      else
#pragma unroll
        for (int k = 0; k < 8; ++k) bv[c][u][k] = 0.f;
    }
  }
  const int64_t tiles = (B + 15) / 16;
  const int64_t per = (tiles + nsplit - 1) / nsplit;
  const int64_t t_lo = split * per, t_hi = std::min<int64_t>(t_lo + per, tiles);
  for (int64_t t = t_lo; t < t_hi; ++t) {
    const int64_t b = t * 16 + n;
    const bool live = b < B;
    const bf16_t* xb = x + (live ? b : 0) * (int64_t)N * E;
    uint4 xi[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xi[ks] = *reinterpret_cast<const uint4*>(xb + fi * E + 32 * ks + 8 * q);
    uint4 xj[PB_PPT][KS];
#pragma unroll
    for (int c = 0; c < PB_PPT; ++c)
#pragma unroll
      for (int u = 0; u < KS; ++u)
        xj[c][u] = *reinterpret_cast<const uint4*>(xb + (j0 + (c < cnt ? c : 0)) * E + 32 * u + 8 * q);
#pragma unroll
    for (int c = 0; c < PB_PPT; ++c) {
      if (c >= cnt) break;
      pb_f32x4 acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt] = pb_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pb_bf16x8, Wf[c][mt][ks]),
                                                            __builtin_bit_cast(pb_bf16x8, xi[ks]), acc[mt], 0, 0, 0);
      }
      float part = 0.f;
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        float xv[8];
        Vec16<bf16_t>::unpack(xj[c][u], xv);
        float r8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float tv = k < 4 ? acc[2 * u][k & 3] : acc[2 * u + 1][k & 3];
          if (MODE == 0) part = fmaf(tv, xv[k], part);
          else r8[k] = fmaf(tv, xv[k], bv[c][u][k]);
        }
        if (MODE == 1 && live)
          *reinterpret_cast<uint4*>(out + ((b * P + p0 + c) * (int64_t)E) + 32 * u + 8 * q) = Vec16<bf16_t>::pack(r8);
      }
      if (MODE == 0) {
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (q == 0 && live) out[b * P + p0 + c] = from_f32<bf16_t>(part);
      }
    }
  }
}

}  // namespace trs

using namespace trs;

/* tasks: device int32 (ntasks, 3) = (i, j0, count <= 3) covering every pair once, adjacent pairs of one i per task */
extern "C" int trs_pair_bilinear_fwd_mfma(const void* x, const void* Wt, const void* bias, const int32_t* tasks,
                                          int32_t ntasks, int32_t mode, int64_t B, int32_t N, int32_t E, int32_t dtype,
                                          void* out, trs_stream_t stream) {
  TRS_REQUIRE(B >= 0 && N >= 2 && ntasks > 0, TRS_EINVAL, "pair_bilinear_fwd_mfma: bad size");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(x && Wt && tasks && out, TRS_EINVAL, "pair_bilinear_fwd_mfma: NULL pointer");
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "pair_bilinear_fwd_mfma: bf16 only (dtype %d)", dtype);
  TRS_REQUIRE(E == 32 || E == 64, TRS_ESHAPE, "pair_bilinear_fwd_mfma: E = %d (32 or 64)", E);
  TRS_REQUIRE(mode == 0 || mode == 1, TRS_EINVAL, "pair_bilinear_fwd_mfma: mode %d", mode);
  TRS_REQUIRE(aligned16(x) && aligned16(Wt) && aligned16(bias) && aligned16(out), TRS_EALIGN,
              "pair_bilinear_fwd_mfma: 16-byte alignment");
  const int64_t tiles = (B + 15) / 16;
  // enough waves for ~2 per SIMD, each with at least a few tiles
  int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(tiles / 4, (2048 + ntasks - 1) / ntasks));
  const int64_t waves = (int64_t)ntasks * nsplit;
  const int grid = (int)((waves + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
#define TRS_PBM(KS_, MODE_)                                                                                         \
  hipLaunchKernelGGL((pair_bil_fwd_mfma_kernel<KS_, MODE_>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x,          \
                     (const bf16_t*)Wt, (const bf16_t*)bias, tasks, ntasks, nsplit, B, N, (bf16_t*)out)
  if (E == 32) { if (mode == 0) TRS_PBM(1, 0); else TRS_PBM(1, 1); }
  else { if (mode == 0) TRS_PBM(2, 0); else TRS_PBM(2, 1); }
#undef TRS_PBM
  return check_launch("pair_bilinear_fwd_mfma");
}
