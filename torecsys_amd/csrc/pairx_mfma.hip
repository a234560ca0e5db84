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

// Waves of 256-thread workgroups of `kernel` that the device holds at once.  The sample splits of these kernels are
// sized from it: a wave owns one (task, split) for the whole launch, so ntasks * nsplit just ABOVE this number (260 tasks
// x 8 splits = 2080 on 2048 slots at two waves per SIMD) leaves a second, nearly empty round.
static int64_t pbm_wave_slots(const void* kernel, size_t dyn_lds) { return (int64_t)resident_blocks(kernel, 256, dyn_lds) * 4; }
// units per wave when `units` are dealt evenly to at most `slots` waves (at least 4 units each); *grid = workgroups of 4
static int64_t pbm_even_share(int64_t units, int64_t slots, int* grid) {
  const int64_t waves = std::max<int64_t>(4, std::min<int64_t>(slots, (units + 3) / 4) / 4 * 4);
  const int64_t per_wave = (units + waves - 1) / waves;
  *grid = (int)((units + per_wave * 4 - 1) / (per_wave * 4));
  return per_wave;
}
// sample splits: as many as fill the slots in ONE round, each with at least min_per units
static int pbm_splits(int64_t units, int ntasks, int64_t min_per, int64_t slots) {
  const int64_t by_slots = std::max<int64_t>(1, slots / ntasks);
  return (int)std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>(1, units / min_per), by_slots));
}

template <int KS /* E/32 */, int MODE /* 0: sum over h, 1: per-h output + bias */>
__global__ __launch_bounds__(256) void pair_bil_fwd_mfma_kernel(const bf16_t* __restrict__ x,
                                                                const bf16_t* __restrict__ Wt /* (P,H,E) */,
                                                                const bf16_t* __restrict__ bias /* (P,E) | null */,
                                                                const int32_t* __restrict__ tasks /* (T,3): i, j0, cnt */,
                                                                int ntasks, int64_t per_wave, int64_t B, int N,
                                                                bf16_t* __restrict__ out) {
  constexpr int E = 32 * KS, MT = 2 * KS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
  const int P = N * (N - 1) / 2;
  // Work = (task, 16-sample tile) pairs, task-major; every wave takes the same number of consecutive pairs (its range
  // may straddle two tasks: the weights are then loaded again).  The kernel is bound by VALU / MFMA issue, not by
  // memory: with whole (task, sample split) units per wave, 1820 waves on 2048 slots left CUs with one workgroup done
  // after 84-115 us and the ones with two after 164.
  const int64_t tiles = (B + 15) / 16;
  const int64_t flat_lo = (int64_t)(blockIdx.x * 4 + wave) * per_wave;
  const int64_t flat_hi = std::min<int64_t>(flat_lo + per_wave, (int64_t)ntasks * tiles);
  for (int64_t flat = flat_lo; flat < flat_hi;) {
  const int task = (int)(flat / tiles);
  const int64_t t_lo = flat - (int64_t)task * tiles, t_hi = std::min<int64_t>(tiles, t_lo + (flat_hi - flat));
  flat += t_hi - t_lo;
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
  // A wave walks its sample tiles one at a time.  All the tasks of a sample split reach the same 16 samples at about the
  // same time, so for its XCD's L2 nearly every tile is a first touch: the x_i / x_j runs arrive with Infinity-Cache
  // latency (2-3 us per tile iteration measured, against ~0.35 us of MFMAs and epilogue).  Two register sets hold the runs
  // of this tile and the next; a set is consumed IN PLACE and each of its registers is re-loaded for the tile after next
  // right behind its last use (x_j of a pair after that pair's epilogue, x_i after the last pair's MFMAs), so a load has
  // between one and two tile times to land and no copy of a register with a load in flight can exist.  Loads are issued
  // by hand (hipcc sinks ordinary loads of read-only memory to their use and puts an s_waitcnt vmcnt(0) behind volatile
  // ones); every lane issues every load -- dead samples and the tiles past the end read valid addresses -- so the counted
  // wait at the top of a tile (all but the 8 youngest loads done = the other set may still be in flight) is exact.
  typedef __attribute__((ext_vector_type(4))) unsigned pb_u32x4;
  constexpr int NLD = KS * (1 + PB_PPT);                 // loads per tile
  pb_u32x4 nxi[2][KS] = {}, nxj[2][PB_PPT][KS] = {};
  auto row_of = [&](int64_t tt) {
    const int64_t tc = tt < t_hi ? tt : t_hi - 1;
    const int64_t b_ = tc * 16 + n;
    return x + (b_ < B ? b_ : 0) * (int64_t)N * E;
  };
  auto load_xi = [&](pb_u32x4 (&xi)[KS], const bf16_t* xb) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16_t* a_ = xb + fi * E + 32 * ks + 8 * q;
      asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(xi[ks]) : "v"(a_));
    }
  };
  auto load_xj = [&](pb_u32x4 (&xj)[KS], int c, const bf16_t* xb) {
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const bf16_t* a_ = xb + (j0 + (c < cnt ? c : 0)) * E + 32 * u + 8 * q;
      asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(xj[u]) : "v"(a_));
    }
  };
  auto tile = [&](int64_t t, pb_u32x4 (&xi)[KS], pb_u32x4 (&xj)[PB_PPT][KS]) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xi[ks]));
#pragma unroll
    for (int c = 0; c < PB_PPT; ++c)
#pragma unroll
      for (int u = 0; u < KS; ++u) asm volatile("" : "+v"(xj[c][u]));
    const int64_t b = t * 16 + n;
    const bool live = b < B;
    const bf16_t* xb2 = row_of(t + 2);
#pragma unroll
    for (int c = 0; c < PB_PPT; ++c) {
      pb_f32x4 acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = pb_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)          // k-steps outermost: consecutive MFMAs never wait for one another
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pb_bf16x8, Wf[c][mt][ks]),
                                                            __builtin_bit_cast(pb_bf16x8, xi[ks]), acc[mt], 0, 0, 0);
      if (c == PB_PPT - 1) {
        __builtin_amdgcn_sched_barrier(0);
        load_xi(xi, xb2);                                  // the tile's last MFMAs are issued: x_i is free
      }
      float part = 0.f, part2 = 0.f;           // two chains: the 16 FMAs of a lane are not one dependent string
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        float xv[8];
        Vec16<bf16_t>::unpack(__builtin_bit_cast(uint4, xj[c][u]), xv);
        float r8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float tv = k < 4 ? acc[2 * u][k & 3] : acc[2 * u + 1][k & 3];
          if (MODE == 0) {
            if (k & 1) part2 = fmaf(tv, xv[k], part2);
            else part = fmaf(tv, xv[k], part);
          } else {
            r8[k] = fmaf(tv, xv[k], bv[c][u][k]);
          }
        }
        if (MODE == 1 && live && c < cnt)
          *reinterpret_cast<uint4*>(out + ((b * P + p0 + c) * (int64_t)E) + 32 * u + 8 * q) = Vec16<bf16_t>::pack(r8);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_xj(xj[c], c, xb2);
      if (MODE == 0) {
        // sum over the four 16-lane rows (the h quarters of the sample in lane n) with two lane-permute instructions;
        // __shfl_xor goes through ds_bpermute and an lgkmcnt wait -- two LDS round trips per pair
        part += part2;
        const auto a2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
        const float h2 = __uint_as_float(a2[0]) + __uint_as_float(a2[1]);
        const auto c2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(h2), __float_as_uint(h2), false, false);
        part = __uint_as_float(c2[0]) + __uint_as_float(c2[1]);
        if (q == 0 && live && c < cnt) out[b * P + p0 + c] = from_f32<bf16_t>(part);
      }
    }
  };
  if (t_lo < t_hi) {
    load_xi(nxi[0], row_of(t_lo));
#pragma unroll
    for (int c = 0; c < PB_PPT; ++c) load_xj(nxj[0][c], c, row_of(t_lo));
    load_xi(nxi[1], row_of(t_lo + 1));
#pragma unroll
    for (int c = 0; c < PB_PPT; ++c) load_xj(nxj[1][c], c, row_of(t_lo + 1));
    int64_t t = t_lo;
    for (; t + 1 < t_hi; t += 2) {
      tile(t, nxi[0], nxj[0]);
      tile(t + 1, nxi[1], nxj[1]);
    }
    if (t < t_hi) tile(t, nxi[0], nxj[0]);
    // the clamped loads behind the last tiles are still landing: both sets stay allocated until they have (registers
    // the compiler considers dead are handed to the next address computation, and a late load would overwrite it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(nxi[s2][ks]));
#pragma unroll
      for (int c = 0; c < PB_PPT; ++c)
#pragma unroll
        for (int u = 0; u < KS; ++u) asm volatile("" ::"v"(nxj[s2][c][u]));
    }
  }
  }   // (task, tile range) segments of this wave
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
  hipStream_t s = (hipStream_t)stream;
#define TRS_PBM(KS_, MODE_)                                                                                         \
  do {                                                                                                              \
    auto kern = pair_bil_fwd_mfma_kernel<KS_, MODE_>;                                                               \
    static const int64_t slots = pbm_wave_slots((const void*)kern, 0);                                              \
    int grid;                                                                                                       \
    const int64_t per_wave = pbm_even_share((int64_t)ntasks * tiles, slots, &grid);                                 \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)Wt, (const bf16_t*)bias,  \
                       tasks, ntasks, per_wave, B, N, (bf16_t*)out);                                                \
  } while (0)
  if (E == 32) { if (mode == 0) TRS_PBM(1, 0); else TRS_PBM(1, 1); }
  else { if (mode == 0) TRS_PBM(2, 0); else TRS_PBM(2, 1); }
#undef TRS_PBM
  return check_launch("pair_bilinear_fwd_mfma");
}

// =============================================================================================================
// Backward of the per-pair bilinear form on the matrix cores.  With gv = g[b,p] (MODE 0) or g[b,p,h] (MODE 1):
//   dL/dT = gT[h] = gv * x_j[h]                     (needs no T)
//   gx_i[e]  = sum_{j>i} sum_h gT[h] W_p[e][h]      kernel XI : task (i, j-run), W_p resident (A rows e, k = h), the gT
//                                                   runs a lane builds ARE the B operand; accumulates over the 3 pairs
//   gx_j[h]  = sum_{i<j} gv * (x_i W_p)[h]          kernel XJ : task (j, i-run), W_p^T resident (A rows h, k = e);
//                                                   MODE 0 folds gv into the B operand and accumulates over the 3 pairs
//                                                   in the MFMA, MODE 1 multiplies the MFMA result by g[h] in registers
//   gW_p[e][h] = sum_b x_i[e] gT[h]                 kernel W  : task = one pair, K = samples: both operands go through a
//                                                   per-wave LDS transpose (2-byte stores, 16-byte fragment loads)
// XI / XJ write one contribution row per (sample, task) -- (B, ntasks, E) bf16, a third of the (B, NC2, E) tensors the
// GEMM route moves five times -- and a small kernel sums the tasks of each field into gx.
namespace trs {

template <int KS, int MODE>
__global__ __launch_bounds__(256) void pair_bil_bwd_xi_mfma_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ x,
                                                                   const bf16_t* __restrict__ W /* (P,E,H) */,
                                                                   const int32_t* __restrict__ tasks, int ntasks,
                                                                   int64_t per_wave, int64_t B, int N,
                                                                   bf16_t* __restrict__ contrib /* (B,ntasks,E) */) {
  constexpr int E = 32 * KS, ET = 2 * KS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
  const int P = N * (N - 1) / 2;
  // equal shares of the (task, sample tile) units, task-major (see pair_bil_fwd_mfma_kernel)
  const int64_t tiles = (B + 15) / 16;
  const int64_t flat_lo = (int64_t)(blockIdx.x * 4 + wave) * per_wave;
  const int64_t flat_hi = std::min<int64_t>(flat_lo + per_wave, (int64_t)ntasks * tiles);
  for (int64_t flat = flat_lo; flat < flat_hi;) {
  const int task = (int)(flat / tiles);
  const int64_t t_lo = flat - (int64_t)task * tiles, t_hi = std::min<int64_t>(tiles, t_lo + (flat_hi - flat));
  flat += t_hi - t_lo;
  const int fi = tasks[3 * task], j0 = tasks[3 * task + 1], cnt = tasks[3 * task + 2];
  const int p0 = pair_index_of(fi, j0, N);
  uint4 Wf[PB_PPT][ET][KS];           // A: row m of tile et <-> e = 32 (et>>1) + 8 (m>>2) + 4 (et&1) + (m&3); k = h
#pragma unroll
  for (int c = 0; c < PB_PPT; ++c) {
    const int pc = p0 + (c < cnt ? c : 0);
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      const int e = 32 * (et >> 1) + 8 * (n >> 2) + 4 * (et & 1) + (n & 3);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        Wf[c][et][ks] = *reinterpret_cast<const uint4*>(W + ((size_t)pc * E + e) * E + 32 * ks + 8 * q);
    }
  }
  for (int64_t t = t_lo; t < t_hi; ++t) {
    const int64_t b = t * 16 + n;
    const bool live = b < B;
    const int64_t bb = live ? b : 0;
    const bf16_t* xb = x + bb * (int64_t)N * E;
    pb_f32x4 acc[ET];
#pragma unroll
    for (int et = 0; et < ET; ++et) acc[et] = pb_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < PB_PPT; ++c) {
      if (c >= cnt) break;
      const float gs = MODE == 0 ? to_f32(g[bb * P + p0 + c]) : 0.f;
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        float xv[8], gv[8];
        Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(xb + (j0 + c) * E + 32 * u + 8 * q), xv);
        if (MODE == 1)
          Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(g + ((bb * P + p0 + c) * (int64_t)E) + 32 * u + 8 * q), gv);
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] *= MODE == 0 ? gs : gv[k];
        const uint4 bg = Vec16<bf16_t>::pack(xv);
#pragma unroll
        for (int et = 0; et < ET; ++et)
          acc[et] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pb_bf16x8, Wf[c][et][u]),
                                                            __builtin_bit_cast(pb_bf16x8, bg), acc[et], 0, 0, 0);
      }
    }
    if (live) {
#pragma unroll
      for (int v = 0; v < KS; ++v) {
        const float run[8] = {acc[2 * v][0], acc[2 * v][1], acc[2 * v][2], acc[2 * v][3],
                              acc[2 * v + 1][0], acc[2 * v + 1][1], acc[2 * v + 1][2], acc[2 * v + 1][3]};
        *reinterpret_cast<uint4*>(contrib + ((b * ntasks + task) * (int64_t)E) + 32 * v + 8 * q) = Vec16<bf16_t>::pack(run);
      }
    }
  }
  }   // segments
}

// tasks here are (j, i0, count): pairs (i0 .. i0+count-1, j)
template <int KS, int MODE>
__global__ __launch_bounds__(256) void pair_bil_bwd_xj_mfma_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ x,
                                                                   const bf16_t* __restrict__ Wt /* (P,H,E) */,
                                                                   const int32_t* __restrict__ tasks, int ntasks,
                                                                   int64_t per_wave, int64_t B, int N,
                                                                   bf16_t* __restrict__ contrib /* (B,ntasks,E) */) {
  constexpr int E = 32 * KS, MT = 2 * KS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
  const int P = N * (N - 1) / 2;
  const int64_t tiles = (B + 15) / 16;
  const int64_t flat_lo = (int64_t)(blockIdx.x * 4 + wave) * per_wave;
  const int64_t flat_hi = std::min<int64_t>(flat_lo + per_wave, (int64_t)ntasks * tiles);
  for (int64_t flat = flat_lo; flat < flat_hi;) {
  const int task = (int)(flat / tiles);
  const int64_t t_lo = flat - (int64_t)task * tiles, t_hi = std::min<int64_t>(tiles, t_lo + (flat_hi - flat));
  flat += t_hi - t_lo;
  const int fj = tasks[3 * task], i0 = tasks[3 * task + 1], cnt = tasks[3 * task + 2];
  uint4 Wf[PB_PPT][MT][KS];
  int pidx[PB_PPT];
#pragma unroll
  for (int c = 0; c < PB_PPT; ++c) {
    pidx[c] = pair_index_of(i0 + (c < cnt ? c : 0), fj, N);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int h = 32 * (mt >> 1) + 8 * (n >> 2) + 4 * (mt & 1) + (n & 3);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        Wf[c][mt][ks] = *reinterpret_cast<const uint4*>(Wt + ((size_t)pidx[c] * E + h) * E + 32 * ks + 8 * q);
    }
  }
  for (int64_t t = t_lo; t < t_hi; ++t) {
    const int64_t b = t * 16 + n;
    const bool live = b < B;
    const int64_t bb = live ? b : 0;
    const bf16_t* xb = x + bb * (int64_t)N * E;
    pb_f32x4 acc[MT];
    float sum[KS][8];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = pb_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KS; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) sum[u][k] = 0.f;
#pragma unroll
    for (int c = 0; c < PB_PPT; ++c) {
      if (c >= cnt) break;
      const float gs = MODE == 0 ? to_f32(g[bb * P + pidx[c]]) : 1.f;
      if (MODE == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = pb_f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        uint4 bx = *reinterpret_cast<const uint4*>(xb + (i0 + c) * E + 32 * ks + 8 * q);
        if (MODE == 0) {                 // fold the per-sample scalar into the B operand: the MFMA sums over the pairs
          float xv[8];
          Vec16<bf16_t>::unpack(bx, xv);
#pragma unroll
          for (int k = 0; k < 8; ++k) xv[k] *= gs;
          bx = Vec16<bf16_t>::pack(xv);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pb_bf16x8, Wf[c][mt][ks]),
                                                            __builtin_bit_cast(pb_bf16x8, bx), acc[mt], 0, 0, 0);
      }
      if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < KS; ++u) {
          float gv[8];
          Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(g + ((bb * P + pidx[c]) * (int64_t)E) + 32 * u + 8 * q), gv);
#pragma unroll
          for (int k = 0; k < 8; ++k)
            sum[u][k] = fmaf(gv[k], k < 4 ? acc[2 * u][k & 3] : acc[2 * u + 1][k & 3], sum[u][k]);
        }
      }
    }
    if (live) {
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        float run[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) run[k] = MODE == 1 ? sum[u][k] : (k < 4 ? acc[2 * u][k & 3] : acc[2 * u + 1][k & 3]);
        *reinterpret_cast<uint4*>(contrib + ((b * ntasks + task) * (int64_t)E) + 32 * u + 8 * q) = Vec16<bf16_t>::pack(run);
      }
    }
  }
  }   // segments
}

// gx[b][f][:] = sum of the XI contributions of field f's tasks + the XJ contributions of field f's tasks
__global__ __launch_bounds__(256) void pair_contrib_reduce_kernel(const bf16_t* __restrict__ ci, const int32_t* __restrict__ seg_i,
                                                                  int nti, const bf16_t* __restrict__ cj,
                                                                  const int32_t* __restrict__ seg_j, int ntj, int64_t B,
                                                                  int N, int E, bf16_t* __restrict__ gx) {
  const int vpr = E / 8;
  const int64_t total = B * N * vpr;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(t % vpr);
    const int64_t bf = t / vpr;
    const int f = (int)(bf % N);
    const int64_t b = bf / N;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int s = seg_i[f]; s < seg_i[f + 1]; ++s) {
      float c[8];
      Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(ci + ((b * nti + s) * (int64_t)E) + 8 * v), c);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += c[k];
    }
    for (int s = seg_j[f]; s < seg_j[f + 1]; ++s) {
      float c[8];
      Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(cj + ((b * ntj + s) * (int64_t)E) + 8 * v), c);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += c[k];
    }
    *reinterpret_cast<uint4*>(gx + bf * (int64_t)E + 8 * v) = Vec16<bf16_t>::pack(acc);
  }
}

// weight gradient: one pair per wave, K = samples through a per-wave LDS transpose, partial per sample split
template <int KS, int MODE>
__global__ __launch_bounds__(256) void pair_bil_bwd_w_mfma_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ x,
                                                                  int nsplit, int64_t B, int N,
                                                                  float* __restrict__ partial /* (nsplit,P,E,H) */) {
  constexpr int E = 32 * KS, ET = 2 * KS, TS = 80;       // TS: bytes per transposed row (32 samples + pad)
  extern __shared__ __attribute__((aligned(16))) char pbw_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
  const int P = N * (N - 1) / 2;
  char* xT = pbw_smem + (size_t)wave * 2 * E * TS;       // [e][32 samples]
  char* gT = xT + E * TS;                                // [h][32 samples]
  const int gw = blockIdx.x * 4 + wave;
  const int p = gw % P, split = gw / P;
  if (split >= nsplit) return;
  int fi, fj;
  pair_ij(p, N, &fi, &fj);
  pb_f32x4 acc[ET][ET];
#pragma unroll
  for (int a = 0; a < ET; ++a)
#pragma unroll
    for (int c = 0; c < ET; ++c) acc[a][c] = pb_f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t steps = (B + 31) / 32;
  const int64_t per = (steps + nsplit - 1) / nsplit;
  const int64_t s_lo = split * per, s_hi = std::min<int64_t>(s_lo + per, steps);
  for (int64_t st = s_lo; st < s_hi; ++st) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int64_t b = st * 32 + 16 * half + n;
      const bool live = b < B;
      const int64_t bb = live ? b : 0;
      const bf16_t* xb = x + bb * (int64_t)N * E;
      const float gs = (MODE == 0 && live) ? to_f32(g[bb * P + p]) : 0.f;
      const int col = 16 * half + n;
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        uint4 xi = *reinterpret_cast<const uint4*>(xb + fi * E + 32 * u + 8 * q);
        float xv[8], gv[8];
        Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(xb + fj * E + 32 * u + 8 * q), xv);
        if (MODE == 1)
          Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(g + ((bb * P + p) * (int64_t)E) + 32 * u + 8 * q), gv);
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = live ? xv[k] * (MODE == 0 ? gs : gv[k]) : 0.f;
        if (!live) xi = make_uint4(0, 0, 0, 0);
        const uint4 gt = Vec16<bf16_t>::pack(xv);
        const uint32_t wx[4] = {xi.x, xi.y, xi.z, xi.w}, wg[4] = {gt.x, gt.y, gt.z, gt.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          *reinterpret_cast<uint16_t*>(xT + (32 * u + 8 * q + k) * TS + col * 2) = (uint16_t)(wx[k >> 1] >> (16 * (k & 1)));
          *reinterpret_cast<uint16_t*>(gT + (32 * u + 8 * q + k) * TS + col * 2) = (uint16_t)(wg[k >> 1] >> (16 * (k & 1)));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    uint4 Bf[ET];
#pragma unroll
    for (int c = 0; c < ET; ++c) Bf[c] = *reinterpret_cast<const uint4*>(gT + (16 * c + n) * TS + q * 16);
#pragma unroll
    for (int a = 0; a < ET; ++a) {
      const uint4 Af = *reinterpret_cast<const uint4*>(xT + (16 * a + n) * TS + q * 16);
#pragma unroll
      for (int c = 0; c < ET; ++c)
        acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pb_bf16x8, Af),
                                                            __builtin_bit_cast(pb_bf16x8, Bf[c]), acc[a][c], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  float* mine = partial + ((size_t)split * P + p) * E * E;
#pragma unroll
  for (int a = 0; a < ET; ++a)
#pragma unroll
    for (int c = 0; c < ET; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(size_t)(16 * a + 4 * q + r) * E + 16 * c + n] = acc[a][c][r];
}

__global__ __launch_bounds__(256) void pair_w_reduce_kernel(const float* __restrict__ part, int nparts, int64_t n,
                                                            bf16_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
    out[i] = from_f32<bf16_t>(s);
  }
}

}  // namespace trs

using namespace trs;

/* Backward, data part.  tasks_i: (nti,3) = (i, j0, count) as in the forward; tasks_j: (ntj,3) = (j, i0, count): pairs
 * (i0..i0+count-1, j).  seg_i / seg_j: (N+1) int32 = first task of every field in the respective list.
 * contrib_i (B,nti,E), contrib_j (B,ntj,E): bf16 scratch.  W (NC2,E,E) [e][h], Wt its transpose [h][e].            */
extern "C" int trs_pair_bilinear_bwd_data_mfma(const void* g, const void* x, const void* W, const void* Wt,
                                               const int32_t* tasks_i, int32_t nti, const int32_t* seg_i,
                                               const int32_t* tasks_j, int32_t ntj, const int32_t* seg_j, int32_t mode,
                                               int64_t B, int32_t N, int32_t E, int32_t dtype, void* contrib_i,
                                               void* contrib_j, void* gx, trs_stream_t stream) {
  TRS_REQUIRE(B >= 0 && N >= 2 && nti > 0 && ntj > 0, TRS_EINVAL, "pair_bilinear_bwd_data_mfma: bad size");
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(g && x && W && Wt && tasks_i && tasks_j && seg_i && seg_j && contrib_i && contrib_j && gx, TRS_EINVAL,
              "pair_bilinear_bwd_data_mfma: NULL pointer");
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "pair_bilinear_bwd_data_mfma: bf16 only (dtype %d)", dtype);
  TRS_REQUIRE(E == 32 || E == 64, TRS_ESHAPE, "pair_bilinear_bwd_data_mfma: E = %d (32 or 64)", E);
  TRS_REQUIRE(mode == 0 || mode == 1, TRS_EINVAL, "pair_bilinear_bwd_data_mfma: mode %d", mode);
  TRS_REQUIRE(aligned16(x) && aligned16(W) && aligned16(Wt) && aligned16(contrib_i) && aligned16(contrib_j) &&
                  aligned16(gx) && (mode == 0 || aligned16(g)),
              TRS_EALIGN, "pair_bilinear_bwd_data_mfma: 16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  const int64_t tiles = (B + 15) / 16;
#define TRS_PBI(KS_, M_)                                                                                            \
  do {                                                                                                              \
    auto ki = pair_bil_bwd_xi_mfma_kernel<KS_, M_>;                                                                 \
    auto kj = pair_bil_bwd_xj_mfma_kernel<KS_, M_>;                                                                 \
    static const int64_t slots_i = pbm_wave_slots((const void*)ki, 0), slots_j = pbm_wave_slots((const void*)kj, 0); \
    int gi, gj;                                                                                                     \
    const int64_t pwi = pbm_even_share((int64_t)nti * tiles, slots_i, &gi);                                         \
    const int64_t pwj = pbm_even_share((int64_t)ntj * tiles, slots_j, &gj);                                         \
    hipLaunchKernelGGL(ki, dim3(gi), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)x, (const bf16_t*)W, tasks_i, \
                       nti, pwi, B, N, (bf16_t*)contrib_i);                                                         \
    hipLaunchKernelGGL(kj, dim3(gj), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)x, (const bf16_t*)Wt,         \
                       tasks_j, ntj, pwj, B, N, (bf16_t*)contrib_j);                                                \
  } while (0)
  if (E == 32) { if (mode == 0) TRS_PBI(1, 0); else TRS_PBI(1, 1); }
  else { if (mode == 0) TRS_PBI(2, 0); else TRS_PBI(2, 1); }
#undef TRS_PBI
  const int64_t total = B * N * (E / 8);
  hipLaunchKernelGGL(pair_contrib_reduce_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, s,
                     (const bf16_t*)contrib_i, seg_i, nti, (const bf16_t*)contrib_j, seg_j, ntj, B, N, E, (bf16_t*)gx);
  return check_launch("pair_bilinear_bwd_data_mfma");
}

extern "C" size_t trs_pair_bilinear_bwd_w_mfma_workspace_bytes(int64_t B, int32_t N, int32_t E) {
  if (B <= 0 || N < 2 || E <= 0) return 256;
  const int P = N * (N - 1) / 2;
  const int ns = pbm_splits((B + 31) / 32, P, 8, 8192);        // upper bound of what the launch picks (8 waves per SIMD)
  return (size_t)ns * P * E * E * 4 + 256;
}

/* Backward, weight part: gW (NC2,E,E) [e][h] bf16 = sum_b x[b,i_p,:]^T (gv * x[b,j_p,:]).                          */
extern "C" int trs_pair_bilinear_bwd_w_mfma(const void* g, const void* x, int32_t mode, int64_t B, int32_t N, int32_t E,
                                            int32_t dtype, void* gW, void* workspace, size_t ws_bytes,
                                            trs_stream_t stream) {
  TRS_REQUIRE(B >= 0 && N >= 2, TRS_EINVAL, "pair_bilinear_bwd_w_mfma: bad size");
  TRS_REQUIRE(g && x && gW && workspace, TRS_EINVAL, "pair_bilinear_bwd_w_mfma: NULL pointer");
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "pair_bilinear_bwd_w_mfma: bf16 only (dtype %d)", dtype);
  TRS_REQUIRE(E == 32 || E == 64, TRS_ESHAPE, "pair_bilinear_bwd_w_mfma: E = %d (32 or 64)", E);
  TRS_REQUIRE(mode == 0 || mode == 1, TRS_EINVAL, "pair_bilinear_bwd_w_mfma: mode %d", mode);
  TRS_REQUIRE(ws_bytes >= trs_pair_bilinear_bwd_w_mfma_workspace_bytes(B, N, E), TRS_EWORKSPACE,
              "pair_bilinear_bwd_w_mfma: workspace too small");
  const int P = N * (N - 1) / 2;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)P * E * E;
  if (B == 0) {
    return zero_bytes(gW, (size_t)n * 2, s);
  }
  const size_t lds = (size_t)4 * 2 * E * 80;
  float* part = (float*)workspace;
  int ns = 1;
#define TRS_PBW(KS_, M_)                                                                                            \
  do {                                                                                                              \
    auto kern = pair_bil_bwd_w_mfma_kernel<KS_, M_>;                                                                \
    static const int64_t slots = std::min<int64_t>(8192, pbm_wave_slots((const void*)kern, lds));                   \
    ns = pbm_splits((B + 31) / 32, P, 8, slots);                                                                    \
    hipLaunchKernelGGL(kern, dim3((int)(((int64_t)P * ns + 3) / 4)), dim3(256), lds, s, (const bf16_t*)g,            \
                       (const bf16_t*)x, ns, B, N, part);                                                           \
  } while (0)
  if (E == 32) { if (mode == 0) TRS_PBW(1, 0); else TRS_PBW(1, 1); }
  else { if (mode == 0) TRS_PBW(2, 0); else TRS_PBW(2, 1); }
#undef TRS_PBW
  hipLaunchKernelGGL(pair_w_reduce_kernel, dim3((int)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, part, ns,
                     n, (bf16_t*)gW);
  return check_launch("pair_bilinear_bwd_w_mfma");
}
