// Weight gradient of a dense layer over the rows of a batch:  dW (M x N) = g^T x,  g (rows, M) = dL/d(pre-activation),
// x (rows, N) = the layer's input, both bf16 row-major -- the contraction runs over the SLOW dimension of both operands
// (multilayer_perceptron.py:53-61 under autograd; rows = B for the deep branch of DeepFM / xDeepFM, B*N for the per-field
// stacks of DCN).  At 416 x 416 one row carries 1664 bytes and 346 kFLOP: 208 FLOP per byte, under the chip's ~310, so
// the kernel is bound by HBM as long as every operand byte is fetched once -- which is what the layout is built for.
//
// One workgroup owns a block of at most 14 x 16 output tiles of 16 x 16 (28 x 7 / 7 x 28 for narrow operands) and a
// contiguous range of rows; it keeps the whole block in MFMA accumulators and writes ONE fp32 partial at the end (summed
// over the row ranges and cast by trs_wgrad_finish).  Two forms of the same loop: four waves, one per SIMD on up to 512
// registers, 49 tiles each (waves as 2 x 2, 4 x 1 or 1 x 4), and eight waves, two per SIMD, as 2 x 4 with exact shares
// (see the kernel).  The workgroups that share a row range (2 x 2 blocks at 416 x 416) sit on the same XCD
// (blockIdx % 8) so that the second reader of a g / x piece finds it in that XCD's L2.
//
// Rows arrive 32 at a time (one MFMA k-step) through registers into a ring of four LDS slots, each
// [16-column panel][row][32 B]: both MFMA operands want the row index along K, which is what ds_read_b64_tr_b16 delivers
// from that image (lane i of a 16-lane group gets column i of a [4 rows][16 columns] block).  Odd lane groups take rows
// +4..7 first, so the two groups of a 32-lane half cover all 64 banks (the permutation of k is the same for both
// operands); 16 consecutive lanes load 256 contiguous bytes of a row and the 8 lanes of a ds_write_b128 group land on
// 4 panels x 32 B, which the panel stride (32 mod 128) spreads over the 32 banks.  The pipeline never drains: the
// fragments of step k+1 are fetched during the last rows of step k, step k+2 is written and step k+6 requested behind
// the second row of step k, one barrier per step.
//
// Measured (2.55 M rows, 416 x 416): 1.16 ms in the eight-wave form (1.28 ms with four waves of 49 tiles) against
// 1.37-1.42 ms for the batched hipBLASLt GEMM it replaces; the loads alone (MFMAs removed) take 0.90-0.96 ms = 4.7 TB/s of
// distinct bytes, the MFMAs alone 0.64 ms -- with one wave per SIMD the two overlap badly (every wait of the wave is a
// wait of the matrix pipe), and 13 % fewer MFMAs bought 13 % of the time at 384 columns; hence the second form.
// 416 x 64: 0.43 ms = 5.7 TB/s (0.52 ms).
#include <stdlib.h>

#include <algorithm>

#include "trs_common.hpp"

namespace trs {

typedef __attribute__((ext_vector_type(4))) short wg_s16x4;
typedef __attribute__((ext_vector_type(8))) short wg_s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 wg_bf16x8;
typedef __attribute__((ext_vector_type(4))) float wg_f32x4;

constexpr int WG_KR = 64;                  // row ranges are cut in units of two of these (four steps)
constexpr int WG_KS = 32;                  // rows per step
constexpr int WG_PANEL = WG_KS * 32 + 32;  // bytes from one 16-column panel of a step to the next (32 mod 128, below)
constexpr int WG_TC = 7;                   // output tiles per wave and direction, at most

struct WgradArgs {
  const uint16_t* g;
  const uint16_t* x;
  float* part;               // (S, M, N) fp32
  int64_t rows;
  int ldg, ldx;              // row strides (elements)
  int M, N;                  // columns of g / x that count (multiples of 8)
  int MB, NB;                // output blocks along M / N
  int slots_per_xcd;         // row ranges per XCD (S = 8 * slots_per_xcd)
};

__host__ __device__ __forceinline__ int split_start(int total, int parts, int k) { return (int)(((int64_t)total * k) / parts); }

// 8 k-values x this lane's column: two transpose reads (``lo`` / ``hi``: the lane's addresses of rows +0..3 / +4..7 -- swapped
// on odd lane groups --, ``off`` a compile-time byte offset that lands in the instruction's offset field)
__device__ __forceinline__ wg_bf16x8 tr_frag(const char* lo_p, const char* hi_p, int off) {
  typedef __attribute__((address_space(3))) wg_s16x4* lds_p;
  const wg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lo_p + off));
  const wg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(hi_p + off));
  const wg_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(wg_bf16x8, v);
}

// acc += A B, accumulator pinned to the accumulation registers and updated in place.  Written as an asm statement because
// with 49 tiles per wave hipcc otherwise gives the result a different register from the addend and moves 196 values
// back every iteration (3.7 register moves per MFMA).  The compiler still tracks the operands (it waits for the LDS reads
// that feed A and B); what it cannot know is that the statement is an MFMA: the accumulators are read only after the
// loop, behind explicit wait states.
__device__ __forceinline__ void wg_mfma(wg_f32x4& acc, const wg_bf16x8& A, const wg_bf16x8& B) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B));
}

// WM x WN waves: 4 (one per SIMD, every wave MC x NC tiles -- short shares run on into the neighbour's panels) or, for the
// 2 x 2 blocking of wide outputs, 8 as 2 x 4 (two per SIMD, 256 registers each: one wave's waits -- rows not there yet,
// LDS writes, the barrier -- sit under the other's MFMAs) with EXACT shares: a wave owns MC or MC-1 by NC or NC-1 tiles
// and runs the copy of the loop compiled for its share, so no MFMA is issued for nothing (13 x 13 tiles: 7|6 x 4|3|3|3,
// paired on the SIMDs as 21+18, 21+18, 21+24, 28+18).
template <int WM, int WN, int MC, int NC, bool CHECK>
__global__ __launch_bounds__(64 * WM * WN, 1) void wgrad_rows_kernel(WgradArgs a) {
  constexpr int NWAVES = WM * WN;
  constexpr bool EXACT = NWAVES == 8;
  extern __shared__ __attribute__((aligned(16))) char wg_lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, i = lane & 15;
  const bool odd = q & 1;

  // ---- which block, which rows
  const int TB = a.MB * a.NB;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int slot = xcd * a.slots_per_xcd + j / TB;
  const int S = 8 * a.slots_per_xcd;
  const int tile = j % TB;
  const int mb = tile / a.NB, nb = tile % a.NB;
  const int Mt = (a.M + 15) >> 4, Nt = (a.N + 15) >> 4;
  const int bm0 = split_start(Mt, a.MB, mb), bm1 = split_start(Mt, a.MB, mb + 1);      // block's tiles along M
  const int bn0 = split_start(Nt, a.NB, nb), bn1 = split_start(Nt, a.NB, nb + 1);
  const int PM = bm1 - bm0, PN = bn1 - bn0, P = PM + PN;
  // (8 waves: waves w and w+4 share a SIMD; 4..7 take the column shares rotated by one, so that the one or two large
  // shares of 13 or 14 columns -- 3,3,3,4 / 3,4,3,4 -- meet a small one: 28 + 18 MFMAs per step instead of 28 + 24)
  const int wm = wave / WN, wn = (EXACT && wave >= 4) ? ((wave + 1) & 3) : wave % WN;
  const int am0 = split_start(PM, WM, wm), mc = split_start(PM, WM, wm + 1) - am0;     // wave's tiles inside the block
  const int an0 = split_start(PN, WN, wn), nc = split_start(PN, WN, wn + 1) - an0;
  // the rows, in steps of 32 (one MFMA k-step); a row range is a multiple of four steps (the loop below has one exit:
  // a second one in the middle costs 200 spilled registers)
  const int64_t quads_total = (a.rows + 4 * WG_KS - 1) / (4 * WG_KS);
  const int64_t h0 = 4 * (quads_total * slot / S), h1 = 4 * (quads_total * (slot + 1) / S);
  const int64_t n_h = h1 - h0;

  // ---- staging: wave w stages panels 8w .. 8w+7 (256 bytes of a row), four rows per instruction, 16 consecutive lanes
  // reading 256 contiguous bytes; the 8 lanes of a ds_write_b128 group land on 4 panels x 32 B of one row, which the
  // panel stride (32 mod 128) spreads over all 32 banks
  constexpr int NLD = 32 / NWAVES;                               // 16-byte loads per thread and step: 8 (4 waves) or 4
  const int panel = (wave & 3) * 8 + ((lane & 15) >> 1);
  const int rlow = (wave >> 2) * 16 + (lane >> 4), half = lane & 1;   // 8 waves: waves 4..7 stage rows 16..31 of the step
  const bool is_g = panel < PM;
  const int col = is_g ? 16 * (bm0 + panel) + 8 * half : 16 * (bn0 + panel - PM) + 8 * half;
  const bool col_ok = panel < P && col + 8 <= (is_g ? a.M : a.N);
  const int ld = is_g ? a.ldg : a.ldx;
  const uint16_t* src = (is_g ? a.g : a.x) + col;
  const int lds_off = panel * WG_PANEL + rlow * 32 + half * 16;
  const int slot_bytes = P * WG_PANEL;                           // one 32-row step in LDS; four of them form a ring

  // four sets of 8 staging registers per thread, named one by one (an array here ends up in scratch memory): the rows
  // of steps h+3 .. h+6 are on their way while step h is multiplied -- ~100 KB per CU in flight
  const int64_t rstep = (int64_t)4 * ld * 2;                    // bytes between this lane's rows of one step
  uint4 sa0, sa1, sa2, sa3, sa4, sa5, sa6, sa7, sb0, sb1, sb2, sb3, sb4, sb5, sb6, sb7;
  uint4 sc0, sc1, sc2, sc3, sc4, sc5, sc6, sc7, sd0, sd1, sd2, sd3, sd4, sd5, sd6, sd7;
#define WG_EACH(X, R) X(R, 0) X(R, 1) X(R, 2) X(R, 3) X(R, 4) X(R, 5) X(R, 6) X(R, 7)
#define WG_LD(R, t) \
  if (t < NLD) R##t = *reinterpret_cast<const uint4*>(p + t * rstep);
#define WG_LDC(R, t)  \
  if (t < NLD)        \
    R##t = (h_ * WG_KS + rlow + 4 * t < a.rows) ? *reinterpret_cast<const uint4*>(p + t * rstep) : make_uint4(0, 0, 0, 0);
#define WG_WR(R, t) \
  if (t < NLD) *reinterpret_cast<uint4*>(dst + 4 * t * 32) = R##t;
#define WG_WR0(R, t) \
  if (t < NLD) *reinterpret_cast<uint4*>(dst + 4 * t * 32) = make_uint4(0, 0, 0, 0);
  // no branches around the loads (they would turn the staging registers into merge points the allocator handles
  // badly): a lane whose columns lie outside its operand reads the operand's first columns instead and never writes
  // them -- its LDS pieces are zeroed once, here.  CHECK (rows not a multiple of 128): rows past the end read as zero.
  const char* src_b = reinterpret_cast<const char*>(col_ok ? src : (is_g ? a.g : a.x));
  const bool writer = panel < P && col_ok;
  if (panel < P && !col_ok) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      char* dst = wg_lds + b * slot_bytes + lds_off;
      WG_EACH(WG_WR0, z)
    }
  }
  // the k-th step this workgroup works on (past the end: the last one again -- loaded, never used)
  auto step_of = [&](int64_t k) { return h0 + (k < n_h ? k : n_h - 1); };
#define WG_LOAD(R, k)                                                 \
  {                                                                   \
    const int64_t h_ = step_of(k);                                    \
    const char* p = src_b + (h_ * WG_KS + rlow) * ld * 2;             \
    if (CHECK) {                                                      \
      WG_EACH(WG_LDC, R)                                              \
    } else {                                                          \
      WG_EACH(WG_LD, R)                                               \
    }                                                                 \
  }
#define WG_WRITE(R, b)                                 \
  {                                                    \
    char* dst = wg_lds + (b) * slot_bytes + lds_off;   \
    if (writer) {                                      \
      WG_EACH(WG_WR, R)                                \
    }                                                  \
  }

  // everything from here on is compiled once per share (MCx x NCx tiles): once for the 4-wave form, four times for EXACT
  auto run = [&]<int MCx, int NCx>() __attribute__((always_inline)) {
    wg_f32x4 acc[MCx][NCx];
  #pragma unroll
    for (int m = 0; m < MCx; ++m)
  #pragma unroll
      for (int n = 0; n < NCx; ++n) acc[m][n] = wg_f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane fragment addresses.  Every wave multiplies MCx x NCx tiles; a wave that owns fewer runs on into the panels
    // behind its own (same cost for the workgroup -- the SIMD with the full share sets the pace --, no branches in the
    // loop, and every read is one base register plus an immediate; the results are dropped at the end)
    const int frag_lane = (8 * q + (i >> 2)) * 32 + (i & 3) * 8;
    const int a_lo = am0 * WG_PANEL + frag_lane + (odd ? 128 : 0), a_hi = am0 * WG_PANEL + frag_lane + (odd ? 0 : 128);
    const int b_lo = (PM + an0) * WG_PANEL + frag_lane + (odd ? 128 : 0);
    const int b_hi = (PM + an0) * WG_PANEL + frag_lane + (odd ? 0 : 128);

    // One 32-row step out of ring slot ``c`` with the fragments of the NEXT step (ring slot ``nx``) fetched on the way:
    // the A fragment of row m+2 (of this step or the next) before the MFMAs of row m, the B fragments in place behind
    // the last row -- so a step starts with its operands in registers.  ``mid`` runs behind the second row: the LDS
    // writes and global loads of the steps further ahead.
    constexpr int PD = MCx >= 2 ? 2 : 1;
    wg_bf16x8 Ar[4], Bf[NCx];                         // A: a ring of four, row m of step c in slot (MCx * c + m) % 4
  #define WG_STEP(c, nx, MID)                                                                                       \
    {                                                                                                               \
      const char* cb = wg_lds + (c) * slot_bytes;                                                                   \
      const char* nb_ = wg_lds + (nx) * slot_bytes;                                                                 \
      constexpr int RO = (MCx * (c)) % 4;                                                                            \
      _Pragma("unroll") for (int m = 0; m < MCx; ++m) {                                                              \
        const int t = (m + PD) % MCx;                                                                                \
        const char* fb = (m + PD < MCx) ? cb : nb_;                                                                  \
        Ar[(RO + m + PD) % 4] = tr_frag(fb + a_lo, fb + a_hi, t * WG_PANEL);                                        \
        _Pragma("unroll") for (int n = 0; n < NCx; ++n) {                                                            \
          wg_mfma(acc[m][n], Ar[(RO + m) % 4], Bf[n]);                                                              \
          if (m == MCx - 1) Bf[n] = tr_frag(nb_ + b_lo, nb_ + b_hi, n * WG_PANEL);                                   \
        }                                                                                                           \
        if (m == (MCx > 2 ? 1 : 0)) { MID }                                                                          \
      }                                                                                                             \
    }

    if (n_h > 0) {
      WG_LOAD(sa, 0)
      WG_LOAD(sb, 1)
      WG_WRITE(sa, 0)
      WG_WRITE(sb, 1)
      WG_LOAD(sc, 2)
      WG_LOAD(sd, 3)
      WG_LOAD(sa, 4)
      WG_LOAD(sb, 5)
      __syncthreads();
      {
        const char* cb = wg_lds;
  #pragma unroll
        for (int n = 0; n < NCx; ++n) Bf[n] = tr_frag(cb + b_lo, cb + b_hi, n * WG_PANEL);
  #pragma unroll
        for (int m = 0; m < PD; ++m) Ar[m] = tr_frag(cb + a_lo, cb + a_hi, m * WG_PANEL);
      }
      // ring slot k % 4 holds step k, register set j % 4 the rows of step j on their way.  At step k: step k+2 is written
      // (loaded four steps ago) and the loads of step k+6 leave into the same registers; one barrier per step: slot k+2
      // complete, slot k free.  (Requesting before the wait, into the set written out a step earlier, measured the same at
      // 416 x 416 and 15 % slower at 416 x 2496.)
      for (int64_t k = 0; k < n_h; k += 4) {
        WG_STEP(0, 1, WG_WRITE(sc, 2) WG_LOAD(sc, k + 6))
        __syncthreads();
        WG_STEP(1, 2, WG_WRITE(sd, 3) WG_LOAD(sd, k + 7))
        __syncthreads();
        WG_STEP(2, 3, WG_WRITE(sa, 0) WG_LOAD(sa, k + 8))
        __syncthreads();
        WG_STEP(3, 0, WG_WRITE(sb, 1) WG_LOAD(sb, k + 9))
        __syncthreads();
      }
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results are not visible before this
    // ---- the block's partial: D[m = 4q + e][n = i]
    float* out = a.part + (int64_t)slot * a.M * a.N;
  #pragma unroll
    for (int m = 0; m < MCx; ++m)
  #pragma unroll
      for (int n = 0; n < NCx; ++n)
        if (m < mc && n < nc) {
          const int row = 16 * (bm0 + am0 + m) + 4 * q, c = 16 * (bn0 + an0 + n) + i;
  #pragma unroll
          for (int e = 0; e < 4; ++e)
            if (row + e < a.M && c < a.N) out[(int64_t)(row + e) * a.N + c] = acc[m][n][e];
        }

  };
  if constexpr (!EXACT) {
    run.template operator()<MC, NC>();
  } else {
    // the four shares of the 2 x 4 form; every copy holds the same number of barriers
    if (mc == MC && nc == NC) run.template operator()<MC, NC>();
    else if (mc == MC) run.template operator()<MC, NC - 1>();
    else if (nc == NC) run.template operator()<MC - 1, NC>();
    else run.template operator()<MC - 1, NC - 1>();
  }
}

// ------------------------------------------------------------------------------------------------ LDS-DMA staging (round 5)
// The same 13 x 13-tile blocks, eight waves with exact shares and pinned accumulators, but the rows reach LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write) and the ring is six steps deep.  The image of a step is
// ROW-MAJOR per operand block -- [32 rows][208 columns] = 13 312 bytes, i.e. 13 pieces of 1 KB that a DMA instruction
// fills contiguously (lane l of piece j brings bytes 1024 j + 16 l of the image: 2.5 rows of 416 contiguous bytes each)
// -- and the transposing fragment reads work on it as they stand: a row pitch of 416 bytes is 32 mod 128, which spreads
// the four rows of a 16-lane group and, with the odd groups' +4-row rotation, the two groups of a half wave over all 64
// banks, exactly as the panel layout above does.  Step k lives in ring slot k % 6; its copy is issued five steps ahead
// (during step k-5, behind the barrier that freed the slot), every wave waits for ITS pieces of step k+1 before the
// barrier that ends step k-1 (three later steps' pieces may still be in flight: s_waitcnt vmcnt(9 | 12)), so four steps
// = ~100 KB per CU are under way at any time -- what the register-staged form keeps in flight with 64 VGPRs per thread.
// Used when both operands cover whole 13-tile blocks in memory (ld >= 208 columns behind every block start) and the rows
// are a multiple of 128; TRS_WGRAD_DMA=0 keeps the register-staged kernel.
constexpr int WD_S = 416;                 // bytes per image row (208 columns)
constexpr int WD_IMG = WG_KS * WD_S;      // one operand block of one step
constexpr int WD_SLOT = 2 * WD_IMG;
constexpr int WD_NSLOT = 6;
constexpr int WD_AHEAD = 5;

__device__ __forceinline__ void wd_dma(const char* src, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(dst), "s"(src) : "memory");
}

template <int MC, int NC>
__global__ __launch_bounds__(512, 1) void wgrad_dma_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wg_lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, i = lane & 15;
  const bool odd = q & 1;
  const int TB = a.MB * a.NB;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int slot = xcd * a.slots_per_xcd + j / TB;
  const int S = 8 * a.slots_per_xcd;
  const int tile = j % TB;
  const int mb = tile / a.NB, nb = tile % a.NB;
  const int Mt = (a.M + 15) >> 4, Nt = (a.N + 15) >> 4;
  const int bm0 = split_start(Mt, a.MB, mb), bm1 = split_start(Mt, a.MB, mb + 1);
  const int bn0 = split_start(Nt, a.NB, nb), bn1 = split_start(Nt, a.NB, nb + 1);
  const int PM = bm1 - bm0, PN = bn1 - bn0;
  const int wm = wave >> 2, wn = wave >= 4 ? ((wave + 1) & 3) : (wave & 3);
  const int am0 = split_start(PM, 2, wm), mc = split_start(PM, 2, wm + 1) - am0;
  const int an0 = split_start(PN, 4, wn), nc = split_start(PN, 4, wn + 1) - an0;
  const int64_t quads_total = a.rows / (4 * WG_KS);      // (rows % 128 == 0: checked by the host)
  const int64_t h0 = 4 * (quads_total * slot / S), h1 = 4 * (quads_total * (slot + 1) / S);
  const int64_t n_h = h1 - h0;

  // ---- this wave's pieces of a step: p = wave + 8 t (t = 0..3, p < 26); pieces 0..12 the g block, 13..25 the x block
  const char* gcol = reinterpret_cast<const char*>(a.g + 16 * bm0);
  const char* xcol = reinterpret_cast<const char*>(a.x + 16 * bn0);
  unsigned voff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int p = wave + 8 * t, op = p >= 13, jj = op ? p - 13 : p;
    const int o = jj * 1024 + 16 * lane, row = o / WD_S, cb = o - row * WD_S;
    voff[t] = (unsigned)(row * (op ? a.ldx : a.ldg) * 2 + cb);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)wg_lds;
  auto issue = [&](int64_t k, unsigned ring_slot) __attribute__((always_inline)) {      // the copy of step k into ring slot ``ring_slot``
    const int64_t h = h0 + (k < n_h ? k : n_h - 1);      // (past the end: the last step again -- landed, never read)
    const char* gs = gcol + h * WG_KS * a.ldg * 2;
    const char* xs = xcol + h * WG_KS * a.ldx * 2;
    const unsigned base = lds0 + ring_slot * WD_SLOT;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int p = wave + 8 * t;
      if (p < 26) wd_dma(p >= 13 ? xs : gs, voff[t], base + (p >= 13 ? WD_IMG + (p - 13) * 1024 : p * 1024));
    }
  };
  // this wave's pieces of all steps but the three youngest have landed (then: barrier -> visible to every wave)
  auto landed = [&]() __attribute__((always_inline)) {
    if (wave < 2) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  auto run = [&]<int MCx, int NCx>() __attribute__((always_inline)) {
    wg_f32x4 acc[MCx][NCx];
#pragma unroll
    for (int m = 0; m < MCx; ++m)
#pragma unroll
      for (int n = 0; n < NCx; ++n) acc[m][n] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
    const int frag_lane = (8 * q + (i >> 2)) * WD_S + (i & 3) * 8;
    const int a_lo = am0 * 32 + frag_lane + (odd ? 4 * WD_S : 0), a_hi = am0 * 32 + frag_lane + (odd ? 0 : 4 * WD_S);
    const int b_lo = WD_IMG + an0 * 32 + frag_lane + (odd ? 4 * WD_S : 0);
    const int b_hi = WD_IMG + an0 * 32 + frag_lane + (odd ? 0 : 4 * WD_S);
    constexpr int PD = MCx >= 2 ? 2 : 1;
    wg_bf16x8 Ar[4], Bf[NCx];
    unsigned cur = 0;      // ring slot of the step being multiplied
    auto nxt_of = [](unsigned c) { return c + 1 == WD_NSLOT ? 0u : c + 1; };
#define WD_STEP(c4, KEXPR)                                                                                          \
    {                                                                                                               \
      const unsigned nx = nxt_of(cur);                                                                              \
      const char* cb = wg_lds + cur * WD_SLOT;                                                                      \
      const char* nb_ = wg_lds + nx * WD_SLOT;                                                                      \
      constexpr int RO = (MCx * (c4)) % 4;                                                                          \
      _Pragma("unroll") for (int m = 0; m < MCx; ++m) {                                                              \
        const int t = (m + PD) % MCx;                                                                                \
        const char* fb = (m + PD < MCx) ? cb : nb_;                                                                  \
        Ar[(RO + m + PD) % 4] = tr_frag(fb + a_lo, fb + a_hi, t * 32);                                              \
        _Pragma("unroll") for (int n = 0; n < NCx; ++n) {                                                            \
          wg_mfma(acc[m][n], Ar[(RO + m) % 4], Bf[n]);                                                              \
          if (m == MCx - 1) Bf[n] = tr_frag(nb_ + b_lo, nb_ + b_hi, n * 32);                                         \
        }                                                                                                           \
        if (m == (MCx > 2 ? 1 : 0)) {                                                                                \
          /* slot (cur + 5) % 6 = (cur - 1) % 6 was freed by the barrier that ended the previous step */           \
          issue((KEXPR) + WD_AHEAD, cur == 0 ? WD_NSLOT - 1 : cur - 1);                                              \
        }                                                                                                           \
      }                                                                                                             \
      landed();                                                                                                     \
      cur = nx;                                                                                                     \
    }
    if (n_h > 0) {
#pragma unroll
      for (int k = 0; k < WD_AHEAD; ++k) issue(k, k);
      landed();      // steps 0 and 1 are in LDS (2, 3, 4 may still be on their way)
      {
        const char* cb = wg_lds;
#pragma unroll
        for (int n = 0; n < NCx; ++n) Bf[n] = tr_frag(cb + b_lo, cb + b_hi, n * 32);
#pragma unroll
        for (int m = 0; m < PD; ++m) Ar[m] = tr_frag(cb + a_lo, cb + a_hi, m * 32);
      }
      for (int64_t k = 0; k < n_h; k += 4) {
        WD_STEP(0, k)
        WD_STEP(1, k + 1)
        WD_STEP(2, k + 2)
        WD_STEP(3, k + 3)
      }
    }
#undef WD_STEP
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // copies still landing; the last MFMAs' results
    float* out = a.part + (int64_t)slot * a.M * a.N;
#pragma unroll
    for (int m = 0; m < MCx; ++m)
#pragma unroll
      for (int n = 0; n < NCx; ++n)
        if (m < mc && n < nc) {
          const int row = 16 * (bm0 + am0 + m) + 4 * q, c = 16 * (bn0 + an0 + n) + i;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (row + e < a.M && c < a.N) out[(int64_t)(row + e) * a.N + c] = acc[m][n][e];
        }
  };
  if (mc == MC && nc == NC) run.template operator()<MC, NC>();
  else if (mc == MC) run.template operator()<MC, NC - 1>();
  else if (nc == NC) run.template operator()<MC - 1, NC>();
  else run.template operator()<MC - 1, NC - 1>();
}

// ---- the same, FOUR waves on a 13 x 26-tile block (one g block against both x blocks of a 400..416-wide x): a wave owns
// all 13 (12) tile rows and 7 | 6 tile columns -- 91 accumulator tiles on 364 of its 512 registers -- so a fragment read
// from LDS feeds 7 or 13 MFMAs instead of 4 or 7: the eight-wave form above reads 88 KB of fragments per 169-tile step,
// which is as many LDS cycles (256 B per clock) as the step has MFMA cycles per SIMD, and the two do not overlap
// perfectly; here it is 80 KB per 338-tile step.  The g block is also fetched once instead of twice.  Three images per
// step (g, x block 0, x block 1: 39 pieces), ring of four slots, copies three steps ahead.  For long row ranges
// (rows >= WD2_MIN_ROWS: twice as many fp32 partials as the eight-wave form for the same number of workgroups).
constexpr int WD2_SLOT = 3 * WD_IMG;
constexpr int WD2_NSLOT = 4;
constexpr int WD2_AHEAD = 3;
constexpr int64_t WD2_MIN_ROWS = 1 << 20;

template <bool AG>
__device__ __forceinline__ void wd_mfma(wg_f32x4& acc, const wg_bf16x8& A, const wg_bf16x8& B) {
  if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(A), "v"(B));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
}

__global__ __launch_bounds__(256, 1) void wgrad_dma2_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char wg_lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, i = lane & 15;
  const bool odd = q & 1;
  const int TB = a.MB;                                   // (a.NB == 2: both x blocks belong to the workgroup)
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int slot = xcd * a.slots_per_xcd + j / TB;
  const int S = 8 * a.slots_per_xcd;
  const int mb = j % TB;
  const int Mt = (a.M + 15) >> 4, Nt = (a.N + 15) >> 4;
  const int bm0 = split_start(Mt, a.MB, mb), PM = split_start(Mt, a.MB, mb + 1) - bm0;
  const int xb = wave >> 1;                              // this wave's x block
  const int bn0 = split_start(Nt, 2, xb), PN = split_start(Nt, 2, xb + 1) - bn0;
  const int an0 = split_start(PN, 2, wave & 1), nc = split_start(PN, 2, (wave & 1) + 1) - an0;
  const int xc0 = 16 * split_start(Nt, 2, 0), xc1 = 16 * split_start(Nt, 2, 1);
  const int64_t quads_total = a.rows / (4 * WG_KS);
  const int64_t h0 = 4 * (quads_total * slot / S), h1 = 4 * (quads_total * (slot + 1) / S);
  const int64_t n_h = h1 - h0;

  // pieces p = wave + 4 t (t = 0..9, p < 39): image p / 13 (g, x block 0, x block 1), KB p % 13 of it
  const char* gcol = reinterpret_cast<const char*>(a.g + 16 * bm0);
  const char* x0col = reinterpret_cast<const char*>(a.x + xc0);
  const char* x1col = reinterpret_cast<const char*>(a.x + xc1);
  unsigned voff[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    const int p = wave + 4 * t, img = p / 13, jj = p - 13 * img;
    const int o = jj * 1024 + 16 * lane, row = o / WD_S, cb = o - row * WD_S;
    voff[t] = (unsigned)(row * (img ? a.ldx : a.ldg) * 2 + cb);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)wg_lds;
  auto issue = [&](int64_t k, unsigned ring_slot) __attribute__((always_inline)) {
    const int64_t h = h0 + (k < n_h ? k : n_h - 1);
    const char* gs = gcol + h * WG_KS * a.ldg * 2;
    const char* x0s = x0col + h * WG_KS * a.ldx * 2;
    const char* x1s = x1col + h * WG_KS * a.ldx * 2;
    const unsigned base = lds0 + ring_slot * WD2_SLOT;
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      const int p = wave + 4 * t;
      if (p < 39) wd_dma(p < 13 ? gs : (p < 26 ? x0s : x1s), voff[t], base + p * 1024);
    }
  };
  auto landed = [&]() __attribute__((always_inline)) {      // all but the youngest step's pieces of this wave
    if (wave < 3) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  auto run = [&]<int MCx, int NCx>() __attribute__((always_inline)) {
    wg_f32x4 acc[MCx][NCx];
#pragma unroll
    for (int m = 0; m < MCx; ++m)
#pragma unroll
      for (int n = 0; n < NCx; ++n) acc[m][n] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
    const int frag_lane = (8 * q + (i >> 2)) * WD_S + (i & 3) * 8;
    const int a_lo = frag_lane + (odd ? 4 * WD_S : 0), a_hi = frag_lane + (odd ? 0 : 4 * WD_S);
    const int b_lo = WD_IMG * (1 + xb) + an0 * 32 + frag_lane + (odd ? 4 * WD_S : 0);
    const int b_hi = WD_IMG * (1 + xb) + an0 * 32 + frag_lane + (odd ? 0 : 4 * WD_S);
    constexpr int PD = 2;
    wg_bf16x8 Ar[4], Bf[NCx];
#define WD2_STEP(c4, KEXPR)                                                                                         \
    {                                                                                                               \
      const char* cb = wg_lds + (c4) * WD2_SLOT;                                                                    \
      const char* nb_ = wg_lds + (((c4) + 1) % WD2_NSLOT) * WD2_SLOT;                                               \
      constexpr int RO = (MCx * (c4)) % 4;                                                                          \
      _Pragma("unroll") for (int m = 0; m < MCx; ++m) {                                                              \
        const int t = (m + PD) % MCx;                                                                                \
        const char* fb = (m + PD < MCx) ? cb : nb_;                                                                  \
        Ar[(RO + m + PD) % 4] = tr_frag(fb + a_lo, fb + a_hi, t * 32);                                              \
        _Pragma("unroll") for (int n = 0; n < NCx; ++n) {                                                            \
          if (m * NCx + n < 64) wd_mfma<true>(acc[m][n], Ar[(RO + m) % 4], Bf[n]);                                   \
          else wd_mfma<false>(acc[m][n], Ar[(RO + m) % 4], Bf[n]);                                                   \
          if (m == MCx - 1) Bf[n] = tr_frag(nb_ + b_lo, nb_ + b_hi, n * 32);                                         \
        }                                                                                                           \
        if (m == 1) issue((KEXPR) + WD2_AHEAD, ((c4) + WD2_AHEAD) % WD2_NSLOT);      /* the slot the last barrier freed */ \
      }                                                                                                             \
      landed();                                                                                                     \
    }
    if (n_h > 0) {
#pragma unroll
      for (int k = 0; k < WD2_AHEAD; ++k) issue(k, k);
      landed();
      {
        const char* cb = wg_lds;
#pragma unroll
        for (int n = 0; n < NCx; ++n) Bf[n] = tr_frag(cb + b_lo, cb + b_hi, n * 32);
#pragma unroll
        for (int m = 0; m < PD; ++m) Ar[m] = tr_frag(cb + a_lo, cb + a_hi, m * 32);
      }
      for (int64_t k = 0; k < n_h; k += 4) {
        WD2_STEP(0, k)
        WD2_STEP(1, k + 1)
        WD2_STEP(2, k + 2)
        WD2_STEP(3, k + 3)
      }
    }
#undef WD2_STEP
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float* out = a.part + (int64_t)slot * a.M * a.N;
#pragma unroll
    for (int m = 0; m < MCx; ++m)
#pragma unroll
      for (int n = 0; n < NCx; ++n)
        if (m < PM && n < nc) {
          const int row = 16 * (bm0 + m) + 4 * q, c = 16 * (bn0 + an0 + n) + i;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (row + e < a.M && c < a.N) out[(int64_t)(row + e) * a.N + c] = acc[m][n][e];
        }
  };
  // (a 12-tile g block runs the 13-row copy on one tile of whatever follows the block in memory: results dropped)
  if (nc == 7) run.template operator()<13, 7>();
  else run.template operator()<13, 6>();
}

static const bool WGRAD_DMA = [] {
  const char* e = getenv("TRS_WGRAD_DMA");
  return !(e && e[0] == '0');
}();
static const bool WGRAD_DMA2 = [] {
  const char* e = getenv("TRS_WGRAD_DMA2");
  return !(e && e[0] == '0');
}();

static const bool WGRAD_EIGHT = [] {
  const char* e = getenv("TRS_WGRAD_EIGHT");
  return !(e && e[0] == '0');
}();

struct WgradPlan {
  int wm, wn, mc, nc, MB, NB, slots_per_xcd;
  int dma = 0;      // 0: register-staged kernel; 1: wgrad_dma_kernel (eight waves, 13 x 13 tiles); 2: wgrad_dma2_kernel (four waves, 13 x 26)
};

static inline int tile_class(int t) { return t <= 1 ? 1 : (t <= 4 ? 4 : WG_TC); }

// slots_per_xcd == 0: not handled
static WgradPlan wgrad_plan(int M, int N, int64_t rows) {
  WgradPlan p{0, 0, 0, 0, 0, 0, 0};
  if (M < 8 || N < 8 || (M & 7) || (N & 7) || rows < 4 * WG_KR) return p;
  const int Mt = (M + 15) / 16, Nt = (N + 15) / 16;
  if (Mt <= 4 * WG_TC && Nt <= WG_TC && Mt + Nt <= 32 && Mt >= Nt) {
    p = WgradPlan{4, 1, WG_TC, Nt <= 4 ? 4 : WG_TC, 1, 1, 0};
  } else if (Mt <= WG_TC && Nt <= 4 * WG_TC && Mt + Nt <= 32) {
    p = WgradPlan{1, 4, tile_class(Mt), WG_TC, 1, 1, 0};
  } else {
    p = WgradPlan{2, 2, WG_TC, WG_TC, (Mt + 2 * WG_TC - 1) / (2 * WG_TC), (Nt + 2 * WG_TC - 1) / (2 * WG_TC), 0};
    while (!is_pow2(p.MB)) ++p.MB;
    while (!is_pow2(p.NB)) ++p.NB;
    if (p.MB * p.NB > 32) return WgradPlan{0, 0, 0, 0, 0, 0, 0};
    // tiles per wave along N once the blocks are cut (2496 columns in 16 blocks: 9 | 10 tiles, 5 per wave)
    if (((Nt + p.NB - 1) / p.NB + 1) / 2 <= 5) p.nc = 5;
    // blocks of 12..14 x 12..16 tiles (416 x 416: 13 x 13): eight waves with exact shares of 7|6 x 4|3 tiles
    const int pm_lo = Mt / p.MB, pm_hi = (Mt + p.MB - 1) / p.MB, pn_lo = Nt / p.NB, pn_hi = (Nt + p.NB - 1) / p.NB;
    if (WGRAD_EIGHT && pm_lo >= 12 && pm_hi <= 14 && pn_lo >= 12 && pn_hi <= 16) {
      p.wn = 4;
      p.mc = 7;
      p.nc = 4;
    } else if (WGRAD_EIGHT && pm_lo >= 12 && pm_hi <= 14 && pn_lo >= 8 && pn_hi <= 12) {
      p.wn = 4;      // 2496 columns in 16 blocks of 9 | 10 tiles: shares of 3 | 2
      p.mc = 7;
      p.nc = 3;
    }
  }
  // the LDS-DMA forms: 2 x 4 waves on 13 x 13-tile blocks whose 208-column images exist inside the M / N columns the caller
  // vouches for, rows a multiple of 128; the four-wave 13 x 26 form for long row ranges against a two-block x
  auto blocks_ok = [](int T, int B, int cols) {
    for (int b = 0; b < B; ++b) {
      const int t0 = split_start(T, B, b), t1 = split_start(T, B, b + 1);
      if (t1 - t0 < 12 || t1 - t0 > 13 || 16 * t0 + 208 > cols) return false;
    }
    return true;
  };
  if (WGRAD_DMA && p.wm == 2 && p.wn == 4 && p.nc == 4 && rows % (4 * WG_KS) == 0 && blocks_ok(Mt, p.MB, M) &&
      blocks_ok(Nt, p.NB, N))
    p.dma = (WGRAD_DMA2 && p.NB == 2 && rows >= WD2_MIN_ROWS) ? 2 : 1;
  int slots = 32 / (p.dma == 2 ? p.MB : p.MB * p.NB);
  // every row range at least 4 stages long
  const int64_t stages = (rows + WG_KR - 1) / WG_KR;
  while (slots > 1 && stages / (8 * slots) < 4) slots >>= 1;
  p.slots_per_xcd = slots;
  return p;
}

template <int WM, int WN, int MC, int NC, bool CHECK>
int wgrad_launch_c(const WgradArgs& a, int grid, size_t lds, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)wgrad_rows_kernel<WM, WN, MC, NC, CHECK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return check_launch("wgrad_rows: LDS attribute");
    attr = true;
  }
  hipLaunchKernelGGL((wgrad_rows_kernel<WM, WN, MC, NC, CHECK>), dim3(grid), dim3(64 * WM * WN), lds, s, a);
  return check_launch("wgrad_rows");
}

}  // namespace trs

using namespace trs;

template <int WM, int WN, int MC, int NC>
static int wgrad_launch(const WgradArgs& a, int grid, size_t lds, hipStream_t s) {
  return (a.rows % (4 * WG_KS)) ? wgrad_launch_c<WM, WN, MC, NC, true>(a, grid, lds, s)
                          : wgrad_launch_c<WM, WN, MC, NC, false>(a, grid, lds, s);
}

extern "C" int32_t trs_wgrad_rows_splits(int32_t M, int32_t N, int64_t rows) {
  const WgradPlan p = wgrad_plan(M, N, rows);
  return 8 * p.slots_per_xcd;
}

extern "C" int trs_wgrad_rows(const void* g, int32_t ldg, const void* x, int32_t ldx, int64_t rows, int32_t M, int32_t N,
                              int32_t dtype, int32_t S, float* part, trs_stream_t stream) {
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "wgrad_rows: bf16 operands only");
  WgradPlan p = wgrad_plan(M, N, rows);
  // S = what trs_wgrad_rows_splits returned, or that number halved any number of times down to 8 (longer row ranges on
  // fewer workgroups: two weight gradients enqueued on two streams then share the chip, see layers._HybridMLP)
  TRS_REQUIRE(p.slots_per_xcd > 0 && S >= 8 && S <= 8 * p.slots_per_xcd && (8 * p.slots_per_xcd) % S == 0 && is_pow2(S / 8),
              TRS_ESHAPE, "wgrad_rows: (M=%d, N=%d, rows=%lld) takes %d row ranges (or that halved down to 8), caller passed %d",
              M, N, (long long)rows, 8 * p.slots_per_xcd, S);
  p.slots_per_xcd = S / 8;
  TRS_REQUIRE(ldg >= M && ldx >= N && (ldg & 7) == 0 && (ldx & 7) == 0 && aligned16(g) && aligned16(x), TRS_ESHAPE,
              "wgrad_rows: row strides must be multiples of 8 elements and cover M / N, operands 16-byte aligned");
  if (!g || !x || !part) return fail(TRS_EINVAL, "wgrad_rows: null pointer");
  hipStream_t s = (hipStream_t)stream;
  WgradArgs a{(const uint16_t*)g, (const uint16_t*)x, part, rows, ldg, ldx, M, N, p.MB, p.NB, p.slots_per_xcd};
  const int Mt = (M + 15) / 16, Nt = (N + 15) / 16;
  const int PM = (Mt + p.MB - 1) / p.MB, PN = (Nt + p.NB - 1) / p.NB;
  const size_t lds = (size_t)(4 * (PM + PN) + WG_TC) * WG_PANEL;      // + the panels a short wave runs on into
  const int grid = 8 * p.slots_per_xcd * p.MB * p.NB;
  TRS_REQUIRE(p.dma == 0 || (int64_t)WG_KS * std::max(ldg, ldx) * 2 < ((int64_t)1 << 31), TRS_ESHAPE, "wgrad_rows: row stride too large");
  if (p.dma == 2) {
    static bool attr2 = false;
    if (!attr2) {
      if (hipFuncSetAttribute((const void*)wgrad_dma2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return check_launch("wgrad_rows(dma2): LDS attribute");
      attr2 = true;
    }
    hipLaunchKernelGGL(wgrad_dma2_kernel, dim3(8 * p.slots_per_xcd * p.MB), dim3(256), (size_t)WD2_NSLOT * WD2_SLOT, s, a);
    return check_launch("wgrad_rows(dma2)");
  }
  if (p.dma == 1) {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute((const void*)wgrad_dma_kernel<7, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return check_launch("wgrad_rows(dma): LDS attribute");
      attr = true;
    }
    hipLaunchKernelGGL((wgrad_dma_kernel<7, 4>), dim3(grid), dim3(512), (size_t)WD_NSLOT * WD_SLOT, s, a);
    return check_launch("wgrad_rows(dma)");
  }
  if (p.wm == 2 && p.wn == 4)
    return p.nc == 4 ? wgrad_launch<2, 4, 7, 4>(a, grid, lds, s) : wgrad_launch<2, 4, 7, 3>(a, grid, lds, s);
  if (p.wm == 2)
    return p.nc == 5 ? wgrad_launch<2, 2, WG_TC, 5>(a, grid, lds, s) : wgrad_launch<2, 2, WG_TC, WG_TC>(a, grid, lds, s);
  if (p.wm == 4) return p.nc == 4 ? wgrad_launch<4, 1, WG_TC, 4>(a, grid, lds, s) : wgrad_launch<4, 1, WG_TC, WG_TC>(a, grid, lds, s);
  if (p.mc == 1) return wgrad_launch<1, 4, 1, WG_TC>(a, grid, lds, s);
  if (p.mc == 4) return wgrad_launch<1, 4, 4, WG_TC>(a, grid, lds, s);
  return wgrad_launch<1, 4, WG_TC, WG_TC>(a, grid, lds, s);
}
