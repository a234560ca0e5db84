// K5 on the matrix cores, packed-fp16 operand form (round 4).
//
//   y[p,c] = sum_{n,h} W[c,n,h] * x0[p,n] * xk[p,h]            p = pixel (b,e)
//
// cin_mfma.hip multiplies the x0[n] factor into the MFMA RESULT (fp32 FMAs: 2.4-4 VALU instructions per 16-cycle MFMA,
// which is what bounds those kernels -- the matrix pipe and the VALU of a SIMD share the issue port).  gfx950 has no
// packed bf16 arithmetic, but it has v_pk_mul_f16, and v_mfma_f32_32x32x16_f16 runs at the bf16 rate: with the CIN
// layer's INTERNAL tensors held in fp16 (hidden state, weights as fragments, the conv-output gradient times a power of
// two) the outer product Z[p,(n,h)] = x0[p,n]*xk[p,h] is formed as an MFMA OPERAND -- 4 v_pk_mul_f16 per fragment, and
// a fragment feeds every channel tile of the wave -- so the contraction is one long GEMM over K = N*H whose
// accumulators never leave the matrix pipe.  fp16 carries 3 more mantissa bits than bf16; what it lacks is range, which
// is handled outside (power-of-two scales handed in as device scalars, torecsys_amd/functional.py).
//
//   forward      B operand = xk fragment (resident in registers) * x0[p,n]   0.5 VALU per 32-cycle MFMA
//   data grad    S_n = W_n^T gy on the pipe, dxk += x0*S, dx0 = sum_h xk*S    2 v_fma_mix per MFMA (fp16 sources, no unpack)
//   weight grad  B operand = xk^T fragment * x0^T vector (pixels on K)       2 VALU per MFMA
//
// All three: wave = one sample's 64 pixels (E = 64) as two 32-pixel tiles of v_mfma_f32_32x32x16_f16, one wave per SIMD
// (512 registers), weights streamed L2 -> LDS by LDS-DMA one step ahead (lane-linear fragment images, conflict-free
// ds_read_b128).
#include "../../../torecsys_amd/csrc/trs_common.hpp"   // error helpers are resolved from libtrs_hip.so at link time

namespace trs {

typedef _Float16 h16;
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int I>
struct IC { static constexpr int value = I; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IC<I>{});
    static_for<I + 1, N>(f);
  }
}

// D row rho of a 32x32 tile sits in register (rho&3) + 4*(rho>>3) of lane group (rho>>2)&1; feeding the A rows in the
// order below gives every lane 16 CONSECUTIVE output channels (32-byte runs of its pixel's row).
__host__ __device__ __forceinline__ int cin16_row_slot(int rho) { return 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3); }

__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// Staging of one step's weight image from L2 into LDS.  A wave moves PP pieces of 1 KiB (one 16-byte vector per lane)
// in groups of <= 4, group Q at sub-step Q of the step.  Two implementations with one interface:
//  * Cin16StageDma (shipped): LDS-DMA (global_load_lds_dwordx4), no registers, no LDS-write instructions; the pieces
//    are waited for with one vmcnt(0) in tail() -- inside a step no other vector-memory operation of the wave is in flight.
//    M0 belongs to the compiler: saved and restored inside the statement.  No "memory" clobber on purpose: the
//    destination is the buffer nobody reads before the next barrier, ordering comes from tail() + the barrier.
//  * Cin16StageReg (-DTRS_CIN16_STAGE_REG, measured and not kept): hand-issued global_load_dwordx4 into a two-set register
//    ring, written to LDS two sub-steps later.  The idea was that an LDS-DMA's issue costs the wave ~100 cycles per piece
//    beside MFMAs (MI355X_MICROARCH.md price list); measured the other way round: forward 8.32 -> 8.87 ms, data gradient
//    9.4 -> 13.7 ms at B = 65 536 (the counted waits in front of the LDS writes park the wave).
template <int NP>
__device__ __forceinline__ void cin16_dma(const char* src_lane, unsigned lds_dst_uniform) {
  static_assert(NP >= 1 && NP <= 4, "pieces per statement");
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst_uniform);
  if constexpr (NP == 4)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(src_lane), "s"(dst));
  else if constexpr (NP == 3)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(src_lane), "s"(dst));
  else if constexpr (NP == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(src_lane), "s"(dst));
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(src_lane), "s"(dst));
}

template <int PP>
struct Cin16StageDma {
  static constexpr int NG = (PP + 3) / 4;
  __host__ __device__ static constexpr int pieces(int q) { return q < PP / 4 ? 4 : (q == PP / 4 ? PP % 4 : 0); }
  // src_wave_lane: the wave's share of the image + lane * 16; lds_wave: the wave's share of the target buffer
  template <int SS>
  __device__ __forceinline__ void substep(const char* src_wave_lane, u32x4* lds_wave, int) {
    if constexpr (SS < NG) cin16_dma<pieces(SS)>(src_wave_lane + SS * 4096, lds_addr_of(lds_wave) + SS * 4096);
  }
  template <int NSS>
  __device__ __forceinline__ void tail(u32x4*, int) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __device__ __forceinline__ void all(const char* src_wave_lane, u32x4* lds_wave, int lane) {
    static_for<0, NG>([&](auto q) { substep<decltype(q)::value>(src_wave_lane, lds_wave, lane); });
    tail<NG>(lds_wave, lane);
  }
};

template <int PP>
struct Cin16StageReg {
  static constexpr int NG = (PP + 3) / 4;
  __host__ __device__ static constexpr int pieces(int q) { return q < PP / 4 ? 4 : (q == PP / 4 ? PP % 4 : 0); }
  u32x4 R[2][4];
  template <int Q>
  __device__ __forceinline__ void load(const char* src_wave_lane) {
    if constexpr (Q < NG) {
      const char* p = src_wave_lane + Q * 4096;
      constexpr int NPC = pieces(Q);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(R[Q & 1][0]) : "v"(p));
      if constexpr (NPC > 1) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(R[Q & 1][1]) : "v"(p));
      if constexpr (NPC > 2) asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(R[Q & 1][2]) : "v"(p));
      if constexpr (NPC > 3) asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(R[Q & 1][3]) : "v"(p));
    }
  }
  // LEFT = pieces of the groups requested after Q that may still be on their way
  template <int Q, int LEFT>
  __device__ __forceinline__ void store(u32x4* lds_wave, int lane) {
    if constexpr (Q >= 0 && Q < NG) {
      constexpr int NPC = pieces(Q);
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LEFT) : "memory");
      asm volatile("" : "+v"(R[Q & 1][0]), "+v"(R[Q & 1][1]), "+v"(R[Q & 1][2]), "+v"(R[Q & 1][3]));
#pragma unroll
      for (int j = 0; j < NPC; ++j) lds_wave[(Q * 4 + j) * 64 + lane] = R[Q & 1][j];
    }
  }
  template <int SS>
  __device__ __forceinline__ void substep(const char* src_wave_lane, u32x4* lds_wave, int lane) {
    store<SS - 2, pieces(SS - 1)>(lds_wave, lane);
    load<SS>(src_wave_lane);
  }
  template <int Q, int NSS>
  __device__ __forceinline__ void tail_from(u32x4* lds_wave, int lane) {
    if constexpr (Q < NG) {
      constexpr int left = [] { int t = 0; for (int q = Q + 1; q < NG; ++q) t += pieces(q); return t; }();
      store<Q, left>(lds_wave, lane);
      tail_from<Q + 1, NSS>(lds_wave, lane);
    }
  }
  template <int NSS>
  __device__ __forceinline__ void tail(u32x4* lds_wave, int lane) { tail_from<(NSS >= 2 ? NSS - 2 : 0), NSS>(lds_wave, lane); }
  template <int Q>
  __device__ __forceinline__ void all_from(const char* src_wave_lane, u32x4* lds_wave, int lane) {
    if constexpr (Q < NG) {
      load<Q>(src_wave_lane);
      store<Q, 0>(lds_wave, lane);
      all_from<Q + 1>(src_wave_lane, lds_wave, lane);
    }
  }
  __device__ __forceinline__ void all(const char* src_wave_lane, u32x4* lds_wave, int lane) { all_from<0>(src_wave_lane, lds_wave, lane); }
};

#ifdef TRS_CIN16_STAGE_REG
template <int PP> using Cin16Stage = Cin16StageReg<PP>;
#else
template <int PP> using Cin16Stage = Cin16StageDma<PP>;
#endif

__device__ __forceinline__ h16x8 cin16_scale(const uint4& f, h16x2 s) {
  const h16x2 a = __builtin_bit_cast(h16x2, f.x) * s, b = __builtin_bit_cast(h16x2, f.y) * s,
              c = __builtin_bit_cast(h16x2, f.z) * s, d = __builtin_bit_cast(h16x2, f.w) * s;
  return h16x8{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// Wp[((n*KS + ks)*CT + ct)*64 + lane][8] = W[32 ct + slot(lane&31)][n][16 ks + 8 (lane>>5) + 0..7]   (fp16, 0 past H)
__global__ __launch_bounds__(256) void cin16_prepack_fwd_kernel(const float* __restrict__ W, h16* __restrict__ Wp, int C,
                                                                int N, int H, int KS) {
  const int CT = C / 32;
  const int64_t total = (int64_t)N * KS * CT * 64;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(t & 63);
    int64_t f = t >> 6;
    const int ct = (int)(f % CT); f /= CT;
    const int ks = (int)(f % KS);
    const int n = (int)(f / KS);
    const int c = 32 * ct + cin16_row_slot(lane & 31);
    const int h0 = 16 * ks + 8 * (lane >> 5);
    h16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (h0 + j < H) ? (h16)W[(size_t)c * N * H + (size_t)n * H + h0 + j] : (h16)0.f;
    *reinterpret_cast<h16x8*>(Wp + t * 8) = v;
  }
}

// x0h (B,N,E) fp16; xkT (B*E rows, ldk) fp16 channels-last, zeros past H; yT (B,E,C) bf16.
// A step = one field n: its C x (16 KS) weight image (KS*CT KiB) is in LDS buffer `par`, the next field's is on its way
// into the other one.  The image stream is cyclic in n and does not depend on the item, so it runs continuously over
// the items a workgroup walks.
// developer ablations of the forward kernel (wrong results, timing only): -DTRS_CIN16_ABL=<bits>
//   1 no staging of the next image   2 no barrier   4 no LDS fragment reads   8 no operand scaling
#ifndef TRS_CIN16_ABL
#define TRS_CIN16_ABL 0
#endif
template <int KS, int CT, bool TRI>
__global__ __launch_bounds__(256) void cin16_fwd_kernel(const h16* __restrict__ x0h, const h16* __restrict__ xkT, int ldk,
                                                        const char* __restrict__ Wp, const float* __restrict__ bias,
                                                        const float* __restrict__ out_mul, bf16_t* __restrict__ yT,
                                                        int64_t nitems, int N, int E) {
  constexpr int C = 32 * CT;
  constexpr int FRB = KS * CT * 1024;       // bytes of one step's image
  constexpr int PP = KS * CT / 4;           // 1 KiB pieces per wave
  static_assert((KS * CT) % 4 == 0 && Cin16Stage<PP>::NG <= KS, "staging split");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, g = lane >> 5;
  const uint4* Wb = reinterpret_cast<const uint4*>(smem);
  unsigned* x0w = reinterpret_cast<unsigned*>(smem + 2 * FRB) + (size_t)wave * N * 32;      // [N][32] (tile 0 | tile 1 << 16)
  float* bs = reinterpret_cast<float*>(smem + 2 * FRB + (size_t)4 * N * 32 * 4);            // [C]
  u32x4* wdst = reinterpret_cast<u32x4*>(smem + wave * (PP * 1024));            // this wave's share of buffer 0
  const char* wsrc = Wp + wave * (PP * 1024) + lane * 16;
  const int epb = E / 64;
  for (int i = threadIdx.x; i < C; i += 256) bs[i] = bias ? bias[i] : 0.f;
  Cin16Stage<PP> stg;
  stg.all(wsrc, wdst, lane);
  __syncthreads();
  const float mul = out_mul ? *out_mul : 1.f;
  int par = 0;
  // fragment sets: k-step ks reads set ks & 1; an odd KS gives its last k-step a third set, because the first fragments
  // of the NEXT step (set 0) are fetched while that k-step's MFMAs are still to be issued
  constexpr int NSET = (!TRI && (KS & 1)) ? 3 : 2;
  auto aset = [](int ks) constexpr { return ((KS & 1) && ks == KS - 1 && KS > 1) ? 2 : (ks & 1); };
  uint4 Af[NSET][CT];
  if constexpr (!TRI) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) Af[aset(0)][ct] = Wb[ct * 64 + lane];
  }
  for (int64_t it0 = (int64_t)blockIdx.x * 4; it0 < nitems; it0 += (int64_t)gridDim.x * 4) {
    const bool live = it0 + wave < nitems;
    const int64_t it = live ? it0 + wave : 0;
    const int64_t b = it / epb;
    const int e0 = (int)(it - b * epb) * 64;
    const int64_t pix0 = b * E + e0;
    uint4 Bf[2][KS];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        Bf[t][ks] = *reinterpret_cast<const uint4*>(xkT + (pix0 + 32 * t + r) * ldk + 16 * ks + 8 * g);
    for (int v = lane; v < N * 32; v += 64) {
      const unsigned short* p = reinterpret_cast<const unsigned short*>(x0h) + (b * N + (v >> 5)) * E + e0 + (v & 31);
      x0w[v] = (unsigned)p[0] | ((unsigned)p[32] << 16);
    }
    // the xk fragments are waited for here, once: inside the loop the only vector-memory operations in flight are the
    // stager's hand-issued loads, which the compiler's vmcnt bookkeeping does not know about
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(Bf[t][ks].x), "v"(Bf[t][ks].y), "v"(Bf[t][ks].z), "v"(Bf[t][ks].w));
    f32x16 acc[2][CT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][ct][i] = 0.f;
    auto loadA = [&](int set, const uint4* img, int ks) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) Af[set][ct] = img[(ks * CT + ct) * 64 + lane];
    };
    auto mfmas = [&](int set, const h16x8& s0, const h16x8& s1) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const h16x8 a = __builtin_bit_cast(h16x8, Af[set][ct]);
        acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, s0, acc[0][ct], 0, 0, 0);
        acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, s1, acc[1][ct], 0, 0, 0);
      }
    };
    int n = 0;
    do {                                   // N >= 1: no zero-trip path (hipcc otherwise shuffles all 256 accumulators)
      const int nn = n + 1 < N ? n + 1 : 0;
      const h16x2 xv = __builtin_bit_cast(h16x2, x0w[n * 32 + r]);
      const h16x2 xlo = {xv.x, xv.x}, xhi = {xv.y, xv.y};
      const uint4* A = Wb + par * (FRB / 16);
      const uint4* Anext = Wb + (par ^ 1) * (FRB / 16);
      if constexpr (TRI) {
        // lower-triangular weights: field n has nothing on h > n, its k-steps past n / 16 hold zeros only
        const int ks_end = n / 16 + 1;
        loadA(0, A, 0);
        static_for<0, KS>([&](auto ksc) {
          constexpr int ks = decltype(ksc)::value;
          stg.template substep<ks>(wsrc + (size_t)nn * FRB, wdst + (par ^ 1) * (FRB / 16), lane);
          if (ks < ks_end) {
            if (ks + 1 < KS && ks + 1 < ks_end) loadA((ks + 1) & 1, A, ks + 1);
            const h16x8 s0 = cin16_scale(Bf[0][ks], xlo), s1 = cin16_scale(Bf[1][ks], xhi);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(ks & 1, s0, s1);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        stg.template tail<KS>(wdst + (par ^ 1) * (FRB / 16), lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      } else {
        static_for<0, KS>([&](auto ksc) {
          constexpr int ks = decltype(ksc)::value;
          if constexpr (!(TRS_CIN16_ABL & 1)) stg.template substep<ks>(wsrc + (size_t)nn * FRB, wdst + (par ^ 1) * (FRB / 16), lane);
          if constexpr (ks + 1 < KS) {
            if constexpr (!(TRS_CIN16_ABL & 4)) loadA(aset(ks + 1), A, ks + 1);
          } else {
            // the step's last k-step: its fragments are in registers, so the hand-over to the next image (image written,
            // own reads done, barrier, first fragments of the next step) happens BEFORE its MFMAs -- the matrix pipe has
            // 16 of them to chew on while the first LDS reads of the next step are in flight
            if constexpr (!(TRS_CIN16_ABL & 1)) stg.template tail<KS>(wdst + (par ^ 1) * (FRB / 16), lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (!(TRS_CIN16_ABL & 2)) __builtin_amdgcn_s_barrier();
            if constexpr (!(TRS_CIN16_ABL & 4)) loadA(aset(0), Anext, 0);
          }
          const h16x8 s0 = (TRS_CIN16_ABL & 8) ? __builtin_bit_cast(h16x8, Bf[0][ks]) : cin16_scale(Bf[0][ks], xlo);
          const h16x8 s1 = (TRS_CIN16_ABL & 8) ? __builtin_bit_cast(h16x8, Bf[1][ks]) : cin16_scale(Bf[1][ks], xhi);
          __builtin_amdgcn_sched_barrier(0);
          mfmas(aset(ks), s0, s1);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      par ^= 1;
    } while (++n < N);
    if (live) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          float f[16];
          asm volatile("" : "+a"(acc[t][ct]));   // stays in the accumulator file until its own turn
#pragma unroll
          for (int i = 0; i < 16; ++i) f[i] = fmaf(acc[t][ct][i], mul, bs[32 * ct + 16 * g + i]);
          uint4* dst = reinterpret_cast<uint4*>(yT + (pix0 + 32 * t + r) * (int64_t)C + 32 * ct + 16 * g);
          dst[0] = Vec16<bf16_t>::pack(f);
          dst[1] = Vec16<bf16_t>::pack(f + 8);
          __builtin_amdgcn_sched_barrier(0);     // one tile at a time: reading all 256 accumulators first spills
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// data gradients:  S_n[h,p] = sum_c W[c,n,h] gy[p,c]  (MFMA, K = channels)
//     dxk[p,h] = sum_n x0[p,n] S_n[h,p]        dx0[p,n] = sum_h xk[p,h] S_n[h,p]
// Both epilogues need S itself, so the x0 factor cannot ride on an operand here: 2 VALU instructions per S element
// (v_fma_mix_f32 takes the fp16 x0 / xk values as they are: no unpack, no conversion).  A wave owns ONE 32-pixel tile
// and runs (field, 32-h tile) groups of KC MFMAs followed by their 32 FMAs; two waves share a SIMD (8 per workgroup), so
// one wave's FMAs sit beside the other's MFMAs without any hand-made pipeline.  gy fragments (the B operand, KC x 4
// registers) and the lane's own xk values stay in registers for the whole item.
// WpT[((n*HT + ht)*KC + kc)*64 + lane][8] = W[16 kc + 8 (lane>>5) + 0..7][n][32 ht + slot(lane&31)]   (fp16, 0 past H)
__global__ __launch_bounds__(256) void cin16_prepack_bwd_kernel(const float* __restrict__ W, h16* __restrict__ WpT, int C,
                                                                int N, int H, int HT) {
  const int KC = C / 16;
  const int64_t total = (int64_t)N * HT * KC * 64;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(t & 63);
    int64_t f = t >> 6;
    const int kc = (int)(f % KC); f /= KC;
    const int ht = (int)(f % HT);
    const int n = (int)(f / HT);
    const int h = 32 * ht + cin16_row_slot(lane & 31);
    const int c0 = 16 * kc + 8 * (lane >> 5);
    h16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = h < H ? (h16)W[(size_t)(c0 + j) * N * H + (size_t)n * H + h] : (h16)0.f;
    *reinterpret_cast<h16x8*>(WpT + t * 8) = v;
  }
}

// gyT rows (B*E) of C = 16 KC fp16; xkT rows of stride ldk >= 32 HT (zeros past H); dx0 (B,N,E) bf16;
// dxkT rows of stride ldo >= 32 HT bf16.  Items are 32-pixel tiles.
template <int KC, int HT, bool TRI>
__global__ __launch_bounds__(512) void cin16_bwd_data_kernel(const h16* __restrict__ x0h, const h16* __restrict__ xkT,
                                                             int ldk, const h16* __restrict__ gyT,
                                                             const char* __restrict__ WpT, const float* __restrict__ out_mul,
                                                             bf16_t* __restrict__ dx0, bf16_t* __restrict__ dxkT, int ldo,
                                                             int64_t nitems, int N, int E) {
  constexpr int C = 16 * KC;
  constexpr int FRB = HT * KC * 1024;
  constexpr int PP = HT * KC / 8;
  static_assert((HT * KC) % 8 == 0 && Cin16Stage<PP>::NG <= HT, "staging split");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, g = lane >> 5;
  const uint4* Wb = reinterpret_cast<const uint4*>(smem);
  unsigned short* x0w = reinterpret_cast<unsigned short*>(smem + 2 * FRB) + (size_t)wave * N * 32;     // [N][32]
  u32x4* wdst = reinterpret_cast<u32x4*>(smem + wave * (PP * 1024));
  const char* wsrc = WpT + wave * (PP * 1024) + lane * 16;
  const int epb = E / 32;
  Cin16Stage<PP> stg;
  stg.all(wsrc, wdst, lane);
  __syncthreads();
  const float mul = out_mul ? *out_mul : 1.f;
  int par = 0;
  for (int64_t it0 = (int64_t)blockIdx.x * 8; it0 < nitems; it0 += (int64_t)gridDim.x * 8) {
    const bool live = it0 + wave < nitems;
    const int64_t it = live ? it0 + wave : 0;
    const int64_t b = it / epb;
    const int e0 = (int)(it - b * epb) * 32;
    const int64_t pix = b * E + e0 + r;
    uint4 Gf[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) Gf[kc] = *reinterpret_cast<const uint4*>(gyT + pix * C + 16 * kc + 8 * g);
    uint4 Xf[HT][2];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int q = 0; q < 2; ++q) Xf[ht][q] = *reinterpret_cast<const uint4*>(xkT + pix * ldk + 32 * ht + 16 * g + 8 * q);
    for (int v = lane; v < N * 32; v += 64)
      x0w[v] = reinterpret_cast<const unsigned short*>(x0h)[(b * N + (v >> 5)) * E + e0 + (v & 31)];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) asm volatile("" ::"v"(Gf[kc].x), "v"(Gf[kc].y), "v"(Gf[kc].z), "v"(Gf[kc].w));
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int q = 0; q < 2; ++q) asm volatile("" ::"v"(Xf[ht][q].x), "v"(Xf[ht][q].y), "v"(Xf[ht][q].z), "v"(Xf[ht][q].w));
    f32x16 dxk[HT];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int i = 0; i < 16; ++i) dxk[ht][i] = 0.f;
    int n = 0;
    do {
      const int nn = n + 1 < N ? n + 1 : 0;
      const float x0f = (float)__builtin_bit_cast(h16, x0w[n * 32 + r]);
      const uint4* A = Wb + par * (FRB / 16);
      const int ht_end = TRI ? n / 32 + 1 : HT;      // tri: field n has no weight on h > n
      float d0 = 0.f;
      static_for<0, HT>([&](auto htc) {
        constexpr int ht = decltype(htc)::value;
        stg.template substep<ht>(wsrc + (size_t)nn * FRB, wdst + (par ^ 1) * (FRB / 16), lane);
        if (!TRI || ht < ht_end) {
          f32x16 S;
#pragma unroll
          for (int i = 0; i < 16; ++i) S[i] = 0.f;
#pragma unroll
          for (int kc = 0; kc < KC; ++kc)
            S = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A[(ht * KC + kc) * 64 + lane]),
                                                       __builtin_bit_cast(h16x8, Gf[kc]), S, 0, 0, 0);
          // a group's FMAs follow its MFMAs in program order (the partner wave of the SIMD fills the matrix pipe meanwhile);
          // left to itself hipcc runs the MFMAs of all HT groups first and keeps HT result tiles alive
          __builtin_amdgcn_sched_barrier(0);
          // opaque copies: the fp16 -> fp32 conversions must not be hoisted out of the loop as 16 HT live registers
          uint4 xq0 = Xf[ht][0], xq1 = Xf[ht][1];
          asm volatile("" : "+v"(xq0.x), "+v"(xq0.y), "+v"(xq0.z), "+v"(xq0.w), "+v"(xq1.x), "+v"(xq1.y), "+v"(xq1.z), "+v"(xq1.w));
          const h16x8 xa = __builtin_bit_cast(h16x8, xq0), xb = __builtin_bit_cast(h16x8, xq1);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            dxk[ht][i] = __builtin_fmaf(x0f, S[i], dxk[ht][i]);
            d0 = __builtin_fmaf((float)(i < 8 ? xa[i] : xb[i - 8]), S[i], d0);
          }
          asm volatile("" : "+v"(dxk[ht]));     // the update stays here (hipcc sinks it past the barrier and keeps HT S tiles)
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      stg.template tail<HT>(wdst + (par ^ 1) * (FRB / 16), lane);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0), __float_as_uint(d0), false, false);
        const float tot = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * mul;
        if (live && g == 0) dx0[(b * N + n) * E + e0 + r] = from_f32<bf16_t>(tot);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      par ^= 1;
    } while (++n < N);
    if (live) {
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) {
        float f[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = dxk[ht][i] * mul;
        uint4* dst = reinterpret_cast<uint4*>(dxkT + pix * ldo + 32 * ht + 16 * g);
        dst[0] = Vec16<bf16_t>::pack(f);
        dst[1] = Vec16<bf16_t>::pack(f + 8);
      }
    }
  }
}

static size_t cin16_bwd_lds(int N, int KC, int HT) { return (size_t)2 * HT * KC * 1024 + (size_t)8 * N * 32 * 2; }

static bool cin16_shape_ok(int N, int H, int C, int E) {
  const int KS = (H + 15) / 16;
  return E % 64 == 0 && (C == 128 || C == 256) && KS >= 1 && KS <= 8 && N >= 1 && N <= 64;
}
static size_t cin16_fwd_lds(int N, int KS, int CT) { return (size_t)2 * KS * CT * 1024 + (size_t)4 * N * 32 * 4 + 32 * CT * 4; }

}  // namespace trs

using namespace trs;

extern "C" int trs_cin16_supported(int32_t N, int32_t H, int32_t C, int32_t E) {
  if (!cin16_shape_ok(N, H, C, E)) return 0;
  return cin16_fwd_lds(N, (H + 15) / 16, C / 32) <= 160 * 1024 ? 1 : 0;
}

extern "C" size_t trs_cin16_fwd_workspace_bytes(int32_t N, int32_t H, int32_t C) {
  if (N <= 0 || H <= 0 || C <= 0) return 0;
  return (size_t)N * ((H + 15) / 16) * (C / 32) * 1024 + 256;
}

extern "C" int trs_cin16_fwd(const void* x0h, const void* xkT, int32_t ldk, const float* W, const float* bias,
                             const float* out_mul, int64_t B, int32_t N, int32_t H, int32_t C, int32_t E, int32_t tri,
                             void* yT, void* workspace, size_t ws_bytes, trs_stream_t stream) {
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(x0h && xkT && W && yT && workspace, TRS_EINVAL, "cin16_fwd: NULL pointer");
  TRS_REQUIRE(B > 0 && trs_cin16_supported(N, H, C, E), TRS_ESHAPE,
              "cin16_fwd: shape not covered (E%%64==0, C in {128,256}, H<=128, N<=64; got N %d H %d C %d E %d)", N, H, C, E);
  const int KS = (H + 15) / 16, CT = C / 32;
  TRS_REQUIRE(ldk % 8 == 0 && ldk >= 16 * KS && aligned16(xkT) && aligned16(yT) && aligned16(workspace), TRS_EINVAL,
              "cin16_fwd: xkT rows must be 16-byte aligned and hold 16*ceil(H/16) values");
  TRS_REQUIRE(!tri || N == H, TRS_EINVAL, "cin16_fwd: tri needs H == N");
  TRS_REQUIRE(ws_bytes >= trs_cin16_fwd_workspace_bytes(N, H, C), TRS_EWORKSPACE, "cin16_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = (int64_t)N * KS * CT * 64;
  hipLaunchKernelGGL(cin16_prepack_fwd_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, s, W,
                     (h16*)workspace, C, N, H, KS);
  const size_t lds = cin16_fwd_lds(N, KS, CT);
  const int64_t nitems = B * (E / 64);
  const int grid = (int)std::min<int64_t>((nitems + 3) / 4, 256);
#define TRS_C16F(KS_, CT_, TRI_)                                                                                       \
  do {                                                                                                                 \
    auto kern = cin16_fwd_kernel<KS_, CT_, TRI_>;                                                                      \
    static size_t attr_lds = 0;                                                                                        \
    if (lds > 64 * 1024 && lds > attr_lds) {                                                                           \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)  \
        return check_launch("cin16_fwd: LDS attribute");                                                               \
      attr_lds = lds;                                                                                                  \
    }                                                                                                                  \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, (const h16*)x0h, (const h16*)xkT, ldk,                     \
                       (const char*)workspace, bias, out_mul, (bf16_t*)yT, nitems, N, E);                              \
  } while (0)
#define TRS_C16F_CT(KS_, TRI_)                   \
  do {                                           \
    if (CT == 8) TRS_C16F(KS_, 8, TRI_);         \
    else TRS_C16F(KS_, 4, TRI_);                 \
  } while (0)
  const bool t3 = tri && KS > 1;
  switch (KS) {
    case 1: TRS_C16F_CT(1, false); break;
    case 2: if (t3) TRS_C16F_CT(2, true); else TRS_C16F_CT(2, false); break;
    case 3: if (t3) TRS_C16F_CT(3, true); else TRS_C16F_CT(3, false); break;
    case 4: if (t3) TRS_C16F_CT(4, true); else TRS_C16F_CT(4, false); break;
    case 5: TRS_C16F_CT(5, false); break;
    case 6: TRS_C16F_CT(6, false); break;
    case 7: TRS_C16F_CT(7, false); break;
    default: TRS_C16F_CT(8, false); break;
  }
#undef TRS_C16F_CT
#undef TRS_C16F
  return check_launch("cin16_fwd");
}

extern "C" size_t trs_cin16_bwd_data_workspace_bytes(int32_t N, int32_t H, int32_t C) {
  if (N <= 0 || H <= 0 || C <= 0) return 0;
  return (size_t)N * ((H + 31) / 32) * (C / 16) * 1024 + 256;
}

extern "C" int trs_cin16_bwd_data(const void* x0h, const void* xkT, int32_t ldk, const void* gyT, const float* W,
                                  const float* out_mul, int64_t B, int32_t N, int32_t H, int32_t C, int32_t E,
                                  int32_t tri, void* dx0, void* dxkT, int32_t ldo, void* workspace, size_t ws_bytes,
                                  trs_stream_t stream) {
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(x0h && xkT && gyT && W && dx0 && dxkT && workspace, TRS_EINVAL, "cin16_bwd_data: NULL pointer");
  TRS_REQUIRE(B > 0 && trs_cin16_supported(N, H, C, E), TRS_ESHAPE,
              "cin16_bwd_data: shape not covered (E%%64==0, C in {128,256}, H<=128, N<=64; got N %d H %d C %d E %d)", N, H, C,
              E);
  const int HT = (H + 31) / 32, KC = C / 16;
  TRS_REQUIRE(ldk % 8 == 0 && ldk >= 32 * HT && ldo % 8 == 0 && ldo >= 32 * HT && aligned16(xkT) && aligned16(gyT) &&
                  aligned16(dxkT) && aligned16(workspace),
              TRS_EINVAL, "cin16_bwd_data: rows must be 16-byte aligned and hold 32*ceil(H/32) values");
  TRS_REQUIRE(!tri || N == H, TRS_EINVAL, "cin16_bwd_data: tri needs H == N");
  TRS_REQUIRE(ws_bytes >= trs_cin16_bwd_data_workspace_bytes(N, H, C), TRS_EWORKSPACE,
              "cin16_bwd_data: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = (int64_t)N * HT * KC * 64;
  hipLaunchKernelGGL(cin16_prepack_bwd_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, s, W,
                     (h16*)workspace, C, N, H, HT);
  const size_t lds = cin16_bwd_lds(N, KC, HT);
  TRS_REQUIRE(lds <= 160 * 1024, TRS_ESHAPE, "cin16_bwd_data: LDS budget exceeded (N %d)", N);
  const int64_t nitems = B * (E / 32);
  const int grid = (int)std::min<int64_t>((nitems + 7) / 8, 256);
#define TRS_C16B(KC_, HT_, TRI_)                                                                                       \
  do {                                                                                                                 \
    auto kern = cin16_bwd_data_kernel<KC_, HT_, TRI_>;                                                                 \
    static size_t attr_lds = 0;                                                                                        \
    if (lds > 64 * 1024 && lds > attr_lds) {                                                                           \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)  \
        return check_launch("cin16_bwd_data: LDS attribute");                                                          \
      attr_lds = lds;                                                                                                  \
    }                                                                                                                  \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, (const h16*)x0h, (const h16*)xkT, ldk, (const h16*)gyT,    \
                       (const char*)workspace, out_mul, (bf16_t*)dx0, (bf16_t*)dxkT, ldo, nitems, N, E);               \
  } while (0)
#define TRS_C16B_HT(KC_)                                   \
  do {                                                     \
    if (HT == 1) TRS_C16B(KC_, 1, false);                  \
    else if (HT == 2 && tri) TRS_C16B(KC_, 2, true);       \
    else if (HT == 2) TRS_C16B(KC_, 2, false);             \
    else if (HT == 3) TRS_C16B(KC_, 3, false);             \
    else TRS_C16B(KC_, 4, false);                          \
  } while (0)
  if (KC == 16) TRS_C16B_HT(16);
  else TRS_C16B_HT(8);
#undef TRS_C16B_HT
#undef TRS_C16B
  return check_launch("cin16_bwd_data");
}
