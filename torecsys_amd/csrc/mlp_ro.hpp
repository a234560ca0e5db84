// SURVEY.md 8f N4, second form of the fused Linear+ReLU stack (layers/ctr/multilayer_perceptron.py:63-84 applied to the
// (B*N, E) rows of DeepAndCrossNetwork, models/ctr/deep_and_cross_network.py:71-87, and the tail of the deep branches of
// DeepFM / xDeepFM): "row owner" kernels.  mlp_fused.hip keeps a 128-row tile of activations in LDS and splits the OUTPUT
// columns over 8 waves: every layer ends with two workgroup barriers and an LDS round trip of the whole tile, and the
// weight fragments come from L2 once per wave and 128 rows.  Here a wave OWNS its rows for the whole stack:
//
//   * D (32 output columns x 32 rows) = A (weights, 32 x 16 per k-step) x B (the rows' activations, 16 x 32 per k-step)
//     with v_mfma_f32_32x32x16_bf16; the weight rows are fed in a permuted order (ro_col_of_row) so that a lane's 16
//     results are columns 32 ct + 16 h + 8 g + (0..7), h = 0, 1 of ITS row: after bias / ReLU / rounding they are, as
//     they stand, the B operands 2 ct + h of the next layer.  A layer's activations never leave the wave.
//   * a 400-wide input is 100 registers per 32 rows; the next layer's input, as it is produced chunk by chunk (32
//     columns), waits in registers (the last NREG chunks) and in LDS (the chunks before: 64 bytes per row and chunk,
//     written and read back by the same lane, no barrier), and takes over the input's registers at the end of the layer.
//     (Reading it back from the global copy the weight-gradient GEMMs need anyway cost 1.0 of 3.6 ms: 6.8 MB per XCD
//     in flight between store and reload do not stay in a 4 MB L2.)
//   * the weights pass through LDS: a chunk (32 output columns x K, <= 26 KB in MFMA fragment order) is copied by LDS-DMA
//     two chunks ahead into a ring of three slots, one workgroup barrier per chunk (in the middle of the chunk, so that
//     the next chunk's first fragments can be read ahead of its first MFMA), one L2 read per workgroup and 256 rows.
//   * the epilogue of chunk c (bias is the accumulators' initial value; ReLU and the sign bits on the packed words;
//     16-byte stores) is spread word by word over the first half of chunk c+1, its stores follow the barrier.
//   * backward: the same walk over the transposed weights; the epilogue multiplies the packed words with the forward's
//     sign bits, and the bias gradients (column sums of every step's input) run on the matrix cores too: a 32 x 16
//     piece goes through 1 KB of LDS, comes back transposed (ds_read_b64_tr_b16) and meets a one-hot B operand, so that
//     one 16 x 16 accumulator tile collects 16 pieces' sums; waves add their tiles to private slices in global memory.
//
// Forward: 8 waves of 32 rows (two per SIMD, 240 registers; the two waves of a SIMD store at different k-steps); backward: 4 waves of 64 rows (464 registers; with 32 rows
// it needs 18 registers more than a wave of 8 has).  256 rows per pass, one workgroup per CU, persistent.  Everything is
// unrolled per stack shape (RoCfg); the shapes the models use are instantiated in mlp_ro_*.hip, one kernel per file,
// any other stack runs on mlp_fused.hip.
//
// What bounds them (profiles/r04_mlp_ro.md): not the matrix pipe (MFMAs + epilogue alone: 1.2 ms of the forward's 2.4 at
// 2.56 M rows) and not HBM (7.2 GB written, exactly the algorithmic bytes, at 3 TB/s against 6.9 measured for a fill) but
// the CU's vector-memory path: a 16-byte-per-lane store touching 32 rows costs ~52 cycles of it, a 1 KB LDS-DMA piece
// ~15, and a pass issues 43 KB of them per 1664-cycle chunk.  Things that looked free and were not: nt stores (5.2 vs
// 2.5 ms: partial lines written through), a value used right behind its load (the compiler's wait counts the weight
// copies it cannot see), loop-invariant address arithmetic (hoisted out of the pass loop by the hundred, then spilled).
#pragma once
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "trs_common.hpp"

namespace trs {

typedef __attribute__((ext_vector_type(8))) __bf16 ro_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ro_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned ro_u32x4;
typedef __attribute__((ext_vector_type(4))) float ro_f32x4;

constexpr int RO_ROWS = 256;                    // rows per workgroup pass: 8 waves x 32 or 4 waves x 64
constexpr int RO_MAXL = 8;
constexpr int RO_MAXKS = 26;                    // widths up to 416
constexpr int RO_MAXCT = RO_MAXKS / 2;
constexpr int RO_SLOTS = 3;                     // weight ring
constexpr int RO_AHEAD = 2;                     // chunks between a copy's issue and its first use
constexpr int RO_MASK_WORDS = 16 * 256;         // 32-bit words of sign bits per pass and layer (16 KB; layout below)
#ifndef TRS_RO_PF
#define TRS_RO_PF 2      // weight fragments read ahead of the MFMAs (3: six registers more, and they are not there)
#endif
constexpr int RO_PF = TRS_RO_PF;
#ifndef TRS_RO_PFB
#define TRS_RO_PFB 2      // the same in the backward kernels
#endif
#ifndef TRS_RO_STAGGER
#define TRS_RO_STAGGER 2      // groups of waves that store at different k-steps (forward; 1: 2.35 ms, 2: 2.24, 4: 2.26)
#endif
#ifndef TRS_RO_NREG
#define TRS_RO_NREG 9
#endif
// chunks of a layer's output that wait for the next layer in registers; the ones before them wait in LDS (64 bytes per
// chunk and row)
__host__ __device__ constexpr int ro_nreg(bool, int) { return TRS_RO_NREG; }
#ifndef TRS_RO_NT
#define TRS_RO_NT 0      // cache policy of the output stores (2 = nt: measured 2x SLOWER, 5.2 vs 2.5 ms -- partial lines written through)
#endif
#ifndef TRS_RO_ABL
#define TRS_RO_ABL 0      // timing experiments (wrong results): 1 no weight copies, 2 no stores, 4 no wait + barrier,
#endif                    // 16 no fragment reads, 32 no column sums, 64 no sign bits (backward)

template <int V>
using ro_ic = std::integral_constant<int, V>;
template <int I, int N, class F>
__device__ __forceinline__ void ro_for(F&& f) {
  if constexpr (I < N) {
    f(ro_ic<I>{});
    ro_for<I + 1, N>(f);
  }
}

// D row rho of a 32 x 32 tile sits in register i = (rho & 3) + 4 * (rho >> 3) of lane group g = (rho >> 2) & 1; feeding
// weight row ro_col_of_row(ct, rho) as A row rho gives lane group g, register i column 32 ct + 16 (i >> 3) + 8 g + (i & 7)
__host__ __device__ __forceinline__ int ro_col_of_row(int ct, int rho) {
  const int g = (rho >> 2) & 1, i = (rho & 3) + 4 * (rho >> 3);
  return 32 * ct + 16 * (i >> 3) + 8 * g + (i & 7);
}

// widths of a stack (<= 416); layer l: K = W[l] in KS = ceil(K / 16) k-steps (at least 2; the rows it reads are zero
// beyond K), N = W[l+1] in CT = ceil(N / 32) chunks of 32 columns (the weights are zero beyond N)
template <int... Ws>
struct RoCfg {
  static constexpr int L = sizeof...(Ws) - 1;
  static constexpr int w(int i) {
    constexpr int a[] = {Ws...};
    return a[i];
  }
  static constexpr int ks(int l) { return w(l) <= 16 ? 2 : (w(l) + 15) / 16; }
  static constexpr int ct(int l) { return (w(l + 1) + 31) / 32; }
  static constexpr int first_chunk(int l) {
    int c = 0;
    for (int i = 0; i < l; ++i) c += ct(i);
    return c;
  }
  static constexpr int NC = first_chunk(L);
  static constexpr int max_ks() {
    int m = 0;
    for (int l = 0; l < L; ++l) m = ks(l) > m ? ks(l) : m;
    return m;
  }
  static constexpr int layer_of(int c) {
    int l = 0;
    while (c >= ct(l)) c -= ct(l++);
    return l;
  }
  static constexpr int ct_of(int c) { return c - first_chunk(layer_of(c)); }
  static constexpr int boff(int l) {          // offset of layer l's output columns among all layers' (bias vector, padded to 32)
    int b = 0;
    for (int i = 0; i < l; ++i) b += 32 * ct(i);
    return b;
  }
  static constexpr int NB = boff(L);
  static constexpr bool ok() {
    for (int i = 0; i <= L; ++i)
      if (w(i) % 8 || w(i) < 8 || w(i) > 16 * RO_MAXKS) return false;
    return L >= 1 && L <= RO_MAXL;
  }
};

struct RoLayer {
  const char* wf;       // fragment-order weights: [chunk][k-step][lane][8 bf16]
  char* out;            // rows x out_stride bf16: the layer's output (hidden activations / d(pre-activation)), or the result
  uint32_t* mask;       // ReLU sign bits of the layer's output, [pass][chunk][thread] words (forward: written; backward: read)
  float* colsum;        // backward: [gridDim.x * waves][N] partial column sums of the layer's output, zeroed by the host (or null)
  int out_stride;       // elements per output row
  int out_cols;         // columns that exist in ``out`` (a multiple of 8)
};
struct RoArgs {
  RoLayer layer[RO_MAXL];
  const float* bias;    // forward: all layers' padded biases, back to back
  const char* in;       // rows x in_stride bf16
  int in_stride;        // = the input's logical width (a multiple of 8)
  int64_t rows;
  uint32_t* mask_in;    // forward: sign bits of the input rows (an upstream ReLU's output), same layout; may be null
  float* colsum_in;     // backward: [gridDim.x * waves][K0] partial column sums of the input rows, zeroed by the host
};

__device__ __forceinline__ unsigned ro_lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// one 1 KB piece (16 bytes per lane) from global memory (uniform ``src`` + this lane's ``voff``) straight into LDS at the
// wave-uniform address ``lds_dst``.  M0 belongs to the compiler: saved and restored inside the statement.
__device__ __forceinline__ void ro_dma1(const char* src, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(dst), "s"(src) : "memory");
}

// the same as a buffer operation (offsets past the end read nothing)
__device__ __forceinline__ void ro_dma1_buf(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(dst), "s"(rs) : "memory");
}

template <int N>
__device__ __forceinline__ void ro_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void ro_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned ro_pk_max0(unsigned w) {
  unsigned x;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(x) : "v"(w));
  return x;
}
__device__ __forceinline__ unsigned ro_pk_flag(unsigned w) {      // [x > 0] of two non-negative bf16: 1 / 0 per half
  unsigned t;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(w), "s"(0x00010001u));
  return t;
}

typedef __attribute__((ext_vector_type(4))) short ro_s16x4;
typedef __attribute__((ext_vector_type(8))) short ro_s16x8;

__device__ __forceinline__ unsigned ro_pk_mul_u16(unsigned a, unsigned b) {
  unsigned x;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(x) : "v"(a), "v"(b));
  return x;
}

// Sign bits of a pass (256 rows) and layer: 16 bits per lane, 32-row tile and chunk -- bit 4 h + k2 is the flag of column
// 32 chunk + 16 h + 8 g + 2 k2 (g: the lane group, = which of the row's two lanes), bit 8 + 4 h + k2 of the column after
// it -- in groups of four chunks: [pass][chunk / 4][slot = 2 (64 (row / 64) + lane) + (row / 32) % 2][chunk % 4], 8 bytes
// per slot and group.  A vector-memory instruction costs what it costs whether it moves 2 or 16 bytes per lane (one
// 256-byte load per chunk was 0.3 of the backward's 3.0 ms), and what it costs grows with the number of separate
// segments it touches (32-byte records per lane, two 16-byte stores each: forward 2.39 -> 2.54 ms): the forward (32 rows
// per wave: one slot per lane) writes 512 contiguous bytes per wave and group, the backward (64 rows per wave: two
// adjacent slots per lane) fetches 1 KB per wave and group by LDS-DMA and picks the chunks' bits out of LDS.
__host__ __device__ constexpr int ro_flag_bit(int t, int k2) { return 16 * (t >> 1) + 4 * (t & 1) + k2; }      // t = 2 r + h

// RT: 32-row tiles per wave -- 2: four waves (one per SIMD, 512 registers) of 64 rows, every weight fragment read from LDS
// feeds two MFMAs; 1: eight waves (two per SIMD, 256 registers) of 32 rows -- twice the LDS reads, but a wave that is
// stuck issuing a store or a copy (~60 cycles each, 12 per chunk) leaves the matrix pipe to its neighbour.
// IN_COLS: columns every input row is known to have (the host picks the instantiation).
template <class Cfg, bool BWD, int IN_COLS, int RT>
__global__ __launch_bounds__(512 / RT, 1) void mlp_ro_kernel(RoArgs a) {
  static_assert(Cfg::ok(), "stack shape");
  constexpr int L = Cfg::L, NC = Cfg::NC, NB = Cfg::NB;
  constexpr int NW = 8 / RT, NT = 64 * NW;             // waves, threads
  constexpr int NQ = 2 * RT;                           // quarters (t = 2 r + h) of a chunk's 32 columns x 32 RT rows per lane
  constexpr int NREGK = ro_nreg(BWD, RT), NLDSK = RO_MAXCT - NREGK;
  constexpr int STASH_WAVE = NLDSK * 2048 * RT;
  constexpr int SLOT = Cfg::max_ks() * 1024;          // bytes per ring slot
  constexpr unsigned RING = RO_SLOTS * SLOT;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [weight ring][the waves' stash][bias]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 5;
  float* bias_s = reinterpret_cast<float*>(smem + RING + NW * STASH_WAVE);      // backward: 1 KB per wave for the column sums instead
  if constexpr (!BWD)
    for (int i = threadIdx.x; i < NB; i += NT) bias_s[i] = a.bias[i];
  const int64_t ntiles = (a.rows + RO_ROWS - 1) / RO_ROWS;
  if ((int64_t)blockIdx.x >= ntiles) return;

  // ---- weight stream: chunk q (counted from this workgroup's first pass on) lives in ring slot q % 3.  The ring
  // positions are carried along (one add and a wrap per chunk) instead of computed from q: everything below is one
  // basic block per pass, and 41 chunks' worth of independent address arithmetic is what the scheduler hoists and spills.
  const unsigned ring = ro_lds_addr(smem);
  const unsigned lane16 = lane * 16;
  unsigned dma_at = ring;                               // slot the next copy goes to (wave-uniform)
  unsigned pass_zero = 0;      // 0, redefined per pass behind an asm, like the weight pointers: keeps ~300 loop-invariant
  const char* wfl[L];          // addresses from being hoisted out of the pass loop (and spilled)
#pragma unroll
  for (int l = 0; l < L; ++l) wfl[l] = a.layer[l].wf;
  auto slot_inc = [](unsigned x, unsigned base) { x += SLOT; return x >= base + RING ? x - RING : x; };
  auto dma_piece = [&](auto cq_, int i) __attribute__((always_inline)) {      // piece i of this wave's share of chunk cq
    constexpr int cq = decltype(cq_)::value, c2 = cq % NC, l2 = Cfg::layer_of(c2), ct2 = Cfg::ct_of(c2), P = Cfg::ks(l2);
    const int p0 = (P * wave) / NW, p1 = (P * (wave + 1)) / NW;      // the chunk's P (= its layer's KS) KB are split over the waves
    if (!(TRS_RO_ABL & 1) && p0 + i < p1)
      ro_dma1(wfl[l2] + (unsigned)((ct2 * P + p0 + i) * 1024), lane16, dma_at + (p0 + i) * 1024);
  };
  constexpr int MAXP = (RO_MAXKS + NW - 1) / NW;      // pieces per wave and chunk, at most
  ro_for<0, RO_AHEAD>([&](auto cq) __attribute__((always_inline)) {
    for (int i = 0; i < MAXP; ++i) dma_piece(cq, i);
    dma_at = slot_inc(dma_at, ring);
  });

  // ---- the rows of a pass: lane (n, g) of wave w holds row 32 RT w + 32 r + n, columns 16 ks + 8 g .. + 7 as B[r][ks].
  // Global accesses are buffer operations (uniform descriptor + one 32-bit byte offset per lane, row tile and tensor +
  // a constant; the host checks that every tensor is below 4 GB): rows past the end read zeros and store nothing.
  const unsigned rowl = 32 * RT * wave + (lane & 31);            // + 32 r: this lane's row inside the pass
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.in), 0, (unsigned)(a.rows * a.in_stride * 2), 0x00020000);
  ro_u32x4 B[RT][RO_MAXKS], Bn[RT][RO_MAXKS];
  auto in_off = [&](int64_t row0, int r) { return (unsigned)((row0 + rowl + 32 * r) * a.in_stride + 8 * g) * 2u; };
  auto load_in = [&](const unsigned (&oin)[RT], auto r_, auto j_) __attribute__((always_inline)) {
    constexpr int r = decltype(r_)::value, j = decltype(j_)::value;
    unsigned off = oin[r] + 32 * j;
    if constexpr (16 * j + 16 > IN_COLS) off = 16 * j + 8 * g < a.in_stride + (int)pass_zero ? off : 0xfffffff0u;      // columns past the row: zeros
    return __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
  };
  unsigned off_in[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) off_in[r] = in_off((int64_t)blockIdx.x * RO_ROWS, r);
  ro_for<0, Cfg::ks(0)>([&](auto j) __attribute__((always_inline)) {
    ro_for<0, RT>([&](auto r) __attribute__((always_inline)) { B[r][j] = load_in(off_in, r, j); });
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  unsigned rd_at = lane16;         // this lane's byte of the slot of the current chunk (relative to smem)
  unsigned stash_at = RING + wave * STASH_WAVE + lane16;      // this lane's 16 bytes of a (chunk, quarter) of the stash
  unsigned bias_at = RING + NW * STASH_WAVE + 32 * g;         // this lane's first bias column
  // (opaque: with the constants visible the compiler folds them into ~300 absolute LDS addresses, hoists those out of the
  // pass loop into scalar registers and spills them; as it is they are one register + an immediate offset each)
  asm volatile("" : "+v"(stash_at), "+v"(bias_at));
  auto lds_frag = [&](unsigned at, int ks) __attribute__((always_inline)) {
    if constexpr (TRS_RO_ABL & 16) return ro_u32x4{at, (unsigned)ks, at, at};
    else return *reinterpret_cast<const ro_u32x4*>(smem + at + ks * 1024);
  };
  auto bias16 = [&](int col0) __attribute__((always_inline)) {      // columns col0 + 8 g + (0..7), col0 + 16 + 8 g + (0..7)
    const float* p = reinterpret_cast<const float*>(smem + bias_at) + col0;
    const ro_f32x4 b0 = *reinterpret_cast<const ro_f32x4*>(p), b1 = *reinterpret_cast<const ro_f32x4*>(p + 4);
    const ro_f32x4 b2 = *reinterpret_cast<const ro_f32x4*>(p + 16), b3 = *reinterpret_cast<const ro_f32x4*>(p + 20);
    return ro_f32x16{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], b2[0], b2[1], b2[2], b2[3], b3[0], b3[1], b3[2], b3[3]};
  };
  const ro_f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // values carried from chunk to chunk
  ro_f32x16 accp[RT];         // the previous chunk's results, epilogue pending
  ro_u32x4 afn[RO_PF];        // the next chunk's first weight fragments
  ro_f32x16 init;             // the next chunk's initial accumulators (bias): the C operand of its first MFMAs
  {
    constexpr int PF0 = Cfg::ks(0) < (BWD ? TRS_RO_PFB : RO_PF) ? Cfg::ks(0) : (BWD ? TRS_RO_PFB : RO_PF);
    ro_for<0, PF0>([&](auto i) __attribute__((always_inline)) { afn[i] = lds_frag(rd_at, i); });
    if constexpr (!BWD) init = bias16(0);
    else init = zero16;
#pragma unroll
    for (int r = 0; r < RT; ++r) accp[r] = init;      // the first pass starts with an "epilogue" of nothing: defined values, stores switched off
  }
  bool have_prev = false;
  unsigned mtile_prev = 0;
  // offsets of this lane's two stores per row tile (see epi_store): rows (n % 16) and (n % 16) + 16 of the tile, column
  // half n / 16
  unsigned off_out[L][RT][2], off_last_prev[RT][2];
#pragma unroll
  for (int r = 0; r < RT; ++r) off_last_prev[r][0] = off_last_prev[r][1] = 0;

  // ---- backward: the sign bits for the pending epilogue, and the column sums (= bias gradients) of what the steps
  // write.  A quarter (32 rows x 16 columns: this lane's 16 bytes and its partner's) goes through 1 KB of LDS and comes
  // back transposed (ds_read_b64_tr_b16) as the A operand of one 16x16x32 MFMA whose B operand is 1 in column (unit % 16)
  // and 0 elsewhere: a 16 x 16 accumulator tile collects the sums of 16 units x 16 columns, a 416-wide output takes two.
  // At the end of a step a wave adds its tiles to ITS slice of the partial sums in global memory (fixed order: the
  // bias gradients are reproducible bit for bit).
  unsigned mcur = 0xffffffffu;
  const unsigned mstage = RING + NW * STASH_WAVE + NW * 1024 + wave * 4096;      // this wave's stage for a layer's sign bits
  ro_f32x4 cs = {0.f, 0.f, 0.f, 0.f};       // sums of the running step's input ...
  ro_f32x4 cso = {0.f, 0.f, 0.f, 0.f};      // ... and, in the last step, of its output: both are under way at once
  unsigned scr_wr = RING + NW * STASH_WAVE + wave * 1024 + (2 * (lane & 31) + g) * 16;
  unsigned scr_lo, scr_hi;
  {
    const int q4 = lane >> 4, i = lane & 15;
    const unsigned fl = RING + NW * STASH_WAVE + wave * 1024 + (8 * q4 + (i >> 2)) * 32 + (i & 3) * 8;
    scr_lo = fl + ((q4 & 1) ? 128 : 0);
    scr_hi = fl + ((q4 & 1) ? 0 : 128);
  }
  asm volatile("" : "+v"(scr_wr), "+v"(scr_lo), "+v"(scr_hi));
  const unsigned sel_lane = lane & 15;
  ro_s16x8 cs_av;      // a unit on its way: transposed, waiting for its MFMA
  auto cs_a = [&](const ro_u32x4& pk) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) ro_s16x4* lds_p;
    *reinterpret_cast<ro_u32x4*>(smem + scr_wr) = pk;
    const ro_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem + scr_lo));
    const ro_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem + scr_hi));
    cs_av = ro_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto cs_b = [&](int unit, ro_f32x4& cs) __attribute__((always_inline)) {
    // (pass_zero: 16 loop-invariant selector vectors would otherwise be built once, kept in 64 registers and spilled)
    unsigned one = (sel_lane | pass_zero) == (unsigned)(unit & 15) ? 0x3f803f80u : 0u;
    asm volatile("" : "+v"(one));      // never one value for two units: it would be kept (in scratch) from layer to layer
    const ro_u32x4 bv = {one, one, one, one};
    cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ro_bf16x8, cs_av), __builtin_bit_cast(ro_bf16x8, bv), cs, 0, 0, 0);
  };
  // cs (units 16 t .. 16 t + 15) is added to this wave's slice of ``part`` ([gridDim.x * NW][n]): lane (n', q) holds
  // columns 16 (16 t + n') + 4 q .. + 3.  One tile is live at a time: a step's sums are flushed after unit 15 and at the
  // end.  Atomic adds nobody else touches: nothing to wait for, and one wave's adds to an address keep their order.
  auto cs_flush = [&](float* part, int n, int t, ro_f32x4& cs) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part, 0, (unsigned)(gridDim.x * NW * n * 4), 0x00020000);
    const unsigned col = 256 * t + (lane & 15) * 16 + (lane >> 4) * 4;
    const unsigned o = col + 4 <= (unsigned)n ? ((blockIdx.x * NW + wave) * n + col) * 4 : 0xfffffff0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(cs[e], rs, o + 4 * e, 0, 0);
    cs = ro_f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- epilogue of chunk pc in 4 NQ steps (u = 4 t + k2: quarter t = 2 r + h, word k2 = two columns) + one finish per quarter
  unsigned pw[NQ][4], mq[NQ];
  auto epi_word = [&](auto pc_, auto u_) __attribute__((always_inline)) {
    constexpr int pc = decltype(pc_)::value, u = decltype(u_)::value, t = u >> 2, k2 = u & 3, r = t >> 1, h = t & 1;
    constexpr bool hidden = Cfg::layer_of(pc) + 1 < L;
    unsigned w = f32x2_to_bf16x2_bits(accp[r][8 * h + 2 * k2], accp[r][8 * h + 2 * k2 + 1]);
    if constexpr (!BWD && hidden) {
      w = ro_pk_max0(w);
      const unsigned f = ro_pk_flag(w);
      mq[t] = k2 == 0 ? f : (mq[t] | (f << k2));
    }
    if constexpr (BWD && !(TRS_RO_ABL & (64 | 128))) {      // times [forward activation > 0], two columns per instruction (all ones when the step has no mask)
      // (mcur: see the end of the chunk body -- a word's two flags are 16 bits apart)
      const unsigned e = (mcur >> (8 * r + 4 * h + k2)) & 0x00010001u;
      w = ro_pk_mul_u16(w, e);
    }
    pw[t][k2] = w;
  };
  // where a quarter's 8 columns go for the next layer: the last chunk's straight into B (its epilogue runs inside the
  // next layer), the NREG-1 chunks before it into Bn (handed over at the end of the layer), earlier ones into LDS
  auto epi_dest = [&](auto pc_, auto t_) __attribute__((always_inline)) {
    constexpr int pc = decltype(pc_)::value, t = decltype(t_)::value, r = t >> 1, h = t & 1;
    constexpr int pl = Cfg::layer_of(pc), pct = Cfg::ct_of(pc), CTp = Cfg::ct(pl);
    constexpr int NL = CTp > NREGK ? CTp - NREGK : 0;
    if constexpr (pl + 1 < L) {
      if constexpr (2 * pct + h < Cfg::ks(pl + 1)) {      // (a width of 400: the last chunk's second half is padding)
        const ro_u32x4 pk = {pw[t][0], pw[t][1], pw[t][2], pw[t][3]};
        if constexpr (pct == CTp - 1) B[r][2 * pct + h] = pk;
        else if constexpr (pct >= NL) Bn[r][2 * pct + h] = pk;
        else *reinterpret_cast<ro_u32x4*>(smem + stash_at + (pct * NQ + t) * 1024) = pk;
      }
    }
  };
  auto epi_colsum = [&](auto pc_, auto t_) __attribute__((always_inline)) {
    constexpr int pc = decltype(pc_)::value, t = decltype(t_)::value;
    constexpr int pl = Cfg::layer_of(pc), pct = Cfg::ct_of(pc), CTp = Cfg::ct(pl);
    const RoLayer& ly = a.layer[pl];
    // a step's input is summed while the step runs (below); what the LAST step writes is nobody's input: its sums (wanted
    // when the stack's input came out of a ReLU) are taken here
    if constexpr (pl == L - 1) {
      if (ly.colsum != nullptr) {
        constexpr int unit = 2 * pct + (t & 1);
        cs_a(ro_u32x4{pw[t][0], pw[t][1], pw[t][2], pw[t][3]});
        cs_b(unit, cso);
        if constexpr (t >= NQ - 2 && ((unit & 15) == 15 || (pct == CTp - 1 && t == NQ - 1))) cs_flush(ly.colsum, 32 * CTp, unit >> 4, cso);
      }
    }
  };
  // The stores of a row tile's 32 x 32 results.  As they come out of the MFMA a lane holds 2 x 16 bytes of ONE row, so a
  // 16-byte-per-lane store touches 32 rows with 32 bytes each -- and the vector-memory path pays per cache line touched
  // (~52 cycles for such a store against ~15 for 1 KB in one piece).  v_permlane16_swap trades the second column half of
  // lanes n < 16 for the first of lanes n + 16: then one store writes rows 0..15 of the tile, 64 bytes each, the other
  // rows 16..31.
  auto epi_store = [&](auto pc_, auto t_, const unsigned (&off)[RT][2], unsigned& mw, bool on) __attribute__((always_inline)) {
    constexpr int pc = decltype(pc_)::value, t = decltype(t_)::value, r = t >> 1, h = t & 1;
    constexpr int pl = Cfg::layer_of(pc), pct = Cfg::ct_of(pc);
    constexpr bool hidden = pl + 1 < L;
    const RoLayer& ly = a.layer[pl];
    if constexpr (!BWD && hidden) mw |= (mq[t] & 0xfu) << ro_flag_bit(t, 0) | (mq[t] >> 16) << (ro_flag_bit(t, 0) + 8);
    if constexpr (h == 1) {
      ro_u32x4 lo, hi;      // after the swap: rows n % 16 / n % 16 + 16, column half n / 16
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        const auto sw = __builtin_amdgcn_permlane16_swap(pw[t - 1][k2], pw[t][k2], false, false);
        lo[k2] = sw[0];
        hi[k2] = sw[1];
      }
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ly.out, 0, (unsigned)(a.rows * ly.out_stride * 2), 0x00020000);
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        unsigned o = off[r][sidx] + 32 * pct * 2;
        if constexpr (!hidden) o = 32 * pct + 16 * ((lane >> 4) & 1) + 8 * g < ly.out_cols ? o : 0xfffffff0u;
        if constexpr (TRS_RO_ABL & 2) o = lo[0] == 0x12345678u ? o : 0xfffffff0u;
        o = on ? o : 0xfffffff0u;      // past the end of every tensor: dropped
        __builtin_amdgcn_raw_buffer_store_b128(sidx == 0 ? lo : hi, rs, o, 0, TRS_RO_NT);
      }
    }
  };
  static_assert(BWD ? RT == 2 : RT == 1, "sign bits: written by 32-row waves (one slot per lane), read by 64-row waves (two)");
  // forward: chunk pct's 16 flags of this lane join the record under construction; 4 chunks (or the layer's last ones)
  // leave as one 8-byte store.  ``mrec``: this lane's slot of the pass's first group (byte offset into the layer's mask)
  unsigned mreg[2];
  auto mask_store = [&](auto pc_, unsigned mrec, unsigned mw, bool on) __attribute__((always_inline)) {
    constexpr int pc = decltype(pc_)::value, pl = Cfg::layer_of(pc), pct = Cfg::ct_of(pc), CTp = Cfg::ct(pl);
    if constexpr (!BWD && pl + 1 < L) {
      constexpr int sl = pct & 3;
      if constexpr ((sl & 1) == 0) mreg[sl >> 1] = mw & 0xffffu;
      else mreg[sl >> 1] |= mw << 16;
      if constexpr (sl == 3 || pct == CTp - 1) {
        if constexpr (sl < 2) mreg[1] = 0;
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(a.layer[pl].mask, 0, (unsigned)(ntiles * RO_MASK_WORDS * 4), 0x00020000);
        const unsigned o = (on && (!(TRS_RO_ABL & 2) || mw == 0x13572468u)) ? mrec + (pct >> 2) * 4096 : 0xfffffff0u;
        typedef __attribute__((ext_vector_type(2))) unsigned ro_u32x2;
        __builtin_amdgcn_raw_buffer_store_b64(ro_u32x2{mreg[0], mreg[1]}, rm, o, 0, 0);
      }
    }
  };

  // backward: a layer's sign bits of this wave (4 groups of 1 KB) by LDS-DMA into the wave's 4 KB stage --
  // requested at the first k-step of the layer's first chunk, landed at that chunk's barrier (its wait), and read chunk
  // by chunk at the end of every chunk for the epilogue that runs in the next one.  Through LDS and not into registers:
  // a register written by a hand-issued load must not be touched before the wait, and nothing keeps the compiler from
  // moving a value it believes defined (to an AGPR, say) while the load is in flight; as compiler-visible loads their
  // wait in front of the first use counts the weight copies the compiler cannot see and parks the wave for them
  // (1.2 of 4.4 ms).
  auto mask_has = [&](auto l_) __attribute__((always_inline)) { return decltype(l_)::value + 1 < L || a.layer[decltype(l_)::value].mask != nullptr; };
  auto mask_load = [&](auto l_, unsigned mrec) __attribute__((always_inline)) {
    constexpr int l = decltype(l_)::value;
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(a.layer[l].mask, 0, (unsigned)(ntiles * RO_MASK_WORDS * 4), 0x00020000);
    if (mask_has(l_)) {
#pragma unroll
      for (int q = 0; q < 4; ++q)      // piece q: chunks 4 q .. 4 q + 3, this lane's two slots (16 bytes)
        if (4 * q < Cfg::ct(l)) ro_dma1_buf(rm, mrec + q * 4096, ring + mstage + q * 1024);
    }
  };

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    asm volatile("s_mov_b32 %0, 0" : "=s"(pass_zero));
#pragma unroll
    for (int l = 0; l < L; ++l) {
      wfl[l] = a.layer[l].wf;
      asm volatile("" : "+s"(wfl[l]));
    }
    const int64_t row0 = tile * RO_ROWS;
    // this lane's (first) sign-bit slot of the pass's first group: byte offset into a layer's mask
    const unsigned mtile = ((unsigned)tile * 2048 + (RT == 2 ? 2 * threadIdx.x : 2 * (64 * (wave >> 1) + lane) + (wave & 1))) * 8;
    const unsigned row_here = (unsigned)row0 + rowl;
    unsigned off_in_next[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) off_in_next[r] = in_off(row0 + (int64_t)gridDim.x * RO_ROWS, r);

    if (!BWD && a.mask_in != nullptr) {
      // the input rows are a ReLU's output: their sign bits, in the layout of a layer output of the same width
      unsigned mi[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
      ro_for<0, (Cfg::ks(0) + 1) / 2>([&](auto ct_) __attribute__((always_inline)) {
        constexpr int ct = decltype(ct_)::value;
        unsigned mw = 0;
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
          if (2 * ct + (t & 1) >= Cfg::ks(0)) continue;
          const ro_u32x4 v = B[t >> 1][2 * ct + (t & 1)];
          unsigned m = 0;
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2) m |= ro_pk_flag(ro_pk_max0(v[k2])) << k2;
          mw |= (m & 0xfu) << ro_flag_bit(t, 0) | (m >> 16) << (ro_flag_bit(t, 0) + 8);
        }
        mi[ct >> 1] |= (mw & 0xffffu) << (16 * (ct & 1));
      });
      const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(a.mask_in, 0, (unsigned)(ntiles * RO_MASK_WORDS * 4), 0x00020000);
      typedef __attribute__((ext_vector_type(2))) unsigned ro_u32x2;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (8 * q < Cfg::ks(0)) __builtin_amdgcn_raw_buffer_store_b64(ro_u32x2{mi[2 * q], mi[2 * q + 1]}, rm, mtile + q * 4096, 0, 0);
    }

    ro_for<0, NC>([&](auto c_) __attribute__((always_inline)) {
      constexpr int c = decltype(c_)::value;
      constexpr int l = Cfg::layer_of(c), ct = Cfg::ct_of(c), KS = Cfg::ks(l), CT = Cfg::ct(l);
      constexpr int MID = KS / 2;
      constexpr int PFD = BWD ? TRS_RO_PFB : RO_PF;
      constexpr int PF = KS < PFD ? KS : PFD;
      constexpr int cn = (c + 1) % NC, ln = Cfg::layer_of(cn), ctn = Cfg::ct_of(cn), KSn = Cfg::ks(ln);
      constexpr int PFn = KSn < PFD ? KSn : PFD;
      constexpr int pc = (c + NC - 1) % NC, pl = Cfg::layer_of(pc);
      constexpr bool last_of_layer = ct == CT - 1;
      constexpr int NLl = (last_of_layer && l + 1 < L && CT > NREGK) ? CT - NREGK : 0;      // chunks to fetch back from LDS here
      constexpr int KSnl = l + 1 < L ? Cfg::ks(l + 1) : 0;
      // the next pass's input is requested during the last layer: in its first chunk when it is narrow (it waits in Bn,
      // which the last layer does not use), otherwise in its last chunk, into the registers this pass's input leaves
      constexpr int KS0 = Cfg::ks(0);
      constexpr bool fetch_in = l == L - 1 && (KS0 <= 8 ? ct == 0 : last_of_layer);
      const unsigned rd_next = slot_inc(rd_at, lane16);

      if constexpr (ct == 0) {      // this lane's offsets into the layer's output (first needed by the next chunk's stores)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int sidx = 0; sidx < 2; ++sidx)
            off_out[l][r][sidx] = ((row_here - (lane & 16) + 32 * r + 16 * sidx) * a.layer[l].out_stride + 8 * g + (lane & 16)) * 2u;
      }
      ro_u32x4 af[KS];
      ro_for<0, PF>([&](auto i) __attribute__((always_inline)) { af[i] = afn[i]; });
      ro_f32x16 acc[RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r] = BWD ? zero16 : init;
      unsigned mw = 0;

      // The previous chunk's epilogue runs in this chunk's FIRST half, word u behind the first MFMA of k-step
      // 1 + u (MID-1) / NU, and its stores right behind the barrier.  In the first chunk of a layer the quarters ARE
      // this layer's last input fragments (k-steps 2 (CT' - 1) + h): a word that would come too late moves in front of
      // the MFMA that needs it.
      constexpr int NU = 4 * NQ;
      constexpr bool feeds = c > 0 && pl != l;
      auto word_nat = [](int u) constexpr {
        const int s = 1 + (u * (MID - 1)) / NU;
        return s > MID - 1 ? MID - 1 : s;
      };
      constexpr int NEED0 = 2 * (Cfg::ct(pl) - 1);      // the previous chunk's quarter h is fragment NEED0 + h of this layer's input
      auto word_step = [word_nat](int u) constexpr {
        const int t = u >> 2, r = t >> 1, h = t & 1, nat = word_nat(u), need = NEED0 + h;
        return (feeds && need < KS && ((r == 0 && nat >= need) || (r == 1 && nat > need))) ? need : nat;
      };
      auto word_pre = [word_nat](int u) constexpr {      // in front of the k-step's first MFMA instead of behind it
        const int t = u >> 2, r = t >> 1, h = t & 1, nat = word_nat(u), need = NEED0 + h;
        return feeds && need < KS && r == 0 && nat >= need;
      };
      auto store_step = [](int t) constexpr {
        const int s = MID + t / 2;
        return s > KS - 1 ? KS - 1 : s;
      };
      auto words = [&](auto ks_, auto pre_) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_)::value;
        constexpr bool pre = decltype(pre_)::value != 0;
        ro_for<0, NU>([&](auto u_) __attribute__((always_inline)) {
          constexpr int u = decltype(u_)::value;
          if constexpr (word_step(u) == ks && word_pre(u) == pre) {
            epi_word(ro_ic<pc>{}, u_);
            if constexpr ((u & 3) == 3) {
              epi_dest(ro_ic<pc>{}, ro_ic<(u >> 2)>{});
              if constexpr (BWD) epi_colsum(ro_ic<pc>{}, ro_ic<(u >> 2)>{});
            }
          }
        });
      };

      ro_for<0, KS>([&](auto ks_) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_)::value;
        if constexpr (BWD && ks == 0 && ct == 0 && !(TRS_RO_ABL & 64)) mask_load(ro_ic<l>{}, mtile);
        words(ks_, ro_ic<1>{});
        if constexpr (ks + PF < KS) af[ks + PF] = lds_frag(rd_at, ks + PF);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ro_bf16x8, af[ks]), __builtin_bit_cast(ro_bf16x8, B[0][ks]), acc[0], 0, 0, 0);
        words(ks_, ro_ic<0>{});
        if constexpr (RT == 2)
          acc[RT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ro_bf16x8, af[ks]), __builtin_bit_cast(ro_bf16x8, B[RT - 1][ks]), acc[RT - 1], 0, 0, 0);
        if constexpr (ks == MID - 1) {
          // chunk c+1's copy (issued a chunk ago) must have landed in every wave before anybody reads it
          if constexpr (!(TRS_RO_ABL & 4)) {
            ro_wait_vm<0>();
            ro_barrier();
          }
        }
        if constexpr (ks >= MID - 1) {
          // after the barrier: chunk c-1's slot is free -> copy chunk c+2 into it, a piece or so per k-step
          constexpr int NP = (Cfg::ks(Cfg::layer_of((c + RO_AHEAD) % NC)) + NW - 1) / NW;
          constexpr int per = (NP + (KS - MID + 1) - 1) / (KS - MID + 1);
#pragma unroll
          for (int i = per * (ks - (MID - 1)); i < per * (ks - (MID - 1) + 1) && i < NP; ++i) dma_piece(ro_ic<c + RO_AHEAD>{}, i);      // (staggering these between the waves: no change)
        }
        if constexpr (ks >= MID) {
#if TRS_RO_STAGGER > 1
          // the waves do not store at the same k-step: group gq (of TRS_RO_STAGGER; the two waves of a SIMD are in different
          // groups) a share of the half chunk later, so that they are not parked on the memory path together
          // (one wave per SIMD, the backward: 2 or 4 groups measured no different from none)
          constexpr int NG = (RT == 1 && KS - MID >= 8) ? TRS_RO_STAGGER : 1;
          const int gq = NG == 1 ? 0 : (NG == 2 ? wave >> 2 : 2 * (wave >> 2) + (wave & 1));
          ro_for<0, NQ>([&](auto t_) __attribute__((always_inline)) {
            ro_for<0, NG>([&](auto g_) __attribute__((always_inline)) {
              constexpr int gg = decltype(g_)::value;
              if constexpr (store_step(decltype(t_)::value) + (gg * (KS - MID - 2)) / NG == ks) {
                if (gq == gg) {
                  if constexpr (c > 0) epi_store(ro_ic<pc>{}, t_, off_out[pl], mw, true);
                  else epi_store(ro_ic<pc>{}, t_, off_last_prev, mw, have_prev);
                }
              }
            });
          });
#else
          ro_for<0, NQ>([&](auto t_) __attribute__((always_inline)) {
            if constexpr (store_step(decltype(t_)::value) == ks) {
              if constexpr (c > 0) epi_store(ro_ic<pc>{}, t_, off_out[pl], mw, true);
              else epi_store(ro_ic<pc>{}, t_, off_last_prev, mw, have_prev);
            }
          });
#endif
          // backward: column sums of this step's input, a few 16-column units per chunk while the registers are not yet
          // full of the next step's input; units in ascending order (the last two arrive with chunk 0's first half)
          if constexpr (BWD && !(TRS_RO_ABL & 32)) {
            constexpr int NCH = CT < 9 ? CT : 9, UPC = (KS + NCH - 1) / NCH, SU = UPC * RT;
            // piece m = i RT + r (unit j = ct UPC + i of row tile r): into LDS and back at k-step pa(m), its MFMA two
            // k-steps later (the LDS round trip), but not behind the next piece's turn
            auto pa = [](int m) constexpr { return MID + (m * (KS - MID)) / SU; };
            auto pb = [pa](int m) constexpr {
              int p = pa(m) + 2;
              if (m + 1 < SU && p > pa(m + 1)) p = pa(m + 1);
              return p > KS - 1 ? KS - 1 : p;
            };
            auto finish = [&](auto m_) __attribute__((always_inline)) {
              constexpr int m = decltype(m_)::value, j = ct * UPC + m / RT, r = m % RT;
              cs_b(j, cs);
              if constexpr (r == RT - 1 && ((j & 15) == 15 || j == KS - 1)) {
                if constexpr (l == 0) cs_flush(a.colsum_in, 32 * ((Cfg::w(0) + 31) / 32), j >> 4, cs);
                else cs_flush(a.layer[l - 1].colsum, 32 * Cfg::ct(l - 1), j >> 4, cs);
              }
            };
            ro_for<0, SU>([&](auto m_) __attribute__((always_inline)) {
              constexpr int m = decltype(m_)::value, j = ct * UPC + m / RT;
              if constexpr (j < KS && pb(m) == ks && pa(m) != ks) finish(m_);
            });
            ro_for<0, SU>([&](auto m_) __attribute__((always_inline)) {
              constexpr int m = decltype(m_)::value, j = ct * UPC + m / RT, r = m % RT;
              if constexpr (j < KS && pa(m) == ks) {
                cs_a(B[r][j]);
                if constexpr (pb(m) == ks) finish(m_);      // (no later k-step left: both halves here)
              }
            });
          }
          // the part of the next layer's input that waited in LDS
          if constexpr (NLl > 0) {
            ro_for<0, NQ * NLl>([&](auto q_) __attribute__((always_inline)) {
              constexpr int q = decltype(q_)::value, pct = q / NQ, t = q % NQ;
              if constexpr (MID + (q * (KS - MID)) / (NQ * NLl) == ks)
                Bn[t >> 1][2 * pct + (t & 1)] = *reinterpret_cast<const ro_u32x4*>(smem + stash_at + q * 1024);
            });
          }
          if constexpr (fetch_in) {
            ro_for<0, KS0>([&](auto j_) __attribute__((always_inline)) {
              constexpr int j = decltype(j_)::value;
              constexpr int pos = KS0 <= 8 ? MID + (j * (KS - MID)) / KS0 : (j < MID ? MID : (j > KS - 1 ? KS - 1 : j));
              if constexpr (pos == ks) ro_for<0, RT>([&](auto r) __attribute__((always_inline)) { Bn[r][j] = load_in(off_in_next, r, j_); });
            });
          }
          // the next chunk's first fragments and bias
          ro_for<0, PFn>([&](auto i_) __attribute__((always_inline)) {
            constexpr int i = decltype(i_)::value;
            constexpr int pos = (KS - PFn + i) < MID ? MID : (KS - PFn + i);
            if constexpr (pos == ks) afn[i] = lds_frag(rd_next, i);
          });
          if constexpr (!BWD && ks == (KS - 2 < MID ? MID : KS - 2)) init = bias16(Cfg::boff(ln) + 32 * ctn);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (c > 0) mask_store(ro_ic<pc>{}, mtile, mw, true);
      else mask_store(ro_ic<pc>{}, mtile_prev, mw, have_prev);
#pragma unroll
      for (int r = 0; r < RT; ++r) accp[r] = acc[r];
      if constexpr (BWD) {      // this chunk's flags (the two row tiles' 16 bits) for its epilogue in the next chunk
        const unsigned m0 = *reinterpret_cast<const unsigned short*>(smem + mstage + (ct >> 2) * 1024 + lane * 16 + (ct & 3) * 2);
        const unsigned m1 = *reinterpret_cast<const unsigned short*>(smem + mstage + (ct >> 2) * 1024 + lane * 16 + 8 + (ct & 3) * 2);
        // bytes [r0 low flags, r1 low flags, r0 high flags, r1 high flags]: a word's two flags 16 bits apart (epi_word)
        mcur = mask_has(ro_ic<l>{}) ? __builtin_amdgcn_perm(m1, m0, 0x05010400u) : 0xffffffffu;
      }
      rd_at = rd_next;
      dma_at = slot_inc(dma_at, ring);
      if constexpr (last_of_layer) {
        // the registers change hands: what waited in LDS or Bn (the last chunk's pair arrives with its epilogue, in the
        // next chunk); after the last layer, the next pass's input
        ro_for<0, (l + 1 < L ? (2 * CT - 2 < KSnl ? 2 * CT - 2 : KSnl) : KS0)>([&](auto j) __attribute__((always_inline)) {
          ro_for<0, RT>([&](auto r) __attribute__((always_inline)) { B[r][j] = Bn[r][j]; });
        });
      }
    });
    have_prev = true;
    mtile_prev = mtile;
#pragma unroll
    for (int r = 0; r < RT; ++r) off_last_prev[r][0] = off_out[L - 1][r][0], off_last_prev[r][1] = off_out[L - 1][r][1];
  }
  // the last chunk's epilogue
  {
    unsigned mw = 0;
    ro_for<0, 4 * NQ>([&](auto u_) __attribute__((always_inline)) { epi_word(ro_ic<NC - 1>{}, u_); });
    ro_for<0, NQ>([&](auto t_) __attribute__((always_inline)) {
      if constexpr (BWD) epi_colsum(ro_ic<NC - 1>{}, t_);
      epi_store(ro_ic<NC - 1>{}, t_, off_last_prev, mw, true);
    });
    mask_store(ro_ic<NC - 1>{}, mtile_prev, mw, true);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // copies still on their way into this workgroup's LDS
}

// ------------------------------------------------------------------------------------------------ weights -> fragment order
// chunk ct, k-step ks, lane (rho = lane & 31, g = lane >> 5), j = 0..7:
//   forward : W[col(ct, rho)][16 ks + 8 g + j]       (output = W's rows, contraction over W's columns)
//   backward: W[16 ks + 8 g + j][col(ct, rho)]       (output = W's columns, contraction over W's rows)
// zero outside the logical (out_f, in_f) matrix; bias -> fp32, zero padded
struct RoColsum {      // partial column sums for the caller to fold: out[i][0..n) = sum over nparts slices of part[i]
  const float* part[RO_MAXL + 1];
  float* out[RO_MAXL + 1];
  int n[RO_MAXL + 1];
  int count, nparts;
};

struct RoPackJob {
  const bf16_t* W;
  const bf16_t* b;
  bf16_t* Wf;
  float* bf;
  int out_f, in_f, CT, KS, bias_n;
};
struct RoPackArgs {
  RoPackJob job[RO_MAXL];
  int transpose;
};
template <class Cfg, bool BWD, int IN_COLS, int RT>
inline int ro_launch(const RoArgs& a, hipStream_t s) {
  static bool attr = false;
  const size_t lds = (size_t)RO_SLOTS * Cfg::max_ks() * 1024 + 8 * (RO_MAXCT - ro_nreg(BWD, RT)) * 2048 + (BWD ? (8 / RT) * (1024 + 4096) : Cfg::NB * 4);
  static_assert(RO_SLOTS * Cfg::max_ks() * 1024 + 8 * (RO_MAXCT - ro_nreg(BWD, RT)) * 2048 + (BWD ? (8 / RT) * (1024 + 4096) : Cfg::NB * 4) <= 160 * 1024, "LDS");
  if (!attr) {
    if (hipFuncSetAttribute((const void*)mlp_ro_kernel<Cfg, BWD, IN_COLS, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return check_launch("mlp_ro: LDS attribute");
    attr = true;
  }
  const int64_t ntiles = (a.rows + RO_ROWS - 1) / RO_ROWS;
  hipLaunchKernelGGL((mlp_ro_kernel<Cfg, BWD, IN_COLS, RT>), dim3((int)std::min<int64_t>(ntiles, 256)), dim3(512 / RT), lds, s, a);
  return check_launch(BWD ? "mlp_ro_bwd" : "mlp_ro_fwd");
}

// the stacks of the models: the per-field MLP of DeepAndCrossNetwork (64 -> 400 x 3 -> 64 on B*N rows) and the layers
// behind the first one of a 400-400-400 deep branch (DeepFM / xDeepFM: 416 -> 400 -> 400 -> 8)
#ifndef TRS_RO_BWD_RT
#define TRS_RO_BWD_RT 2
#endif
constexpr int RO_BWD_RT = TRS_RO_BWD_RT;      // the backward kernels: 64 rows per wave (with 32 they do not fit 256 registers)
using RoDcn = RoCfg<64, 400, 400, 400, 64>;
using RoTail = RoCfg<416, 400, 400, 8>;
using RoTailB = RoCfg<8, 400, 400, 416>;      // the same stack walked backwards


// one launcher per instantiated kernel, each in a file of its own (a kernel takes minutes to compile)
int ro_launch_dcn_fwd(const RoArgs& a, hipStream_t s);
int ro_launch_dcn_bwd(const RoArgs& a, hipStream_t s);
int ro_launch_tail_fwd(const RoArgs& a, hipStream_t s);
int ro_launch_tail_bwd(const RoArgs& a, hipStream_t s);

}  // namespace trs
