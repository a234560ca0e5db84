// K4 on the matrix cores (bf16): x_{l+1} = x0 * (x_l W_l^T + b_l) + x0 for all L layers of a 16-row tile
// without leaving registers.
//
// Formulation: Y^T = W * X^T with v_mfma_f32_16x16x32_bf16:  A = W tile (M = e_out, K = e_in),
// B = X^T (K = e_in, N = 16 rows), D = Y^T (M = e_out, N = rows).  D's lane layout is
// (col = lane&15 -> row r of x, rows 4*(lane>>4)+i -> e_out slot), B's is (col = lane&15 -> r,
// k = 8*(lane>>4)+j).  The rows of W are fed in a permuted order
//     e_out(mt, q, i) = 32*(mt>>1) + 8*q + 4*(mt&1) + i          (mt = M-tile, q = lane>>4, i = reg)
// so that the 16 outputs a lane holds for its row r are exactly the 8-element chunks [32c+8q, 32c+8q+8)
// of that row: the NEXT layer's B operand is the lane's own registers (converted to bf16, no shuffle,
// no LDS), x0/out move as 16-byte vectors, and the elementwise epilogue x0*(u+b)+x0 is lane-local (the
// bias enters as the accumulator's initial value).  W_l fragments are pre-packed once per call into
// fragment order (1 KiB per (layer, mt, ks), lane-linear -> conflict-free ds_read_b128) and kept in LDS.
// HBM traffic: x read once, out written once (2*E*2 bytes per row); FLOPs 2*E*E*L per row on MFMA.
#include "trs_common.hpp"

namespace trs {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// e_out of D-row slot (mt, m) where m = 4*q + i is the row inside the 16-row M tile
__host__ __device__ __forceinline__ int cross_row_of_slot(int mt, int m) {
  return 32 * (mt >> 1) + 8 * (m >> 2) + 4 * (mt & 1) + (m & 3);
}

// Pre-pack: Wp[(l*NT + mt)*KS + ks][lane][8] = W_l[row(mt, lane&15)][32*ks + 8*(lane>>4) + 0..7]
// TRANSPOSE: pack W_l^T instead (used by the backward's data-gradient chain g_l = W_l^T du_l).
template <bool TRANSPOSE>
__global__ __launch_bounds__(256) void cross_prepack_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ b,
                                                            bf16_t* __restrict__ Wp, float* __restrict__ bp, int E,
                                                            int L) {
  const int NT = E / 16, KS = E / 32;
  const int total = L * NT * KS * 64;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int lane = t & 63;
    int f = t >> 6;
    const int ks = f % KS;
    f /= KS;
    const int mt = f % NT;
    const int l = f / NT;
    const int row = cross_row_of_slot(mt, lane & 15);
    const int k0 = 32 * ks + 8 * (lane >> 4);
    const bf16_t* Wl = W + (size_t)l * E * E;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      Wp[(size_t)t * 8 + j] = TRANSPOSE ? Wl[(size_t)(k0 + j) * E + row] : Wl[(size_t)row * E + k0 + j];
  }
  if (bp != nullptr)
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < L * E; t += gridDim.x * blockDim.x) bp[t] = to_f32(b[t]);
}

template <int NT>
struct XTile {
  float v[NT][4];  // D layout: v[mt][i] <-> e = 32*(mt>>1) + 8*q + 4*(mt&1) + i of row r = lane&15
};

template <int NT>
__device__ __forceinline__ void load_tile(const uint4* __restrict__ x, int64_t row, int64_t rows, int E, int q,
                                          XTile<NT>& t, uint4* raw) {
  constexpr int KS = NT / 2;
#pragma unroll
  for (int c = 0; c < KS; ++c) {
    uint4 u = make_uint4(0, 0, 0, 0);
    if (row < rows) u = x[(row * E + 32 * c + 8 * q) >> 3];
    raw[c] = u;
    float f[8];
    Vec16<bf16_t>::unpack(u, f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      t.v[2 * c][i] = f[i];
      t.v[2 * c + 1][i] = f[4 + i];
    }
  }
}

template <int NT>
__device__ __forceinline__ void pack_tile(const XTile<NT>& t, uint4* raw) {
  constexpr int KS = NT / 2;
#pragma unroll
  for (int c = 0; c < KS; ++c) {
    float f[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[i] = t.v[2 * c][i];
      f[4 + i] = t.v[2 * c + 1][i];
    }
    raw[c] = Vec16<bf16_t>::pack(f);
  }
}

// one layer for TP tiles at once: acc[t][mt] = bias + sum_ks A[mt][ks] * B[t][ks]; every A fragment read
// from LDS feeds TP MFMAs, and the TP independent chains let the MFMA of one tile overlap the VALU
// epilogue of the other inside a single wave.
template <int NT, int TP>
__device__ __forceinline__ void layer_matmul(const uint4* __restrict__ Wfrag /* [NT][KS][64] */,
                                             const float* __restrict__ bias /* [E] or null */,
                                             const uint4 (*B)[NT / 2], int lane, int q, f32x4 (*acc)[NT]) {
  constexpr int KS = NT / 2;
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
    f32x4 init = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + 32 * (mt >> 1) + 8 * q + 4 * (mt & 1));
      init = f32x4{bv.x, bv.y, bv.z, bv.w};
    }
#pragma unroll
    for (int t = 0; t < TP; ++t) acc[t][mt] = init;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint4 a = Wfrag[(mt * KS + ks) * 64 + lane];
#pragma unroll
      for (int t = 0; t < TP; ++t)
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                             __builtin_bit_cast(bf16x8, B[t][ks]), acc[t][mt], 0, 0, 0);
    }
  }
}

// RESIDENT: all layers' fragments + biases are copied into LDS once per workgroup; otherwise they are
// read from the pre-packed global buffer (L1/L2-resident) -- large E*E*L only.
template <int NT, bool RESIDENT>
__global__ __launch_bounds__(256) void cross_mfma_fwd_kernel(const uint4* __restrict__ x,
                                                             const uint4* __restrict__ Wp,
                                                             const float* __restrict__ bp, int64_t rows, int L,
                                                             uint4* __restrict__ out) {
  constexpr int KS = NT / 2;
  constexpr int E = NT * 16;
  constexpr int TP = NT <= 4 ? 2 : 1;           // tiles per wave iteration
  constexpr int FRAG_PER_LAYER = NT * KS * 64;  // uint4 units
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint4* Wl = Wp;
  const float* bl = bp;
  if (RESIDENT) {
    uint4* Ws = reinterpret_cast<uint4*>(smem);
    float* bs = reinterpret_cast<float*>(smem + (size_t)L * FRAG_PER_LAYER * 16);
    for (int i = threadIdx.x; i < L * FRAG_PER_LAYER; i += blockDim.x) Ws[i] = Wp[i];
    for (int i = threadIdx.x; i < L * E; i += blockDim.x) bs[i] = bp[i];
    __syncthreads();
    Wl = Ws;
    bl = bs;
  }
  const int lane = threadIdx.x & 63, q = lane >> 4, r = lane & 15;
  const int64_t ngroups = (rows + 16 * TP - 1) / (16 * TP);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t grp = wave; grp < ngroups; grp += nwaves) {
    XTile<NT> x0[TP], cur[TP];
    uint4 B[TP][KS];
    int64_t row[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      row[t] = (grp * TP + t) * 16 + r;
      load_tile<NT>(x, row[t], rows, E, q, x0[t], B[t]);
    }
    for (int l = 0; l < L; ++l) {
      f32x4 acc[TP][NT];
      layer_matmul<NT, TP>(Wl + (size_t)l * FRAG_PER_LAYER, bl + l * E, B, lane, q, acc);
#pragma unroll
      for (int t = 0; t < TP; ++t) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int i = 0; i < 4; ++i) cur[t].v[mt][i] = fmaf(x0[t].v[mt][i], acc[t][mt][i], x0[t].v[mt][i]);
        pack_tile<NT>(cur[t], B[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      if (row[t] < rows) {
#pragma unroll
        for (int c = 0; c < KS; ++c) out[(row[t] * E + 32 * c + 8 * q) >> 3] = B[t][c];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward on the matrix cores.  Workgroup = BW_WAVES waves, wave w owns the 16-row tile w of a row group.
// Per tile, in registers: recompute x_1..x_{L-1} (kept as bf16 B fragments), then for l = L-1..0
//   u_l  = W_l x_l + b_l (recomputed, MFMA)          du = g * x0          dx0 += g * (u_l + 1)
//   g    = W_l^T du      (MFMA on the transposed fragments; skipped for l = 0 when detach_first)
// dW_l = du^T x_l contracts over ROWS, which live on lanes in the layout above, so du and x_l are staged
// transposed in LDS ([e][rows of the group], bf16) once per layer and every wave accumulates its share of the E x E
// output tiles (K = rows of the group) in registers across the whole kernel; db_l rides along as
// one extra tile against an all-ones B operand.  Per-workgroup partial dW/db go to a workspace and are
// summed by a second kernel (no float atomics).
constexpr int BW_WAVES = 4;  // one wave per SIMD: the whole 512-register file per wave, no spills at L = 6
constexpr int BW_ROWS = 16 * BW_WAVES;          // rows per workgroup iteration
constexpr int BW_STR = BW_ROWS * 2 + 16;        // bytes per staged e-row (pad: conflict-free b128 reads)

template <int NT, int L>
__global__ __launch_bounds__(64 * BW_WAVES, 1) void cross_mfma_bwd_kernel(
    const uint4* __restrict__ x, const uint4* __restrict__ gout, const uint4* __restrict__ Wp,
    const uint4* __restrict__ WTp, const float* __restrict__ bp, int64_t rows, uint4* __restrict__ dx,
    float* __restrict__ dWpart, float* __restrict__ dbpart, int detach_first) {
  constexpr int KS = NT / 2;
  constexpr int E = NT * 16;
  constexpr int FRAG = NT * KS * 64;                       // uint4 per layer
  constexpr int OUT_TILES = NT * NT;                       // 16x16 tiles of dW_l
  constexpr int TPW = (OUT_TILES + BW_WAVES - 1) / BW_WAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* Ws = reinterpret_cast<uint4*>(smem);
  uint4* WTs = Ws + L * FRAG;
  float* bs = reinterpret_cast<float*>(WTs + L * FRAG);
  char* duT = reinterpret_cast<char*>(bs + L * E);
  char* xT = duT + E * BW_STR;
  for (int i = threadIdx.x; i < L * FRAG; i += blockDim.x) { Ws[i] = Wp[i]; WTs[i] = WTp[i]; }
  for (int i = threadIdx.x; i < L * E; i += blockDim.x) bs[i] = bp[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  f32x4 dWacc[L][TPW];
  f32x4 dbacc[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    dbacc[l] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TPW; ++t) dWacc[l][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);  // 8 x bf16(1.0)
  const int64_t ngroups = (rows + BW_ROWS - 1) / BW_ROWS;
  // software prefetch: the raw 16-byte vectors of the NEXT group's x / g tiles are loaded while this group
  // is being processed (one wave per SIMD: nothing else would hide the HBM latency)
  uint4 nx_raw[KS], ng_raw[KS];
  auto fetch = [&](int64_t grp_) {
    const int64_t row_ = grp_ * BW_ROWS + wave * 16 + r;
#pragma unroll
    for (int c = 0; c < KS; ++c) {
      nx_raw[c] = make_uint4(0, 0, 0, 0);
      ng_raw[c] = make_uint4(0, 0, 0, 0);
      if (grp_ < ngroups && row_ < rows) {
        nx_raw[c] = x[(row_ * E + 32 * c + 8 * q) >> 3];
        ng_raw[c] = gout[(row_ * E + 32 * c + 8 * q) >> 3];
      }
    }
  };
  fetch(blockIdx.x);
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t row = grp * BW_ROWS + wave * 16 + r;
    XTile<NT> g, dx0;
    uint4 Bx[L][KS];
#pragma unroll
    for (int c = 0; c < KS; ++c) {
      Bx[0][c] = nx_raw[c];
      float f[8];
      Vec16<bf16_t>::unpack(ng_raw[c], f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        g.v[2 * c][i] = f[i];
        g.v[2 * c + 1][i] = f[4 + i];
      }
    }
    fetch(grp + gridDim.x);
    // x0 is bf16 in memory: its fp32 values are re-derived from Bx[0] where needed (saves 16 registers)
    auto x0v = [&](int mt, int i) -> float {
      const uint4& u = Bx[0][mt >> 1];
      const unsigned w = (mt & 1) ? (i < 2 ? u.z : u.w) : (i < 2 ? u.x : u.y);
      return (i & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
    };
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) dx0.v[mt][i] = 0.f;
    // forward recompute of x_1 .. x_{L-1}
#pragma unroll
    for (int l = 0; l + 1 < L; ++l) {
      f32x4 acc[1][NT];
      layer_matmul<NT, 1>(Ws + l * FRAG, bs + l * E, &Bx[l], lane, q, acc);
      XTile<NT> nx;
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) nx.v[mt][i] = fmaf(x0v(mt, i), acc[0][mt][i], x0v(mt, i));
      pack_tile<NT>(nx, Bx[l + 1]);
    }
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
      f32x4 acc[1][NT];
      layer_matmul<NT, 1>(Ws + l * FRAG, bs + l * E, &Bx[l], lane, q, acc);   // u_l
      XTile<NT> du;
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          du.v[mt][i] = g.v[mt][i] * x0v(mt, i);
          dx0.v[mt][i] = fmaf(g.v[mt][i], acc[0][mt][i] + 1.f, dx0.v[mt][i]);
        }
      uint4 Bdu[KS];
      pack_tile<NT>(du, Bdu);
      // stage du^T and x_l^T as bf16 at [e][row].  Lane pairs (r, r^1) trade one dword so that every lane
      // writes a full dword = rows (r&~1, r|1) of ONE e (even lanes take element 2h, odd lanes 2h+1); the
      // byte column is XOR-swizzled with ((e>>3)&3)<<5 so the four q-groups of a wave hit different banks.
      {
        const int colpair = (wave * 16 + (r & ~1)) * 2;
        const int odd = r & 1;
#pragma unroll
        for (int c = 0; c < KS; ++c) {
          const unsigned wd[4] = {Bdu[c].x, Bdu[c].y, Bdu[c].z, Bdu[c].w};
          const unsigned wx[4] = {Bx[l][c].x, Bx[l][c].y, Bx[l][c].z, Bx[l][c].w};
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const unsigned od = __shfl_xor(wd[h], 1, 64), ox = __shfl_xor(wx[h], 1, 64);
            // even lane: (mine.lo, other.lo) -> element 2h ; odd lane: (other.hi, mine.hi) -> element 2h+1
            const unsigned vd = odd ? ((od >> 16) | (wd[h] & 0xffff0000u)) : ((wd[h] & 0xffffu) | (od << 16));
            const unsigned vx = odd ? ((ox >> 16) | (wx[h] & 0xffff0000u)) : ((wx[h] & 0xffffu) | (ox << 16));
            const int e = 32 * c + 8 * q + 2 * h + odd;
            const int off = e * BW_STR + (colpair ^ (q << 5));
            *reinterpret_cast<unsigned*>(duT + off) = vd;
            *reinterpret_cast<unsigned*>(xT + off) = vx;
          }
        }
      }
      // data-gradient chain (register-only: overlaps the staging barrier)
      XTile<NT> gn;
      const bool need_g = (l > 0) || (detach_first == 0);
      if (need_g) {
        f32x4 ga[1][NT];
        layer_matmul<NT, 1>(WTs + l * FRAG, nullptr, &Bdu, lane, q, ga);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int i = 0; i < 4; ++i) gn.v[mt][i] = ga[0][mt][i];
      }
      __syncthreads();
      // dW_l tiles: out tile t = wave*TPW + k -> (mo, no);  A = duT[16*mo + m][rows], B = xT[16*no + n][rows]
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        const int t = wave * TPW + k;
        if (t < OUT_TILES) {
          const int mo = t / NT, no = t - mo * NT;
#pragma unroll
          for (int kk = 0; kk < BW_ROWS / 32; ++kk) {
            const int sw = ((r >> 3) & 1) << 5;   // ((e>>3)&3)<<5 with e = 16*tile + r  ->  bit 3 of r (bit 4 of e is 0)
            const uint4 a = *reinterpret_cast<const uint4*>(duT + (16 * mo + r) * BW_STR +
                                                            (((32 * kk + 8 * q) * 2) ^ (sw | ((mo & 1) << 6))));
            const uint4 b = *reinterpret_cast<const uint4*>(xT + (16 * no + r) * BW_STR +
                                                            (((32 * kk + 8 * q) * 2) ^ (sw | ((no & 1) << 6))));
            dWacc[l][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                                  __builtin_bit_cast(bf16x8, b), dWacc[l][k], 0, 0, 0);
          }
        }
      }
      if (wave < NT) {   // db_l rows 16*wave .. +15: du^T x ones
#pragma unroll
        for (int kk = 0; kk < BW_ROWS / 32; ++kk) {
          const uint4 a = *reinterpret_cast<const uint4*>(
              duT + (16 * wave + r) * BW_STR + (((32 * kk + 8 * q) * 2) ^ ((((r >> 3) & 1) << 5) | ((wave & 1) << 6))));
          dbacc[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                             __builtin_bit_cast(bf16x8, ones), dbacc[l], 0, 0, 0);
        }
      }
      __syncthreads();
      if (need_g) g = gn;
    }
    if (row < rows) {
      XTile<NT> o;
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) o.v[mt][i] = dx0.v[mt][i] + (detach_first ? 0.f : g.v[mt][i]);
      uint4 raw[KS];
      pack_tile<NT>(o, raw);
#pragma unroll
      for (int c = 0; c < KS; ++c) dx[(row * E + 32 * c + 8 * q) >> 3] = raw[c];
    }
  }
  // partial results of this workgroup: D layout -> (row m = 4q+i, col n = r)
  float* myW = dWpart + (size_t)blockIdx.x * L * E * E;
  float* myb = dbpart + (size_t)blockIdx.x * L * E;
#pragma unroll
  for (int l = 0; l < L; ++l) {
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int t = wave * TPW + k;
      if (t < OUT_TILES) {
        const int mo = t / NT, no = t - mo * NT;
#pragma unroll
        for (int i = 0; i < 4; ++i) myW[(size_t)l * E * E + (16 * mo + 4 * q + i) * E + 16 * no + r] = dWacc[l][k][i];
      }
    }
    if (wave < NT && r == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) myb[l * E + 16 * wave + 4 * q + i] = dbacc[l][i];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward, role-specialised workgroup (the shipped path).
//
// The weight gradient dW_l = du_l^T x_l contracts over ROWS for every layer; its fp32 accumulators (E*E*L values:
// 384 registers per lane at E = 64, L = 6) are what forced one wave per SIMD, and in a symmetric design -- every
// wave owning a row tile AND a share of the accumulators -- du^T / x^T of all waves must be exchanged through LDS
// with two barriers per layer in the middle of every wave's dependent chain (the first version of this kernel: 2.1 ms
// at 2.5 M rows, half of its wave cycles spent waiting).  Here the two kinds of work get their own waves, 8 waves
// (two per SIMD, 256 registers each) per workgroup:
//   * waves 0..3 ("chain"): per step one layer of the gradient chain for one 16-row tile each, never touching dW.
//     They hand du_l and x_l to the other four waves as plain bf16 ROWS ([row][e], 16-byte stores of the registers
//     they already hold -- no shuffles, no transposing VALU work) in a double-buffered LDS slab;
//   * waves 4..7 ("dW"): each owns a quarter of the E*E*L accumulators (96 registers at E = 64, L = 6) and nothing
//     else; one step behind, they read the slab with ds_read_b64_tr_b16 -- the LDS transpose read delivers [row][e]
//     data with rows along K, i.e. directly as the A (du^T) and B (x) operands -- and issue independent MFMAs.
// One barrier per step.  Waves w and w+4 of a workgroup land on the same SIMD, so every SIMD hosts one chain wave
// (MFMA + VALU epilogues, 16 MFMAs per step) beside one dW wave (MFMA only, 8 per step): the two instruction mixes
// complement each other.  db_l = column sums of du_l is taken by the dW waves from the A operands they already hold
// (VALU adds beside the MFMAs) and reduced across lanes once at the end.
constexpr int B2_CHAIN = 4;                         // chain waves per workgroup (one 16-row tile each)
constexpr int B2_DW = 4;                            // weight-gradient waves
constexpr int B2_ROWS = B2_CHAIN * 16;              // rows per workgroup step (64)

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// 8 k-values (rows k0 .. k0+7 of the slab) x this lane's column: two transpose reads of a [4 rows][16 cols] block.
// Lane i of a 16-lane group supplies the address of 4 contiguous bf16 of row (i>>2), columns 4*(i&3)..+3, and
// receives column i of the block (element j = row j).
__device__ __forceinline__ s16x8 slab_frag(const char* slab, int str, int k0, int col0, int i) {
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const char* p = slab + (k0 + (i >> 2)) * str + (col0 + 4 * (i & 3)) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 4 * str));
  return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int NT, int L>
__global__ __launch_bounds__(64 * (B2_CHAIN + B2_DW), 2) void cross_mfma_bwd2_kernel(
    const uint4* __restrict__ x, const uint4* __restrict__ gout, const uint4* __restrict__ Wp,
    const uint4* __restrict__ WTp, const float* __restrict__ bp, int64_t rows, uint4* __restrict__ dx,
    float* __restrict__ dWpart, float* __restrict__ dbpart, int detach_first) {
  constexpr int KS = NT / 2;
  constexpr int E = NT * 16;
  constexpr int FRAG = NT * KS * 64;            // uint4 per layer
  constexpr int STR = E * 2 + 16;               // bytes per slab row (pad: conflict-free 16-byte row stores and tr reads)
  constexpr int SLAB = B2_ROWS * STR;           // one tensor of one buffer
  constexpr int TPW = NT * NT / B2_DW;          // dW output tiles per dW wave: tiles t = w', w'+4, ... -> (t / NT, t % NT)
  static_assert((NT * NT) % B2_DW == 0, "E must be a multiple of 32");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* Ws = reinterpret_cast<uint4*>(smem);
  uint4* WTs = Ws + L * FRAG;
  float* bs = reinterpret_cast<float*>(WTs + L * FRAG);
  char* stage = reinterpret_cast<char*>(bs + L * E);          // [buffer 2][du | x][B2_ROWS][STR]
  for (int i = threadIdx.x; i < L * FRAG; i += blockDim.x) { Ws[i] = Wp[i]; WTs[i] = WTp[i]; }
  for (int i = threadIdx.x; i < L * E; i += blockDim.x) bs[i] = bp[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  const int64_t ngroups = (rows + B2_ROWS - 1) / B2_ROWS;
  int step0 = 0;                                 // global step counter at the start of the group (buffer parity)

  if (wave < B2_CHAIN) {
    // ------------------------------------------------------------------ chain waves
    uint4 nx_raw[KS], ng_raw[KS];
    auto fetch = [&](int64_t grp_) {
      const int64_t row_ = grp_ * B2_ROWS + wave * 16 + r;
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        nx_raw[c] = make_uint4(0, 0, 0, 0);
        ng_raw[c] = make_uint4(0, 0, 0, 0);
        if (grp_ < ngroups && row_ < rows) {
          nx_raw[c] = x[(row_ * E + 32 * c + 8 * q) >> 3];
          ng_raw[c] = gout[(row_ * E + 32 * c + 8 * q) >> 3];
        }
      }
    };
    fetch(blockIdx.x);
    char* const myrow = stage + (wave * 16 + r) * STR + 16 * q;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x, step0 += L + 1) {
      XTile<NT> g, dx0;
      uint4 Bx[L][1][KS];
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        Bx[0][0][c] = nx_raw[c];
        float f[8];
        Vec16<bf16_t>::unpack(ng_raw[c], f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          g.v[2 * c][i] = f[i];
          g.v[2 * c + 1][i] = f[4 + i];
        }
      }
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) dx0.v[mt][i] = 0.f;
      fetch(grp + gridDim.x);
      // x0 is bf16 in memory: its fp32 values are re-derived from Bx[0] where needed
      auto x0v = [&](int mt, int i) -> float {
        const uint4& u = Bx[0][0][mt >> 1];
        const unsigned w = (mt & 1) ? (i < 2 ? u.z : u.w) : (i < 2 ? u.x : u.y);
        return (i & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
      };
      // step 0: forward recompute of x_1 .. x_{L-1} (bf16 B fragments, the values the forward kernel produced)
#pragma unroll
      for (int l = 0; l + 1 < L; ++l) {
        f32x4 acc[1][NT];
        layer_matmul<NT, 1>(Ws + l * FRAG, bs + l * E, Bx[l], lane, q, acc);
        XTile<NT> nx;
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int i = 0; i < 4; ++i) nx.v[mt][i] = fmaf(x0v(mt, i), acc[0][mt][i], x0v(mt, i));
        pack_tile<NT>(nx, Bx[l + 1][0]);
      }
      __syncthreads();
      // steps 1..L: layers L-1 .. 0
#pragma unroll
      for (int l = L - 1; l >= 0; --l) {
        char* rowp = myrow + (((step0 + (L - l)) & 1) ? 2 * SLAB : 0);
        f32x4 acc[1][NT];
        layer_matmul<NT, 1>(Ws + l * FRAG, bs + l * E, Bx[l], lane, q, acc);   // u_l
        XTile<NT> du;
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            du.v[mt][i] = g.v[mt][i] * x0v(mt, i);
            dx0.v[mt][i] = fmaf(g.v[mt][i], acc[0][mt][i] + 1.f, dx0.v[mt][i]);
          }
        uint4 Bdu[1][KS];
        pack_tile<NT>(du, Bdu[0]);
        // hand du_l and x_l to the dW waves as plain rows
#pragma unroll
        for (int c = 0; c < KS; ++c) {
          *reinterpret_cast<uint4*>(rowp + 64 * c) = Bdu[0][c];
          *reinterpret_cast<uint4*>(rowp + SLAB + 64 * c) = Bx[l][0][c];
        }
        const bool need_g = (l > 0) || (detach_first == 0);
        if (need_g) {
          f32x4 ga[1][NT];
          layer_matmul<NT, 1>(WTs + l * FRAG, nullptr, Bdu, lane, q, ga);
#pragma unroll
          for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) g.v[mt][i] = ga[0][mt][i];
        }
        __syncthreads();
      }
      const int64_t row = grp * B2_ROWS + wave * 16 + r;
      if (row < rows) {
        XTile<NT> o;
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int i = 0; i < 4; ++i) o.v[mt][i] = dx0.v[mt][i] + (detach_first ? 0.f : g.v[mt][i]);
        uint4 raw[KS];
        pack_tile<NT>(o, raw);
#pragma unroll
        for (int c = 0; c < KS; ++c) dx[(row * E + 32 * c + 8 * q) >> 3] = raw[c];
      }
    }
  } else {
    // ------------------------------------------------------------------ dW waves
    const int wq = wave - B2_CHAIN;
    f32x4 dWacc[L][TPW];
    float dbacc[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      dbacc[l] = 0.f;
#pragma unroll
      for (int k = 0; k < TPW; ++k) dWacc[l][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto consume = [&](const char* buf, f32x4 (&acc)[TPW], float& db) {
#pragma unroll
      for (int ks = 0; ks < B2_ROWS / 32; ++ks) {
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          const int t = wq + B2_DW * k, mo = t / NT, no = t % NT;
          const s16x8 A = slab_frag(buf, STR, 32 * ks + 8 * q, 16 * mo, r);
          const s16x8 Bf = slab_frag(buf + SLAB, STR, 32 * ks + 8 * q, 16 * no, r);
          acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, Bf),
                                                           acc[k], 0, 0, 0);
          if (no == mo % NT && (NT >= B2_DW ? mo == wq : true)) {
            // db: this lane's 8 rows of column e_out = 16*mo + r (every mo is taken by exactly one wave)
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += bf16_bits_to_f32((uint32_t)(uint16_t)A[j]);
            db += sum;
          }
        }
      }
    };
    bool first = true;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x, step0 += L + 1) {
      // step 0: layer 0 of the previous group (written in its last step)
      if (!first) consume(stage + (((step0 - 1) & 1) ? 2 * SLAB : 0), dWacc[0], dbacc[0]);
      first = false;
      __syncthreads();
#pragma unroll
      for (int l = L - 1; l >= 0; --l) {
        // step L-l: the chain waves are on layer l; these waves take layer l+1, written one step earlier
        if (l + 1 < L)
          consume(stage + (((step0 + (L - l) - 1) & 1) ? 2 * SLAB : 0), dWacc[l + 1 < L ? l + 1 : 0],
                  dbacc[l + 1 < L ? l + 1 : 0]);
        __syncthreads();
      }
    }
    if (!first) consume(stage + (((step0 - 1) & 1) ? 2 * SLAB : 0), dWacc[0], dbacc[0]);
    // partial results of this workgroup: D layout -> (row m = 4q+i -> e_out, col n = r -> e_in)
    float* myW = dWpart + (size_t)blockIdx.x * L * E * E;
    float* myb = dbpart + (size_t)blockIdx.x * L * E;
#pragma unroll
    for (int l = 0; l < L; ++l) {
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        const int t = wq + B2_DW * k, mo = t / NT, no = t % NT;
#pragma unroll
        for (int i = 0; i < 4; ++i) myW[(size_t)l * E * E + (16 * mo + 4 * q + i) * E + 16 * no + r] = dWacc[l][k][i];
        if (no == mo % NT && (NT >= B2_DW ? mo == wq : true)) {
          float v = dbacc[l];            // the four q-groups hold different rows of the same column
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          if (q == 0) myb[l * E + 16 * mo + r] = v;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void cross_reduce_partials_kernel(const float* __restrict__ part, int nparts,
                                                                    int n, float* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
    out[i] += s;
  }
}

constexpr int BW_MAX_BLOCKS = 256;

static size_t cross_pack_bytes(int E, int L) { return (size_t)L * E * E * 2; }

size_t cross_mfma_workspace_bytes(int E, int L) {
  // [Wp fwd][Wp transposed][bias fp32][per-workgroup dW partials][db partials], each 256-byte aligned
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  return 2 * al(cross_pack_bytes(E, L)) + al((size_t)L * E * 4) + al((size_t)BW_MAX_BLOCKS * L * E * E * 4) +
         al((size_t)BW_MAX_BLOCKS * L * E * 4);
}

static bool cross_mfma_covers(int E, int L) { return E % 32 == 0 && E >= 32 && E <= 128 && L >= 1; }

int cross_mfma_fwd(const void* x, const void* W, const void* b, int64_t rows, int E, int L, void* out,
                   void* workspace, size_t ws_bytes, hipStream_t s) {
  if (!cross_mfma_covers(E, L) || !aligned16(x) || !aligned16(out) || workspace == nullptr) return 1;
  if (ws_bytes < cross_mfma_workspace_bytes(E, L)) return fail(TRS_EWORKSPACE, "cross_fwd: workspace too small");
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  bf16_t* Wp = (bf16_t*)workspace;
  float* bp = (float*)((char*)workspace + 2 * al(cross_pack_bytes(E, L)));
  hipLaunchKernelGGL((cross_prepack_kernel<false>), dim3(std::min(64, (L * E * E / 8 + 255) / 256)), dim3(256), 0, s,
                     (const bf16_t*)W, (const bf16_t*)b, Wp, bp, E, L);
  const size_t lds = cross_pack_bytes(E, L) + (size_t)L * E * 4;
  const bool resident = lds <= 64 * 1024;
  const int64_t ntiles = (rows + 15) / 16;
  const int grid = (int)std::min<int64_t>((ntiles + 7) / 8, resident ? 256 * 3 : 256 * 8);
#define TRS_CF(NT_)                                                                                              \
  do {                                                                                                           \
    if (resident)                                                                                                \
      hipLaunchKernelGGL((cross_mfma_fwd_kernel<NT_, true>), dim3(grid), dim3(256), lds, s, (const uint4*)x,     \
                         (const uint4*)Wp, bp, rows, L, (uint4*)out);                                            \
    else                                                                                                         \
      hipLaunchKernelGGL((cross_mfma_fwd_kernel<NT_, false>), dim3(grid), dim3(256), 0, s, (const uint4*)x,      \
                         (const uint4*)Wp, bp, rows, L, (uint4*)out);                                            \
  } while (0)
  switch (E / 16) {
    case 2: TRS_CF(2); break;
    case 4: TRS_CF(4); break;
    case 6: TRS_CF(6); break;
    default: TRS_CF(8); break;
  }
#undef TRS_CF
  return check_launch("cross_fwd(mfma)");
}

template <int NT, int L>
static int cross_bwd_launch(const void* x, const void* g, const uint4* Wp, const uint4* WTp, const float* bp, int64_t rows,
                            void* dx, float* dWpart, float* dbpart, float* dW, float* db, int detach_first,
                            hipStream_t s) {
  constexpr int E = NT * 16;
  const size_t lds = (size_t)2 * L * E * E * 2 + (size_t)L * E * 4 + (size_t)4 * B2_ROWS * (E * 2 + 16);
  auto kern = cross_mfma_bwd2_kernel<NT, L>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return check_launch("cross_bwd(mfma): LDS attribute");
    attr_set = true;
  }
  const int64_t ngroups = (rows + B2_ROWS - 1) / B2_ROWS;
  const int grid = (int)std::min<int64_t>(ngroups, BW_MAX_BLOCKS);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (B2_CHAIN + B2_DW)), lds, s, (const uint4*)x, (const uint4*)g, Wp, WTp,
                     bp, rows, (uint4*)dx, dWpart, dbpart, detach_first);
  hipLaunchKernelGGL(cross_reduce_partials_kernel, dim3((L * E * E + 255) / 256), dim3(256), 0, s, dWpart, grid,
                     L * E * E, dW);
  hipLaunchKernelGGL(cross_reduce_partials_kernel, dim3((L * E + 255) / 256), dim3(256), 0, s, dbpart, grid, L * E, db);
  return check_launch("cross_bwd(mfma)");
}

int cross_mfma_bwd(const void* x, const void* W, const void* b, const void* g, int64_t rows, int E, int L, void* dx,
                   float* dW, float* db, int detach_first, void* workspace, size_t ws_bytes, hipStream_t s) {
  if (!(E == 32 || E == 64) || L < 1 || L > 6 || !aligned16(x) || !aligned16(g) || !aligned16(dx) ||
      workspace == nullptr)
    return 1;
  if (ws_bytes < cross_mfma_workspace_bytes(E, L)) return fail(TRS_EWORKSPACE, "cross_bwd: workspace too small");
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  char* ws = (char*)workspace;
  bf16_t* Wp = (bf16_t*)ws;
  bf16_t* WTp = (bf16_t*)(ws + al(cross_pack_bytes(E, L)));
  float* bp = (float*)(ws + 2 * al(cross_pack_bytes(E, L)));
  float* dWpart = (float*)((char*)bp + al((size_t)L * E * 4));
  float* dbpart = (float*)((char*)dWpart + al((size_t)BW_MAX_BLOCKS * L * E * E * 4));
  const int pgrid = std::min(64, (L * E * E / 8 + 255) / 256);
  hipLaunchKernelGGL((cross_prepack_kernel<false>), dim3(pgrid), dim3(256), 0, s, (const bf16_t*)W, (const bf16_t*)b, Wp,
                     bp, E, L);
  hipLaunchKernelGGL((cross_prepack_kernel<true>), dim3(pgrid), dim3(256), 0, s, (const bf16_t*)W, (const bf16_t*)b, WTp,
                     (float*)nullptr, E, L);
#define TRS_CB(NT_, L_)                                                                                              \
  return cross_bwd_launch<NT_, L_>(x, g, (const uint4*)Wp, (const uint4*)WTp, bp, rows, dx, dWpart, dbpart, dW, db,  \
                                   detach_first, s)
  if (E == 32) {
    switch (L) {
      case 1: TRS_CB(2, 1);
      case 2: TRS_CB(2, 2);
      case 3: TRS_CB(2, 3);
      case 4: TRS_CB(2, 4);
      case 5: TRS_CB(2, 5);
      default: TRS_CB(2, 6);
    }
  }
  switch (L) {
    case 1: TRS_CB(4, 1);
    case 2: TRS_CB(4, 2);
    case 3: TRS_CB(4, 3);
    case 4: TRS_CB(4, 4);
    case 5: TRS_CB(4, 5);
    default: TRS_CB(4, 6);
  }
#undef TRS_CB
}

}  // namespace trs
