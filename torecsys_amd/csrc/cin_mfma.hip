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
#include "trs_common.hpp"

namespace trs {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__host__ __device__ __forceinline__ int cin_chan_of_slot(int j, int ct2, int m) {
  return 32 * j + 8 * (m >> 2) + 4 * ct2 + (m & 3);
}

// Wp[((j*N + n)*2 + ct2)*KS + ks][lane][8] = Wc[chan(j,ct2,lane&15)][n*H + 32*ks + 8*(lane>>4) + 0..7] (0 past H)
__global__ __launch_bounds__(256) void cin_prepack_fwd_kernel(const bf16_t* __restrict__ Wc, bf16_t* __restrict__ Wp,
                                                              int C, int N, int H, int KS) {
  const int64_t total = (int64_t)(C / 32) * N * 2 * KS * 64;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(t & 63);
    int64_t f = t >> 6;
    const int ks = (int)(f % KS); f /= KS;
    const int ct2 = (int)(f & 1); f >>= 1;
    const int n = (int)(f % N);
    const int j = (int)(f / N);
    const int c = cin_chan_of_slot(j, ct2, lane & 15);
    const int h0 = 32 * ks + 8 * (lane >> 4);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int h = h0 + jj;
      Wp[t * 8 + jj] = h < H ? Wc[(size_t)c * N * H + (size_t)n * H + h] : bf16_t{0};
    }
  }
}

template <int KS, int P>
__global__ __launch_bounds__(256) void cin_cl_fwd_kernel(const bf16_t* __restrict__ x0T, int ld0,
                                                         const bf16_t* __restrict__ xkT, int ldk,
                                                         const uint4* __restrict__ Wp, const float* __restrict__ bias,
                                                         bf16_t* __restrict__ yT, int64_t B, int N, int C, int E) {
  constexpr int PIX = 16 * P;
  constexpr int FR = 2 * KS * 64;                      // uint4 per (j,n) step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* Abuf = reinterpret_cast<uint4*>(smem);                          // [2][FR]
  unsigned short* x0s = reinterpret_cast<unsigned short*>(smem + 2 * FR * 16);  // [4 waves][N][PIX]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  unsigned short* x0w = x0s + (size_t)wave * N * PIX;
  const int items_per_b = E / PIX;
  const int64_t nitems = B * items_per_b;
  const int nj = C / 32;
  const int nsteps = nj * N;
  for (int64_t it0 = (int64_t)blockIdx.x * 4; it0 < nitems; it0 += (int64_t)gridDim.x * 4) {
    const int64_t it = it0 + wave;
    const bool live = it < nitems;
    const int64_t b = live ? it / items_per_b : 0;
    const int e0 = live ? (int)(it - b * items_per_b) * PIX : 0;
    const int64_t pix0 = b * E + e0;                                     // first pixel row of this wave
    // B operands: xk^T rows of my pixels
    uint4 Bf[P][KS];
#pragma unroll
    for (int t = 0; t < P; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Bf[t][ks] = make_uint4(0, 0, 0, 0);
        if (live) Bf[t][ks] = *reinterpret_cast<const uint4*>(xkT + (pix0 + 16 * t + r) * ldk + 32 * ks + 8 * q);
      }
    __syncthreads();   // previous item's readers of x0s / Abuf are done
    // x0 of my pixels -> LDS [n][pixel] (bf16)
    if (live) {
      for (int v = lane; v < PIX * ((N + 7) / 8); v += 64) {
        const int p = v % PIX, ch = v / PIX;
        const uint4 u = *reinterpret_cast<const uint4*>(x0T + (pix0 + p) * ld0 + 8 * ch);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int n = 8 * ch + jj;
          if (n < N) x0w[n * PIX + p] = (unsigned short)(jj & 1 ? w[jj >> 1] >> 16 : w[jj >> 1] & 0xffffu);
        }
      }
    }
    // stage step 0
    for (int i = threadIdx.x; i < FR; i += 256) Abuf[i] = Wp[i];
    __syncthreads();
    f32x4 acc[P][2];
    for (int step = 0; step < nsteps; ++step) {
      const int j = step / N, n = step - j * N;
      const uint4* A = Abuf + (step & 1) * FR;
      // prefetch the next step's fragments into registers
      uint4 nxt[(FR + 255) / 256];
      if (step + 1 < nsteps) {
#pragma unroll
        for (int k = 0; k < (FR + 255) / 256; ++k) {
          const int i = threadIdx.x + 256 * k;
          if (i < FR) nxt[k] = Wp[(size_t)(step + 1) * FR + i];
        }
      }
      if (n == 0) {
#pragma unroll
        for (int ct2 = 0; ct2 < 2; ++ct2) {
          const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + 32 * j + 8 * q + 4 * ct2)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int t = 0; t < P; ++t) acc[t][ct2] = f32x4{bv.x, bv.y, bv.z, bv.w};
        }
      }
      f32x4 T[P][2];
#pragma unroll
      for (int t = 0; t < P; ++t) { T[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; T[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ct2 = 0; ct2 < 2; ++ct2)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint4 a = A[(ct2 * KS + ks) * 64 + lane];
#pragma unroll
          for (int t = 0; t < P; ++t)
            T[t][ct2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                                __builtin_bit_cast(bf16x8, Bf[t][ks]), T[t][ct2], 0, 0, 0);
        }
#pragma unroll
      for (int t = 0; t < P; ++t) {
        const float xv = __uint_as_float((unsigned)x0w[n * PIX + 16 * t + r] << 16);
#pragma unroll
        for (int ct2 = 0; ct2 < 2; ++ct2)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t][ct2][i] = fmaf(xv, T[t][ct2][i], acc[t][ct2][i]);
      }
      if (n == N - 1 && live) {
#pragma unroll
        for (int t = 0; t < P; ++t) {
          float f[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) { f[i] = acc[t][0][i]; f[4 + i] = acc[t][1][i]; }
          *reinterpret_cast<uint4*>(yT + (pix0 + 16 * t + r) * (int64_t)C + 32 * j + 8 * q) = Vec16<bf16_t>::pack(f);
        }
      }
      if (step + 1 < nsteps) {
        uint4* Anext = Abuf + ((step + 1) & 1) * FR;
#pragma unroll
        for (int k = 0; k < (FR + 255) / 256; ++k) {
          const int i = threadIdx.x + 256 * k;
          if (i < FR) Anext[i] = nxt[k];
        }
      }
      __syncthreads();
    }
  }
}

size_t cin_mfma_fwd_workspace_bytes(int N, int H, int C) {
  const int KS = (H + 31) / 32;
  return (size_t)(C / 32) * N * 2 * KS * 64 * 16 + (size_t)C * 4 + 512;
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
               int H, int C, int E, void* yT, void* workspace, size_t ws_bytes, hipStream_t s) {
  if (!cin_cl_covers(N, H, C, E)) return 1;
  const int KS = (H + 31) / 32;
  if (ld0 % 8 != 0 || ld0 < ((N + 7) / 8) * 8 || ldk % 8 != 0 || ldk < 32 * KS || !aligned16(x0T) || !aligned16(xkT) ||
      !aligned16(yT) || workspace == nullptr)
    return 1;
  if (ws_bytes < cin_mfma_fwd_workspace_bytes(N, H, C)) return fail(TRS_EWORKSPACE, "cin_cl_fwd: workspace too small");
  bf16_t* Wp = (bf16_t*)workspace;
  float* bf = (float*)((char*)workspace + (size_t)(C / 32) * N * 2 * KS * 64 * 16);
  const int64_t total = (int64_t)(C / 32) * N * 2 * KS * 64;
  hipLaunchKernelGGL(cin_prepack_fwd_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, s,
                     (const bf16_t*)Wc, Wp, C, N, H, KS);
  if (bias) hipLaunchKernelGGL(cin_bias_to_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const bf16_t*)bias, bf, C);
  const int P = E % 64 == 0 ? 4 : (E % 32 == 0 ? 2 : 1);
  const int64_t nitems = B * (E / (16 * P));
  const int grid = (int)std::min<int64_t>((nitems + 3) / 4, 256 * 2);
  const size_t lds = (size_t)2 * 2 * KS * 64 * 16 + (size_t)4 * N * 16 * P * 2;
  if (lds > 64 * 1024) return 1;
#define TRS_CINF(KS_, P_)                                                                                          \
  hipLaunchKernelGGL((cin_cl_fwd_kernel<KS_, P_>), dim3(grid), dim3(256), lds, s, (const bf16_t*)x0T, ld0,          \
                     (const bf16_t*)xkT, ldk, (const uint4*)Wp, bias ? bf : (const float*)nullptr, (bf16_t*)yT, B, N, \
                     C, E)
#define TRS_CINF_P(KS_)              \
  do {                               \
    if (P == 4) TRS_CINF(KS_, 4);    \
    else if (P == 2) TRS_CINF(KS_, 2); \
    else TRS_CINF(KS_, 1);           \
  } while (0)
  switch (KS) {
    case 1: TRS_CINF_P(1); break;
    case 2: TRS_CINF_P(2); break;
    case 4: TRS_CINF_P(4); break;
    default: TRS_CINF_P(8); break;
  }
#undef TRS_CINF_P
#undef TRS_CINF
  return check_launch("cin_cl_fwd");
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
                              void* yT, void* workspace, size_t ws_bytes, trs_stream_t stream) {
  if (B == 0) return TRS_OK;
  TRS_REQUIRE(x0T && xkT && Wc && yT, TRS_EINVAL, "cin_cl_fwd: NULL pointer");
  TRS_REQUIRE(B > 0 && N > 0 && H > 0 && C > 0 && E > 0, TRS_EINVAL, "cin_cl_fwd: bad size");
  TRS_REQUIRE(dtype == TRS_BF16, TRS_EDTYPE, "cin_cl_fwd: bf16 only (dtype %d)", dtype);
  const int rc = cin_cl_fwd(x0T, ld0, xkT, ldk, Wc, bias, B, N, H, C, E, yT, workspace, ws_bytes, (hipStream_t)stream);
  if (rc == 1) return fail(TRS_ESHAPE, "cin_cl_fwd: shape not covered (need C%%32==0, E%%16==0, H<=256, padded rows)");
  return rc;
}
