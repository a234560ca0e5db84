// Host side of the row-owner fused MLP kernels (mlp_ro.hpp): weights into fragment order, argument blocks, dispatch to
// the instantiated stack shapes (mlp_ro_dcn_*.hip, mlp_ro_tail_*.hip).
#include "mlp_ro.hpp"

namespace trs {

__global__ __launch_bounds__(256) void mlp_ro_prepack_kernel(RoPackArgs a) {
  const RoPackJob& j = a.job[blockIdx.y];
  const int total = j.CT * j.KS * 64;
  const bool vec = !a.transpose && (j.in_f & 7) == 0 && (reinterpret_cast<uintptr_t>(j.W) & 15u) == 0;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int lane = t & 63, f = t >> 6;
    const int ks = f % j.KS, ct = f / j.KS;
    const int oc = ro_col_of_row(ct, lane & 31);
    const int k0 = 16 * ks + 8 * (lane >> 5);
    uint4* dst = reinterpret_cast<uint4*>(j.Wf + (size_t)t * 8);
    if (vec) {
      *dst = (oc < j.out_f && k0 + 8 <= j.in_f) ? *reinterpret_cast<const uint4*>(j.W + (size_t)oc * j.in_f + k0)
                                                 : make_uint4(0, 0, 0, 0);
      continue;
    }
    uint16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + e;
      v[e] = 0;
      if (!a.transpose) {
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

static inline int ro_pad32(int x) { return (x + 31) / 32 * 32; }

template <class Cfg>
static bool ro_matches(int L, const int32_t* w, bool reversed) {
  if (L != Cfg::L) return false;
  for (int i = 0; i <= L; ++i)
    if (w[reversed ? L - i : i] != Cfg::w(i)) return false;
  return true;
}
static inline int ro_ks(int w) { return w <= 16 ? 2 : (w + 15) / 16; }

// Which kernel family runs a stack is a PER-CALL argument of the two entry points (trs_mlp_fused_family resolves a
// request; the backward is told the family its forward ran -- the sign-bit layouts differ).  There is no mutable state:
// the only process-wide input is the policy of TRS_MLP_FAMILY_AUTO, read once from the environment when the library is
// loaded -- TRS_MLP_RO = 0: never the row-owner kernels; 1 (default): for the shapes they are built for from RO_MIN_ROWS
// rows on (below that a workgroup sees one or two passes and the pipeline between passes -- next input, previous
// epilogue -- has nothing to run against: DeepFM's tail at B = 65 536, one pass per CU, forward 74 vs 101 us but
// backward 125 vs 92 us); 2: for those shapes at any size.
static const int RO_AUTO_POLICY = [] {
  const char* e = getenv("TRS_MLP_RO");
  return e != nullptr && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1;
}();
constexpr int64_t RO_MIN_ROWS = 131072;
constexpr int64_t RO_MIXED_MIN_ROWS = 32768;
static const bool RO_MIXED = [] {      // TRS_MLP_MIXED=0: AUTO never picks the mixed family
  const char* e = getenv("TRS_MLP_MIXED");
  return !(e != nullptr && e[0] == '0');
}();

bool mlp_ro_shape_ok(int L, const int32_t* widths, int64_t rows) {
  if (rows * 1024 >= ((int64_t)1 << 32)) return false;
  return ro_matches<RoDcn>(L, widths, false) || ro_matches<RoTail>(L, widths, false);
}

// request (TRS_MLP_FAMILY_*) -> the family that runs (TILE / ROW_OWNER), 0 when the request cannot be met
int mlp_resolve_family(int L, const int32_t* widths, int64_t rows, int request) {
  const bool ok = mlp_ro_shape_ok(L, widths, rows);
  switch (request) {
    case TRS_MLP_FAMILY_TILE: return TRS_MLP_FAMILY_TILE;
    case TRS_MLP_FAMILY_ROW_OWNER: return ok ? TRS_MLP_FAMILY_ROW_OWNER : 0;
    case TRS_MLP_FAMILY_MIXED: return ok ? TRS_MLP_FAMILY_MIXED : 0;
    case TRS_MLP_FAMILY_AUTO:
      if (ok && (RO_AUTO_POLICY == 2 || (RO_AUTO_POLICY == 1 && rows >= RO_MIN_ROWS))) return TRS_MLP_FAMILY_ROW_OWNER;
      // below that: the row-owner FORWARD (one pass per CU at 65 536 rows: 74 us against the tile kernel's 94-101) with the
      // tile BACKWARD (92 against 119-125), which reads the row-owner sign-bit layout -- from RO_MIXED_MIN_ROWS rows on
      // (fewer rows do not fill the 256-row passes of 256 workgroups)
      if (ok && RO_AUTO_POLICY == 1 && RO_MIXED && rows >= RO_MIXED_MIN_ROWS) return TRS_MLP_FAMILY_MIXED;
      return TRS_MLP_FAMILY_TILE;
    default: return 0;
  }
}

size_t mlp_ro_mask_bytes(int64_t rows) { return (size_t)((rows + RO_ROWS - 1) / RO_ROWS) * RO_MASK_WORDS * 4; }

// same contract as trs_mlp_fused_fwd (mlp_fused.hip), which hands over after its argument checks
int mlp_ro_fwd(const void* x, int64_t rows, int L, const int32_t* widths, const void* const* weights,
               const void* const* biases, void* const* hidden, void* const* masks, void* mask_in, void* y, void* workspace,
               hipStream_t s, int phase, int x_stride) {
  const bool pack_only = phase == TRS_MLP_PHASE_PACK;
  RoArgs a;
  a.in = (const char*)x;
  a.in_stride = x_stride;      // (both instantiated stacks read every column of their input width: the stride is only a pitch)
  a.rows = rows;
  a.mask_in = (uint32_t*)mask_in;
  a.colsum_in = nullptr;
  char* wsp = (char*)workspace;
  size_t wbytes = 0;
  for (int l = 0; l < L; ++l) wbytes += (size_t)ro_pad32(widths[l]) * ro_pad32(widths[l + 1]) * 2;
  wbytes = (wbytes + 255) / 256 * 256;
  float* bias_base = (float*)(wsp + wbytes);
  a.bias = bias_base;
  RoPackArgs pk;
  pk.transpose = 0;
  int pk_blocks = 1;
  size_t woff = 0, boff = 0;
  for (int l = 0; l < L; ++l) {
    const int KS = ro_ks(widths[l]), N = ro_pad32(widths[l + 1]);
    pk.job[l] = RoPackJob{(const bf16_t*)weights[l], (const bf16_t*)biases[l], (bf16_t*)(wsp + woff), bias_base + boff,
                          widths[l + 1], widths[l], N / 32, KS, N};
    pk_blocks = std::max(pk_blocks, std::min(256, (N / 32 * KS * 64 + 255) / 256));
    RoLayer& ly = a.layer[l];
    ly.wf = wsp + woff;
    ly.out = pack_only ? nullptr : (char*)(l + 1 < L ? hidden[l] : y);
    ly.out_stride = l + 1 < L ? N : widths[L];
    ly.out_cols = l + 1 < L ? N : widths[L];
    ly.mask = (l + 1 < L && !pack_only) ? (uint32_t*)masks[l] : nullptr;
    ly.colsum = nullptr;
    woff += (size_t)(N / 32) * KS * 1024;
    boff += N;
  }
  if (phase != TRS_MLP_PHASE_RUN) hipLaunchKernelGGL(mlp_ro_prepack_kernel, dim3(pk_blocks, L), dim3(256), 0, s, pk);
  if (pack_only) return check_launch("mlp_ro_fwd(pack)");
  return ro_matches<RoDcn>(L, widths, false) ? ro_launch_dcn_fwd(a, s) : ro_launch_tail_fwd(a, s);
}


// same contract as trs_mlp_fused_bwd_data (mlp_fused.hip), which hands over after its argument checks and folds the
// partial column sums listed in ``cs`` afterwards (its reduction kernel)
int mlp_ro_bwd(const void* gy, int64_t rows, int L, const int32_t* widths, const void* const* weights,
               const void* const* masks, void* const* gz, float* const* gbias, void* gx, const void* mask_in,
               float* gbias_in, void* workspace, hipStream_t s, RoColsum* cs, int phase) {
  const bool pack_only = phase == TRS_MLP_PHASE_PACK;
  RoArgs a;
  a.in = (const char*)gy;
  a.in_stride = widths[L];
  a.rows = rows;
  a.mask_in = nullptr;
  a.bias = nullptr;
  const int64_t ntiles = (rows + RO_ROWS - 1) / RO_ROWS;
  const int grid = (int)std::min<int64_t>(ntiles, 256);
  const int nparts = grid * (8 / RO_BWD_RT);
  char* wsp = (char*)workspace;
  size_t wbytes = 0;
  for (int l = 0; l < L; ++l) wbytes += (size_t)ro_pad32(widths[l]) * ro_pad32(widths[l + 1]) * 2;
  wbytes = (wbytes + 255) / 256 * 256;
  float* part_base = (float*)(wsp + wbytes);
  RoPackArgs pk;
  pk.transpose = 1;
  int pk_blocks = 1;
  size_t woff = 0, poff = 0;
  cs->count = 0;
  cs->nparts = nparts;
  // the input rows' column sums: the last layer's bias gradient
  a.colsum_in = part_base + poff;
  cs->part[cs->count] = a.colsum_in;
  cs->out[cs->count] = pack_only ? nullptr : gbias[L - 1];
  cs->n[cs->count++] = ro_pad32(widths[L]);
  poff += (size_t)nparts * ro_pad32(widths[L]);
  for (int sidx = 0; sidx < L; ++sidx) {
    const int l = L - 1 - sidx;
    const int KS = ro_ks(widths[l + 1]), N = ro_pad32(widths[l]);      // contraction over layer l's outputs, output = its inputs
    pk.job[sidx] = RoPackJob{(const bf16_t*)weights[l], nullptr, (bf16_t*)(wsp + woff), nullptr, widths[l + 1], widths[l], N / 32, KS, 0};
    pk_blocks = std::max(pk_blocks, std::min(256, (N / 32 * KS * 64 + 255) / 256));
    RoLayer& ly = a.layer[sidx];
    ly.wf = wsp + woff;
    ly.out = pack_only ? nullptr : (char*)(l > 0 ? gz[l - 1] : gx);
    ly.out_stride = l > 0 ? N : widths[0];
    ly.out_cols = l > 0 ? N : widths[0];
    ly.mask = pack_only ? nullptr : (uint32_t*)(l > 0 ? masks[l - 1] : mask_in);
    ly.colsum = nullptr;
    if (l > 0 || mask_in != nullptr || pack_only) {      // (a PACK call zeroes the input sums' slice whether or not it will be used)
      ly.colsum = part_base + poff;
      cs->part[cs->count] = ly.colsum;
      cs->out[cs->count] = pack_only ? nullptr : (l > 0 ? gbias[l - 1] : gbias_in);
      cs->n[cs->count++] = N;
      poff += (size_t)nparts * N;
    }
    woff += (size_t)(N / 32) * KS * 1024;
  }
  // (the partial column sums are accumulated by atomics: zeroed with the weights, i.e. in a PACK call when there is one)
  if (phase != TRS_MLP_PHASE_RUN) {
    if (int rc = zero_bytes(part_base, poff * 4, s)) return rc;
    hipLaunchKernelGGL(mlp_ro_prepack_kernel, dim3(pk_blocks, L), dim3(256), 0, s, pk);
  }
  if (pack_only) return check_launch("mlp_ro_bwd(pack)");
  return ro_matches<RoDcn>(L, widths, false) ? ro_launch_dcn_bwd(a, s) : ro_launch_tail_bwd(a, s);
}

}  // namespace trs
