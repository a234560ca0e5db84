/*
 * trs_abi.h -- C ABI of libtrs_hip.so, the MI355X (gfx950) kernels behind the torecsys
 * embedding-lookup + feature-interaction hot path.
 *
 * The reference (p768lwy3/torecsys) is pure Python on PyTorch: it has no FFI of its own, the
 * "plugin interface" for this path is nn.Module.forward().  Each entry point below therefore
 * cites the reference forward() (file:line, relative to the reference root) whose ATen ops it
 * replaces.  The Python host side (torecsys_amd/) mirrors those nn.Modules and calls these
 * symbols through ctypes; INTEGRATION.md shows the binding a maintainer adds on the reference side.
 *
 * Conventions
 *  - plain C: POD arguments only -- device pointers, sizes, dtype codes, a hipStream_t as void*;
 *  - the CALLER owns every buffer (inputs, outputs, workspaces); the library never allocates,
 *    frees or keeps a pointer after the call; all tensors are dense, row-major, contiguous;
 *  - every call only ENQUEUES work on the caller's stream (no hidden synchronisation);
 *  - return 0 (TRS_OK) or a negative TRS_E* code; trs_last_error_string() (thread-local) explains;
 *  - no C++ exception crosses this boundary; the library is stateless and re-entrant;
 *  - dtype codes: TRS_F32 / TRS_BF16 for values, TRS_I64 / TRS_I32 for indices;
 *  - "offsets" is the per-field row offset vector (N int64, device) added to the raw indices,
 *    or NULL for none; "err_flag" (device int32, may be NULL) is set to 1 when an index falls
 *    outside [0,V) -- the row is then skipped (zeros on a gather) instead of read out of bounds.
 */
#ifndef TRS_ABI_H_
#define TRS_ABI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRS_ABI_VERSION 1

enum { TRS_F32 = 0, TRS_BF16 = 1 };
enum { TRS_I64 = 0, TRS_I32 = 1 };
enum {
  TRS_OK = 0,
  TRS_EINVAL = -1,     /* NULL pointer / negative size / bad flag */
  TRS_EDTYPE = -2,     /* unsupported dtype code */
  TRS_ESHAPE = -3,     /* unsupported shape (e.g. N or E beyond a kernel's limits) */
  TRS_EALIGN = -4,     /* pointer not aligned for the vector path */
  TRS_ELAUNCH = -5,    /* hip launch error (hipGetLastError) */
  TRS_EWORKSPACE = -6  /* workspace too small */
};

typedef void* trs_stream_t; /* hipStream_t */

int trs_version(void);
const char* trs_last_error_string(void);

/* ---- diagnostics: device-side timestamps that survive hipGraph replay -----------------------
 * Enqueues a one-lane kernel that appends the device wall clock (constant rate, trs_wall_clock_khz())
 * to a ring in device memory:  i = ring[0]++;  ring[1 + i % capacity] = wall_clock64().
 * Two marks bracketing a launch on the same stream measure it like a HIP event pair does, but each
 * replay of a captured graph appends a new sample (a captured event pair only keeps the last one).
 * ring: capacity+1 uint64, zero-initialised by the caller.  No reference counterpart (measurement only). */
int trs_mark_timestamp(uint64_t* ring, int32_t capacity, trs_stream_t stream);
int64_t trs_wall_clock_khz(void);

/* ---- K1: row gather ---------------------------------------------------------------------
 * out[b,n,:] = table[idx[b,n] + offsets[n], :]            (bit-exact copy)
 * replaces aten::add + aten::embedding in
 *   torecsys/inputs/base/multi_indices_emb.py:104-105, single_index_emb.py:56-57.            */
int trs_gather_rows(const void* table, int64_t V, int32_t E, int32_t dtype,
                    const void* idx, int32_t idx_dtype, const int64_t* offsets,
                    int64_t B, int32_t N, void* out, int32_t* err_flag, trs_stream_t stream);

/* ---- N separate tables, one index column each: a StackedInput of SingleIndexEmbeddings in one launch -------------
 * inputs/base/stacked_inp.py:94-134 calls N SingleIndexEmbeddings (single_index_emb.py:48-59: nn.Embedding on a (B,1)
 * column) and concatenates along N.  out[b,n,:] = tables[n][idx[b,n],:], bit-exact.  tables: DEVICE array of N base
 * pointers (rows of E elements, 16-byte aligned when E*sizeof(T) is a multiple of 16); table_rows: DEVICE array of N row
 * counts; idx (B,N) int64/int32 RAW per-table indices (no offsets).  An index outside its table reads as a zero row and
 * raises *err_flag.  Backward: the caller buckets idx over the concatenated row space (trs_csr_build with offsets =
 * cumulative row counts) and scatters into one (sum rows, E) gradient whose slices are the tables' gradients.          */
int trs_gather_rows_tables(const void* const* tables, const int64_t* table_rows, int32_t E, int32_t dtype, const void* idx,
                           int32_t idx_dtype, int64_t B, int32_t N, void* out, int32_t* err_flag, trs_stream_t stream);

/* ---- I3: field-aware gather ---------------------------------------------------------------
 * out[b, i*N+j, :] = tables[i][idx[b,j] + offsets[j], :]   for i,j in [0,N)
 * `tables` = device array of N table base pointers (each V x E).
 * replaces the N lookups + cat of multi_indices_field_aware_emb.py:102-105.                   */
int trs_fa_gather_rows(const void* const* tables, int64_t V, int32_t E, int32_t dtype,
                       const void* idx, int32_t idx_dtype, const int64_t* offsets,
                       int64_t B, int32_t N, void* out, int32_t* err_flag, trs_stream_t stream);

/* ---- row-bucketed index (CSR) ---------------------------------------------------------------
 * Groups the B*N lookups by destination row: perm[row_start[r] .. row_start[r+1]) lists the
 * flat positions p = b*N+n with idx[p]+offsets[p%N] == r.  Built once per batch, shared by every
 * table looked up with the same indices.  row_start: V+1 int32; perm: B*N int32.
 * Replaces the index_add loop of aten::embedding_dense_backward (reached from
 * multi_indices_emb.py:105 through autograd) with an atomics-free segmented reduction.       */
size_t trs_csr_workspace_bytes(int64_t V, int64_t BN);
int trs_csr_build(const void* idx, int32_t idx_dtype, const int64_t* offsets, int64_t B, int32_t N,
                  int64_t V, int32_t* row_start, int32_t* perm, void* workspace, size_t ws_bytes,
                  int32_t* err_flag, trs_stream_t stream);

/* ---- K1 backward: dense-gradient scatter ----------------------------------------------------
 * grad_table[r,:] = sum_{p in row r} ( g_rows[p,:]  +  g_fm[b_p,:] * (fm_sum[b_p,:] - table[r,:]) )
 * for EVERY r in [0,V) (rows without lookups are written as zeros: nn.Embedding's default
 * sparse=False gradient, multi_indices_emb.py:48).  g_rows (B*N x E, may be NULL) is the
 * gradient of the gathered block; the second term (g_fm B x E, fm_sum B x E fp32, table; all
 * NULL to disable) is the FM second-order backward dx = g*(S - x) of
 * layers/ctr/factorization_machine.py:62-73 fused into the same pass.  padding_row (-1: none)
 * gets a zero gradient (nn.Embedding padding_idx).  fp32 accumulation, one rounding on store.
 * g_fm_cols = E: g_fm holds full (B x E) rows; g_fm_cols = 1 (with fm_sum): g_fm holds ONE value per sample, the
 * gradient is constant along E (the backward of a sum over E -- what the reference's FM / DeepFM models feed back,
 * models/ctr/deep_fm.py:55-110): the walk then reads 144 instead of 256 bytes of FM operands per lookup.
 * With fm_sum == NULL, g_fm is a plain per-sample gradient broadcast over the N fields (the
 * first-order sum's backward; g_fm_cols = E).  g_rows_batch_stride (rows; 0 = N) lets g_rows be a (B, N, E) slice
 * of a larger (B, M, E) tensor: the row of position (b,n) is b*stride + n.
 * Rows with more than 256 lookups (Zipf-hot rows) are queued in `workspace`
 * (trs_scatter_workspace_bytes) and reduced by whole workgroups; the workspace also holds the
 * per-sample [g*S | g] rows the FM term is read from.                         */
size_t trs_scatter_workspace_bytes(int64_t BN, int32_t N, int32_t E, int32_t dtype);
int trs_scatter_rows(const void* g_rows, int64_t g_rows_batch_stride, const void* g_fm, int32_t g_fm_cols,
                     const float* fm_sum, const void* table, const int32_t* row_start,
                     const int32_t* perm, int64_t BN, int64_t V, int32_t E, int32_t N, int32_t dtype,
                     int64_t padding_row, void* grad_table, void* workspace, size_t ws_bytes,
                     trs_stream_t stream);

/* trs_scatter_rows plus the dense gradient of the companion first-order table (V x 1) of the same lookups in the same
 * bucket walk: grad_first[r] = sum over the lookups (b,n) of row r of g_first[b,n]  (g_first: (B,N), one value per
 * lookup; every row of grad_first is written).  Rows must be whole 16-byte vectors (E*sizeof(dtype) % 16 == 0).
 * Replaces the second embedding backward of the (B,N,1) first-order lookup (torch embedding_dense_backward of
 * multi_indices_emb.py:104-105 at embed_size = 1).                                                               */
int trs_scatter_rows_first(const void* g_rows, int64_t g_rows_batch_stride, const void* g_fm, int32_t g_fm_cols,
                           const float* fm_sum,
                           const void* table, const int32_t* row_start, const int32_t* perm, int64_t BN, int64_t V,
                           int32_t E, int32_t N, int32_t dtype, int64_t padding_row, void* grad_table,
                           const void* g_first, void* grad_first, void* workspace, size_t ws_bytes,
                           trs_stream_t stream);

/* Same walk, but the finished row sum is APPLIED to the table row in place by a fused sparse optimizer
 *  instead of being written out: optimizer 1 = SGD  w -= lr*g;
 * 2 = Adagrad  state += g*g, w -= lr*g/(sqrt(state)+eps)  (state: V x E fp32).  Exactly equivalent to the
 * dense torch.optim.SGD / Adagrad step (no momentum / weight decay): rows nobody looked up have zero gradient
 * and are not touched -- neither the dense V x E gradient nor a dense optimizer pass over the table exists.  */
int trs_scatter_rows_update(const void* g_rows, int64_t g_rows_batch_stride, const void* g_fm, int32_t g_fm_cols,
                            const float* fm_sum, void* table, const int32_t* row_start, const int32_t* perm,
                            int64_t BN, int64_t V, int32_t E, int32_t N, int32_t dtype, int64_t padding_row,
                            int32_t optimizer, float lr, float eps, float* state, void* workspace,
                            size_t ws_bytes, trs_stream_t stream);

/* Same pass with a lazy Adam step (torch.optim.SparseAdam semantics: moments and weights of looked-up rows only;
 * the reference trains with dense Adam, trainer/torecsys_pipeline.py:562-578, which cannot exist at 1 B rows):
 *   m = m + (g - m)(1 - beta1);  v = v + (g*g - v)(1 - beta2);  w -= step_size * m / (sqrt(v) + eps)
 * with step_size = lr * sqrt(1 - beta2^t) / (1 - beta1^t) computed by the caller; exp_avg / exp_avg_sq: V x E fp32. */
int trs_scatter_rows_update_adam(const void* g_rows, int64_t g_rows_batch_stride, const void* g_fm, int32_t g_fm_cols,
                                 const float* fm_sum, void* table, const int32_t* row_start, const int32_t* perm,
                                 int64_t BN, int64_t V, int32_t E, int32_t N, int32_t dtype, int64_t padding_row,
                                 float step_size, float beta1, float beta2, float eps, float* exp_avg,
                                 float* exp_avg_sq, void* workspace, size_t ws_bytes, trs_stream_t stream);

/* The same fused optimizer step when the bucketed rows are a COMPACT list of U distinct table rows (the owner side of
 * a row-sharded table: a 125 M-row shard cannot afford a V-sized bucket index per step):  row_start (U+1) / perm (K)
 * bucket the K gradient rows g_rows (K,E) by compact row u, row_map[u] is the table row that compact row u updates
 * (distinct values).  optimizer 1 = SGD, 2 = Adagrad (state), 3 = lazy Adam (lr = bias-corrected step size, state =
 * exp_avg, state2 = exp_avg_sq); state buffers are V x E fp32.  Workspace: trs_scatter_workspace_bytes(K, 1, E, dtype). */
int trs_scatter_rows_update_mapped(const void* g_rows, void* table, const int32_t* row_map, const int32_t* row_start,
                                   const int32_t* perm, int64_t K, int64_t U, int64_t V, int32_t E, int32_t dtype,
                                   int32_t optimizer, float lr, float eps, float beta1, float beta2, float* state,
                                   float* state2, void* workspace, size_t ws_bytes, trs_stream_t stream);

/* ---- K1+K2(+K8): fused embedding lookup + FM second order ----------------------------------
 * emb[b,n,:]  = table[idx[b,n]+offsets[n], :]                       (optional, may be NULL)
 * fm[b,:]     = 0.5 * ((sum_n x)^2 - sum_n x^2)                     (optional, may be NULL)
 * fm_sum[b,:] = sum_n x   (fp32, optional; saved for the backward)
 * first[b]    = sum_n first_table[idx[b,n]+offsets[n]]              (optional E=1 first-order term)
 * replaces multi_indices_emb.py:104-105 + factorization_machine.py:62-73
 * (+ models/ctr/deep_fm.py:73-89 first-order sum) in one pass over the table rows.           */
int trs_embed_fm(const void* table, int64_t V, int32_t E, int32_t dtype,
                 const void* idx, int32_t idx_dtype, const int64_t* offsets, int64_t B, int32_t N,
                 void* emb, void* fm, float* fm_sum, const void* first_table, void* first,
                 int32_t* err_flag, trs_stream_t stream);

/* The same pass when the first-order term stays a per-field tensor, as the reference's models consume it
 * (``feat_inputs`` (B,N,1) of models/ctr/deep_fm.py:73-89, factorization_machine.py model, xdeep_fm.py):
 * first_vals[b,n] = first_table[idx[b,n]+offsets[n]]  (0 for an out-of-range lookup) -- replaces the separate
 * MultiIndicesEmbedding(embed_size=1) lookup (multi_indices_emb.py:104-105) of the same index block.            */
int trs_embed_fm_fields(const void* table, int64_t V, int32_t E, int32_t dtype,
                        const void* idx, int32_t idx_dtype, const int64_t* offsets, int64_t B, int32_t N,
                        void* emb, void* fm, float* fm_sum, const void* first_table, void* first_vals,
                        int32_t* err_flag, trs_stream_t stream);

/* ---- K2: FM layer on a materialised block -------------------------------------------------
 * fwd: fm[b,:] = 0.5*((sum_n x)^2 - sum_n x^2), fm_sum[b,:] = sum_n x (fp32, optional)
 * bwd: dx[b,n,:] = g[b,:] * (fm_sum[b,:] - x[b,n,:])
 * layers/ctr/factorization_machine.py:62-73.                                                  */
int trs_fm_fwd(const void* x, int64_t B, int32_t N, int32_t E, int32_t dtype, void* fm, float* fm_sum,
               trs_stream_t stream);
int trs_fm_bwd(const void* x, const void* g, const float* fm_sum, int64_t B, int32_t N, int32_t E,
               int32_t dtype, void* dx, trs_stream_t stream);

/* ---- K7: inner-product network ------------------------------------------------------------
 * fwd: out[b,p(i,j)] = sum_e x[b,i,e]*x[b,j,e], i<j lexicographic, P = N(N-1)/2
 * bwd: dx[b,i,:] = sum_{j!=i} g[b,p(min,max)] * x[b,j,:]
 * layers/ctr/inner_product_network.py:68-74.                                                  */
int trs_pair_dot_fwd(const void* x, int64_t B, int32_t N, int32_t E, int32_t dtype, void* out,
                     trs_stream_t stream);
int trs_pair_dot_bwd(const void* x, const void* g, int64_t B, int32_t N, int32_t E, int32_t dtype,
                     void* dx, trs_stream_t stream);

/* K7 fused with K1: out[b,p(i,j)] = <row(b,i), row(b,j)> with row(b,n) = table[idx[b,n]+offsets[n], :] read straight from
 * the embedding table (one wave per sample; an out-of-range id reads as a zero row and sets *err_flag); emb (optional,
 * may be NULL): the looked-up (B,N,E) block, written on the way for the backward / other consumers.  bf16 tables whose
 * rows the matrix-core path covers (E % 32 == 0, E <= 128, N <= 64); TRS_ESHAPE otherwise (callers then use
 * trs_gather_rows + trs_pair_dot_fwd).  Replaces multi_indices_emb.py:104-105 + inner_product_network.py:68-74.   */
int trs_embed_pair_dot(const void* table, int64_t V, int32_t E, int32_t dtype, const void* idx, int32_t idx_dtype,
                       const int64_t* offsets, int64_t B, int32_t N, void* emb, void* out, int32_t* err_flag,
                       trs_stream_t stream);

/* ---- K3: field-aware FM pair products -----------------------------------------------------
 * fwd: out[b,p(i,j),:] = x[b,i*N+j,:] * x[b,j*N+i,:], i<j      x: (B, N*N, E)
 * bwd: dx[b,i*N+j,:] = g[b,p,:]*x[b,j*N+i,:]; dx[b,j*N+i,:] = g[b,p,:]*x[b,i*N+j,:]; diagonal 0
 * layers/ctr/field_aware_factorization_machine.py:69-87.
 * trs_ffm_fused_fwd gathers straight from the N tables (never materialises (B,N*N,E)):
 *   out[b,p(i,j),:] = tables[i][g_j,:] * tables[j][g_i,:],  g_n = idx[b,n]+offsets[n].       */
int trs_ffm_fwd(const void* x, int64_t B, int32_t N, int32_t E, int32_t dtype, void* out,
                trs_stream_t stream);
int trs_ffm_bwd(const void* x, const void* g, int64_t B, int32_t N, int32_t E, int32_t dtype,
                void* dx, trs_stream_t stream);
int trs_ffm_fused_fwd(const void* const* tables, int64_t V, int32_t E, int32_t dtype,
                      const void* idx, int32_t idx_dtype, const int64_t* offsets, int64_t B, int32_t N,
                      void* out, int32_t* err_flag, trs_stream_t stream);
/* dense gradients of the N tables of trs_ffm_fused_fwd (row_start/perm from trs_csr_build on the same
 * indices): grad_tables[i][r,:] = sum_{(b,j) in row r, j != i} gout[b,p(i,j),:] * tables[j][g_i,:].
 * row_ids_t (optional, may be NULL): the global row ids g_n = idx[b,n]+offsets[n] TRANSPOSED, (N, B) int32 -- the walk
 * of table i then finds the g_i of its lookups in one 4 B-per-sample column (256 KB at B = 65 536: cache-resident)
 * instead of one random 8-byte load per lookup from the (B, N) matrix.                                          */
int trs_ffm_fused_bwd(const void* const* tables, int64_t V, int32_t E, int32_t dtype, const void* idx,
                      int32_t idx_dtype, const int64_t* offsets, const void* gout,
                      const int32_t* row_start, const int32_t* perm, int64_t B, int32_t N,
                      void* const* grad_tables, const int32_t* row_ids_t, trs_stream_t stream);

/* ---- K4: cross network ----------------------------------------------------------------------
 * x_{l+1} = x0 * (x_l W_l^T + b_l) + x0, l = 0..L-1, rows = B*N vectors of length E.
 * W: (L,E,E) row-major [l][out][in] as nn.Linear.weight; b: (L,E).
 * bwd with detach_first=1 reproduces cross_network.py:65 (x_0 enters layer 0's linear map
 * detached); detach_first=0 gives the textbook gradient.
 *   dx (rows,E), dW (L,E,E) fp32, db (L,E) fp32 (dW/db are ACCUMULATED into: zero them first).
 * workspace (trs_cross_workspace_bytes; may be 0 / NULL for fp32) holds the W fragments re-packed for
 * the MFMA path (bf16, E % 32 == 0, E <= 128); without it the generic VALU kernels run.
 * layers/ctr/cross_network.py:65-79.                                                          */
size_t trs_cross_workspace_bytes(int64_t rows, int32_t E, int32_t L, int32_t dtype);
int trs_cross_fwd(const void* x, const void* W, const void* b, int64_t rows, int32_t E, int32_t L,
                  int32_t dtype, void* out, void* workspace, size_t ws_bytes, trs_stream_t stream);
int trs_cross_bwd(const void* x, const void* W, const void* b, const void* g, int64_t rows, int32_t E,
                  int32_t L, int32_t dtype, int32_t detach_first, void* dx, float* dW, float* db,
                  void* workspace, size_t ws_bytes, trs_stream_t stream);

/* ---- K5/K6: compress interaction network, one layer ------------------------------------------
 * y[b,c,e] = bias[c] + sum_{n,h} Wc[c, n*H+h] * x0[b,n,e] * xk[b,h,e]
 *   x0: (B,N,E) the embedding block; xk: (B,H,E) previous hidden ((B,N,E) for layer 0);
 *   Wc: (C, N*H) = nn.Conv1d(N*H, C, 1).weight squeezed; bias (C) or NULL; y: (B,C,E).
 * The outer product Z[b,(n,h),e] is formed on the fly (never written to memory).
 * stats (2*C fp32, optional): per-channel sum and sum of squares of y over (B,E), accumulated
 * (zero first) -- the BatchNorm1d batch statistics.
 * bwd: given gy (B,C,E): dWc (C,N*H) fp32 accumulated, dx0 (B,N,E) and dxk (B,H,E) written.
 * layers/ctr/compress_interaction_network.py:125-137.                                          */
int trs_cin_fwd(const void* x0, const void* xk, const void* Wc, const void* bias, int64_t B, int32_t N,
                int32_t H, int32_t C, int32_t E, int32_t dtype, void* y, float* stats,
                trs_stream_t stream);
int trs_cin_bwd(const void* x0, const void* xk, const void* Wc, const void* gy, int64_t B, int32_t N,
                int32_t H, int32_t C, int32_t E, int32_t dtype, float* dWc, void* dx0, void* dxk,
                int32_t accumulate_dx0, trs_stream_t stream);

/* ---- K5 on the matrix cores: channels-last CIN contraction (bf16) ------------------------------
 * Same contraction as trs_cin_fwd with activations stored channels-last:
 *   x0T (B,E,ld0): x0T[b,e,n] = x0[b,n,e], row stride ld0 (multiple of 8, zero-filled past N)
 *   xkT rows = B*E pixels of stride ldk (multiple of 8, >= 32*ceil(H/32), zero-filled past H):
 *       xkT[b*E+e, h] = xk[b,h,e]   (a strided view of the previous layer's yT is fine)
 *   yT  (B,E,C): yT[b,e,c] = y[b,c,e]
 * Requirements: C % 32 == 0, E % 16 == 0, H <= 256.  workspace: trs_cin_cl_workspace_bytes.
 * The x0[n] factor multiplies the MFMA result, so the outer product never exists even in registers.
 * tri != 0 (needs xkT == x0T, ldk == ld0, N == H): the caller states that Wc[c, n*H + h] == 0 for every h > n -- the form the first
 * layer's weights take once W[n,h] + W[h,n] is folded onto h <= n, which is exact there because xk IS x0 and
 * the products x0[n]*x0[h] are symmetric -- and the k-steps that only hold such zeros are skipped.
 * layers/ctr/compress_interaction_network.py:125-137.                                          */
size_t trs_cin_cl_workspace_bytes(int32_t N, int32_t H, int32_t C);
int trs_cin_cl_fwd(const void* x0T, int32_t ld0, const void* xkT, int32_t ldk, const void* Wc,
                   const void* bias, int64_t B, int32_t N, int32_t H, int32_t C, int32_t E, int32_t dtype,
                   int32_t tri, void* yT, void* workspace, size_t ws_bytes, trs_stream_t stream);
/* data gradients of the same contraction, channels-last: given gyT (B,E,C)
 *   dx0T (B,E,ld0)  : dx0T[b,e,n] = sum_{c,h} gy*Wc*xk   (zeros past N)
 *   dxkT rows of stride ldo (>= 32*ceil(H/32)): dxkT[b*E+e,h] = sum_{c,n} gy*Wc*x0
 * Requirements: C in {32,64,128,256}, E % 16 == 0.  bias gradient = sum of gy over (b,e) (caller).
 * tri: as in trs_cin_cl_fwd; xkT is x0T there, so the two gradients belong to the same values: dx0T receives
 * their SUM (added in fp32, rounded once) and dxkT is not written (may be NULL).                        */
size_t trs_cin_cl_bwd_data_workspace_bytes(int32_t N, int32_t H, int32_t C);
int trs_cin_cl_bwd_data(const void* x0T, int32_t ld0, const void* xkT, int32_t ldk, const void* gyT,
                        const void* Wc, int64_t B, int32_t N, int32_t H, int32_t C, int32_t E,
                        int32_t dtype, int32_t tri, void* dx0T, void* dxkT, int32_t ldo, void* workspace,
                        size_t ws_bytes, trs_stream_t stream);

/* weight gradient of the same contraction on the matrix cores; CHANNELS-FIRST operands (the sum runs over
 * pixels, which must be the contiguous axis): gy (B,C,E), x0 (B,N,E), xk (B,H,E) bf16;
 * dW (C, N*H) fp32 is ACCUMULATED into (zero it first).  C in {64,128,256}, E in {32,64,128}.
 * tri != 0 (needs xk == x0, N == H): the first layer, where dW[c,n,h] is symmetric in (n,h) -- only the
 * 16-h blocks at or below the diagonal are computed and the rest is mirrored; the result is the same full
 * (C, N*H) gradient of the UNFOLDED weights.                                                          */
size_t trs_cin_dw_workspace_bytes(int64_t B, int32_t N, int32_t H, int32_t C);
int trs_cin_dw(const void* gy, const void* x0, const void* xk, int64_t B, int32_t N, int32_t H, int32_t C,
               int32_t E, int32_t dtype, int32_t tri, float* dW, void* workspace, size_t ws_bytes,
               trs_stream_t stream);

/* The same two gradients for a layer whose output channels beyond the first C receive NO gradient: the reference splits
 * EVERY CIN layer's 2H output channels into a "direct" and a "hidden" half (compress_interaction_network.py:151-156: the
 * `i != len-1` guard is always true) and never uses the LAST layer's hidden half (:176-181 reads the direct halves only),
 * so dL/dy of those channels is exactly zero -- through BatchNorm and the activation too, which act per channel.  Their
 * terms of the contraction over c (data gradients) and their rows of dW are zeros that need not be computed:
 *   gyT (B,E,>=C) with ldg channels between two pixels, Wc = the first C rows of the layer's weight; results identical
 *   to trs_cin_cl_bwd_data on the full tensors (the omitted terms are exact zeros);
 *   gy (B,>=C,E) with gy_batch_stride elements between two samples; dW (C, N*H) = the first C rows of the gradient (the
 *   caller zero-fills the rest).                                                                                      */
int trs_cin_cl_bwd_data_live(const void* x0T, int32_t ld0, const void* xkT, int32_t ldk, const void* gyT, int32_t ldg,
                             const void* Wc, int64_t B, int32_t N, int32_t H, int32_t C, int32_t E, int32_t dtype,
                             void* dx0T, void* dxkT, int32_t ldo, void* workspace, size_t ws_bytes, trs_stream_t stream);
int trs_cin_dw_live(const void* gy, int64_t gy_batch_stride, const void* x0, const void* xk, int64_t B, int32_t N, int32_t H,
                    int32_t C, int32_t E, int32_t dtype, float* dW, void* workspace, size_t ws_bytes, trs_stream_t stream);

/* ---- MLP backward epilogue (the GEMMs stay on hipBLASLt) -----------------------------------------------
 * y = relu(linear(x)):  gz = gy * (y > 0) and gb[c] = sum_r gz[r,c] (fp32) in ONE pass over (rows, C) instead of
 * ATen's threshold_backward + column sum.  C*sizeof(T) must be a multiple of 16 and <= 4096.
 * layers/ctr/multilayer_perceptron.py:53-61 (Linear_i / Activation_i pairs).                                 */
size_t trs_relu_bwd_bias_workspace_bytes(int64_t rows, int32_t C);
int trs_relu_bwd_bias(const void* gy, const void* y, int64_t rows, int32_t C, int32_t dtype, void* gz,
                      float* gb, void* workspace, size_t ws_bytes, trs_stream_t stream);

/* ---- per-field MLP of DeepAndCrossNetwork as one kernel per direction -------------------------
 * models/ctr/deep_and_cross_network.py:71-87 applies MultilayerPerceptionLayer (layers/ctr/multilayer_perceptron.py:
 * 53-84: Linear -> ReLU ... -> Linear) to every (sample, field) row of the (B,N,E) block.  bf16 only.
 *   widths[0..num_layers]: in_features of layer 0, then out_features of every layer (multiples of 8, <= 512);
 *   weights[l]: (widths[l+1], widths[l]) row-major = nn.Linear.weight; biases[l]: (widths[l+1]); host arrays of
 *   device pointers.  ReLU after every layer but the last.
 * fwd: hidden[l] (rows, pad32(widths[l+1])) for l < num_layers-1 = the ReLU outputs, zero in the padding columns
 *      (kept for the weight-gradient GEMMs); masks[l]: trs_mlp_fused_mask_bytes(rows) bytes, the sign bits
 *      [hidden > 0] in the kernel's own (pass, lane) order -- opaque, read only by trs_mlp_fused_bwd_data on the same
 *      rows and widths; y (rows, widths[num_layers]).
 * bwd_data: gy (rows, widths[num_layers]) -> gz[l] (rows, pad32(widths[l+1])), l < num_layers-1 = gradient w.r.t. the
 *      pre-activation of layer l (that of the last layer is gy itself); gbias[l] (pad32(widths[l+1]) fp32, written) for
 *      every layer; gx (rows, widths[0]).  Weight gradients are left to the caller: dW_l = gz_l^T @ input_l.
 * A stack fed by the ReLU output of a layer in front of it (the hipBLASLt first layer of a deep branch): mask_in
 *      (trs_mlp_fused_mask_bytes(rows) bytes, may be NULL) receives the sign bits [x > 0] in the forward; given to the
 *      backward together with gbias_in (pad32(widths[0]) fp32, written), gx becomes the gradient w.r.t. that layer's
 *      pre-activation, gx * [x > 0], and gbias_in its column sums = that layer's bias gradient (needs num_layers <= 7).
 * trs_mlp_fused_supported: 1 when the widths fit the kernel (and its LDS budget).                              */
int trs_mlp_fused_supported(int32_t num_layers, const int32_t* widths);
/* Kernel families.  The two stack shapes of the models (64-400-400-400-64 on B*N rows; 416-400-400-8 behind a wide first
 * layer) have a second pair of kernels (csrc/mlp_ro.hpp: a wave owns 64 / 32 rows for the whole stack) behind the same
 * two entry points, and the two families lay the sign bits out differently.  Which family runs is a PER-CALL argument --
 * the library keeps no mode:
 *   trs_mlp_fused_fwd(family = AUTO | TILE | ROW_OWNER); AUTO = ROW_OWNER for the covered shapes from 131 072 rows on
 *   (the start-up environment may set TRS_MLP_RO=0: never / 2: at any size -- read once when the library is loaded);
 *   trs_mlp_fused_family(num_layers, widths, rows, request) -> the family (TILE or ROW_OWNER) that request runs, 0 when it
 *   cannot be met (ROW_OWNER on an uncovered shape; the forward then returns TRS_EINVAL).  A pure function.
 *   trs_mlp_fused_bwd_data(family) must be given THAT value (TILE, ROW_OWNER or MIXED; AUTO is TRS_EINVAL): the caller records
 *   what its forward ran, so a policy or size threshold cannot come between a forward and its backward.              */
#define TRS_MLP_FAMILY_AUTO 0
#define TRS_MLP_FAMILY_TILE 1
#define TRS_MLP_FAMILY_ROW_OWNER 2
#define TRS_MLP_FAMILY_MIXED 3      /* row-owner forward, tile backward reading the row-owner sign-bit layout (covered
                                       shapes below 131 072 rows: each direction on the family that is faster there) */
/* Phases.  Both entry points (and trs_rows_gemm) first copy the weights into MFMA fragment order in the workspace, then run.
 * The copy depends on the parameters only, so a caller may take it off the critical path of its step:
 *   phase = ALL : copy, then run (one call does everything);
 *   phase = PACK: only the copy (+ zeroing of the workspace's partial sums) -- enqueue it on ANY stream as soon as the
 *                 parameters are final, e.g. on a side stream at the start of the step; pointers to rows / outputs /
 *                 masks may be NULL, `rows`, `widths`, `weights`, `biases`, `family` and the workspace must be the ones
 *                 of the RUN call;
 *   phase = RUN : the workspace holds what a PACK call with the same arguments left (the caller orders the two calls:
 *                 same stream, or an event); nothing is copied.                                                       */
/* x_stride (trs_mlp_fused_fwd): elements between consecutive rows of x; 0 = widths[0].  A larger stride (a multiple of 8)
 * makes the stack read the first widths[0] columns of wider rows -- the 416-wide tail of a deep branch on the 512-wide
 * output of the library GEMM in front of it; row-owner and mixed family only (TRS_ESHAPE otherwise).  The matching
 * backward may then be called with widths[0] = that stride under TRS_MLP_FAMILY_MIXED: it computes the gradient of all
 * x_stride columns (those past the forward's width meet zero weights and come out as zeros).                       */
#define TRS_MLP_PHASE_ALL 0
#define TRS_MLP_PHASE_PACK 1
#define TRS_MLP_PHASE_RUN 2
int32_t trs_mlp_fused_family(int32_t num_layers, const int32_t* widths, int64_t rows, int32_t request);
size_t trs_mlp_fused_workspace_bytes(int32_t num_layers, const int32_t* widths);
size_t trs_mlp_fused_mask_bytes(int64_t rows);
int trs_mlp_fused_fwd(const void* x, int64_t rows, int32_t num_layers, const int32_t* widths,
                      const void* const* weights, const void* const* biases, void* const* hidden, void* const* masks,
                      void* mask_in, void* y, int32_t dtype, int32_t family, int32_t phase, int32_t x_stride,
                      void* workspace, size_t ws_bytes, trs_stream_t stream);
int trs_mlp_fused_bwd_data(const void* gy, int64_t rows, int32_t num_layers, const int32_t* widths,
                           const void* const* weights, const void* const* masks, void* const* gz, float* const* gbias,
                           void* gx, const void* mask_in, float* gbias_in, int32_t dtype, int32_t family, int32_t phase,
                           void* workspace, size_t ws_bytes, trs_stream_t stream);

/* Every weight copy a deep branch needs, in ONE launch: the PACK phases of trs_mlp_fused_fwd (into ws_fwd), of
 * trs_mlp_fused_bwd_data (into ws_bwd, same size; NULL: none) and of the trs_rows_gemm of the Linear in front of the stack
 * (gemm_W (gemm_out_f, gemm_in_f) into ws_gemm; NULL: none); the three RUN-phase calls then find what their own PACK call
 * would have left.  For stacks that `family` resolves to the tile kernels (TRS_ESHAPE otherwise: use the PACK phases).
 * multilayer_perceptron.py:53-61 -- three 6-9 us launches with 5 us gaps in front of a 65 536-row deep branch's kernels
 * become one that runs beside the first layer's GEMM.                                                                */
int trs_mlp_pack_branch(int64_t rows, int32_t num_layers, const int32_t* widths, const void* const* weights,
                        const void* const* biases, int32_t family, void* ws_fwd, void* ws_bwd, size_t ws_bytes,
                        const void* gemm_W, int32_t gemm_out_f, int32_t gemm_in_f, void* ws_gemm, size_t ws_gemm_bytes,
                        trs_stream_t stream);

/* dst (rows, c_out) = [src (rows, c_in) | zeros], bf16: the gradient of the first c_in columns of a padded output (the
 * logit column of a deep branch whose last layer runs at 8 columns) in one launch.                                   */
int trs_pad_cols(const void* src, int32_t c_in, void* dst, int32_t c_out, int64_t rows, int32_t dtype,
                 trs_stream_t stream);

/* ---- one wide layer with a short contraction: the input gradient of a deep branch's first Linear ----------------
 * y (rows, in_f) = x[:, :out_f] @ W, x (rows, x_stride) and W (out_f, in_f) = nn.Linear(in_f, out_f).weight, bf16:
 * dL/d(input) of that layer from dL/d(pre-activation) (multilayer_perceptron.py:53-61 under autograd).  out_f <= 512
 * (a 128-row tile of x stays in LDS), in_f % 8 == 0, x_stride % 8 == 0 and >= out_f rounded up to 32 (the columns
 * past out_f meet zero weight rows).  workspace: trs_rows_gemm_workspace_bytes (fragment-order copy of W); phase:
 * TRS_MLP_PHASE_* as above (PACK: x and y may be NULL).                                                              */
size_t trs_rows_gemm_workspace_bytes(int32_t out_f, int32_t in_f);
int trs_rows_gemm_supported(int32_t out_f, int32_t in_f, int32_t x_stride);
int trs_rows_gemm(const void* x, int64_t rows, int32_t x_stride, const void* W, int32_t out_f, int32_t in_f,
                  int32_t dtype, int32_t phase, void* y, void* workspace, size_t ws_bytes, trs_stream_t stream);

/* ---- row-sharded tables (multi-GPU lookup) -----------------------------
 * Bucket the B*N global row ids of the local batch by owner rank (owner = id / rows_per_rank):
 *   counts[w]  = number of ids owned by rank w                                  (W int64)
 *   send_ids   = LOCAL row ids (id - owner*rows_per_rank) grouped by owner      (B*N int32)
 *   send_pos   = for each grouped slot k the original flat position p = b*N+n   (B*N int32)
 *   inv_pos    = inverse permutation, inv_pos[p] = k (may be NULL)              (B*N int32)
 * Order inside one owner's group is unspecified.  workspace from trs_bucket_workspace_bytes. */
size_t trs_bucket_workspace_bytes(int64_t BN, int32_t world);
int trs_bucket_by_owner(const void* idx, int32_t idx_dtype, const int64_t* offsets, int64_t B, int32_t N,
                        int64_t rows_per_rank, int32_t world, int64_t* counts, int32_t* send_ids,
                        int32_t* send_pos, int32_t* inv_pos, void* workspace, size_t ws_bytes,
                        trs_stream_t stream);
/* Un-permute of the sharded lookup (step 5 of torecsys_amd/dist.py's forward) with the lookups this rank owns ITSELF read
 * straight from its shard: slot s = inv_pos[p]; s in [self_lo, self_lo + self_n) -> row local[send_ids[s]] (ids outside
 * [0, n_valid) read as zero rows and raise err_flag); every other slot -> back[s - (s >= self_lo + self_n ? self_n : 0)]
 * (the received rows are stored WITHOUT the self segment: back_rows of them).  Writes emb (B,N,E), and -- optional, as
 * trs_embed_fm -- the FM second-order term fm (B,E) and the fp32 field sums fm_sum (B,E).  New: the reference has no
 * distributed code. */
int trs_embed_fm_sharded(const void* back, int64_t back_rows, const void* local, int64_t local_rows, int64_t n_valid,
                         int32_t E, int32_t dtype, const int32_t* inv_pos, const int32_t* send_ids, int32_t self_lo,
                         int32_t self_n, int64_t B, int32_t N, void* emb, void* fm, float* fm_sum, int32_t* err_flag,
                         trs_stream_t stream);
/* out[pos[k],:] = rows[k,:]  (un-permute received rows into the (B*N,E) block) */
int trs_scatter_by_pos(const void* rows, const int32_t* pos, int64_t K, int32_t E, int32_t dtype,
                       void* out, trs_stream_t stream);
/* out[k,:] = g_block[pos[k],:] + g_fm[b,:]*(fm_sum[b,:] - x[pos[k],:]), b = pos[k]/N: the block gradient (and
 * the FM second-order backward when the FM term was fused into the sharded lookup) in exchange order, one pass.
 * g_block or the (g_fm, fm_sum, x) triple may be NULL.  pos[k] < 0 marks a padding slot of a fixed-capacity
 * exchange: its row is written as zeros.                                                                      */
int trs_permute_grad(const void* g_block, const void* g_fm, const float* fm_sum, const void* x,
                     const int32_t* pos, int64_t K, int32_t N, int32_t E, int32_t dtype, void* out,
                     trs_stream_t stream);
/* out[k,:] = rows[pos[k],:]  (permute the block gradient into exchange order) */
int trs_gather_by_pos(const void* rows, const int32_t* pos, int64_t K, int32_t E, int32_t dtype,
                      void* out, trs_stream_t stream);

/* ---- Linear with one output unit (the logit layer of the DeepFM / xDeepFM MLP, multilayer_perceptron.py:51) -------
 * out[r] = h[r,:] . w + bias[0]: a row-wise dot product, not a GEMM.  h (rows, C), w (C); C * sizeof(T) / 16 must be
 * a power of two <= 64.  bwd: gh (rows, C) = g[r] * w (may be NULL); gw (C), gb (1) fp32, written (both or neither). */
int trs_rowdot_fwd(const void* h, const void* w, const void* bias, int64_t rows, int32_t C, int32_t dtype, void* out,
                   trs_stream_t stream);
size_t trs_rowdot_bwd_workspace_bytes(int64_t rows, int32_t C);
int trs_rowdot_bwd(const void* g, const void* h, const void* w, int64_t rows, int32_t C, int32_t dtype, void* gh,
                   float* gw, float* gb, void* workspace, size_t ws_bytes, trs_stream_t stream);

/* ---- batched 2-D transposition with zero padding (the channels-last entry of the CIN layer) --------------------------
 * out[b][c][r] = in[b][r][c] (r < R, c < Cc), 0 for R <= r < ld_out.  in (B, R, ld_in), out (B, Cc, ld_out), bf16;
 * R, Cc, ld_in <= 64; ld_in, ld_out multiples of 8.  compress_interaction_network.py:105 (align_to('B','E','N')):
 * x0 (B,N,E) -> x0T (B,E,ld0) with the field axis padded to the matrix-core k-step, and its gradient back.          */
int trs_transpose_pad(const void* in, int64_t B, int32_t R, int32_t Cc, int32_t ld_in, void* out, int32_t ld_out,
                      int32_t dtype, trs_stream_t stream);

/* ---- one-output Linear over the concatenation of two blocks, without the concatenation -----------------------------
 * The head of deep_and_cross_network.py:82-92: torch.cat([cross_out, deep_out], dim='O') -> flatten(('N','O'),'O') ->
 * nn.Linear(cat_size, 1).  a (rows, N, Ea), d (rows, N, Eb) contiguous, w (N * (Ea + Eb)) in the Linear's own order
 * (field-major [a_f | d_f]), bias (1) or NULL:
 *   out[r] = sum_f  a[r,f,:] . w[f*(Ea+Eb) : +Ea]  +  d[r,f,:] . w[f*(Ea+Eb)+Ea : +Eb]  + bias[0]
 * bwd: ga = g[r] * (a's part of w), gd = g[r] * (d's part) (either may be NULL); gw (N*(Ea+Eb)) and gb (1) fp32,
 * written (both or neither).  Ea, Eb whole 16-byte vectors; N * (Ea + Eb) * sizeof(T) / 16 <= 1024.               */
int trs_cat_head_fwd(const void* a, const void* d, const void* w, const void* bias, int64_t rows, int32_t N, int32_t Ea,
                     int32_t Eb, int32_t dtype, void* out, trs_stream_t stream);
size_t trs_cat_head_bwd_workspace_bytes(int64_t rows, int32_t N, int32_t Ea, int32_t Eb);
int trs_cat_head_bwd(const void* g, const void* a, const void* d, const void* w, int64_t rows, int32_t N, int32_t Ea,
                     int32_t Eb, int32_t dtype, void* ga, void* gd, float* gw, float* gb, void* workspace,
                     size_t ws_bytes, trs_stream_t stream);

/* ---- finish of a split-K weight gradient (the K = batch GEMM of multilayer_perceptron.py's nn.Linear backward) -----
 * part (S, R, Cc) fp32 partial products  ->  gw (out_rows, out_cols) = sum_s part[s, :out_rows, :out_cols] cast to
 * dtype (the un-padded corner when the GEMMs ran on zero-padded weights); gb (out_rows) = cast(gb_f32) (both or
 * neither; needs out_rows <= 256 * ceil(out_cols / 256)).                                                          */
int trs_wgrad_finish(const float* part, int32_t S, int32_t R, int32_t Cc, int32_t out_rows, int32_t out_cols,
                     int32_t dtype, void* gw, const float* gb_f32, void* gb, trs_stream_t stream);
/* the same for partial products stored transposed, part (S, Cc, R) (slices of x^T g):
 * gw[r, c] = sum_s part[s, c, r] for r < out_rows <= R, c < out_cols <= Cc.                                          */
int trs_wgrad_finish_t(const float* part, int32_t S, int32_t Cc, int32_t R, int32_t out_rows, int32_t out_cols,
                       int32_t dtype, void* gw, const float* gb_f32, void* gb, trs_stream_t stream);
/* dW partials by a hand-written kernel instead of a batched library GEMM: part (S, M, N) fp32, slice s = g[rows_s, :M]^T
 * x[rows_s, :N] over the s-th of S contiguous row ranges (bf16 operands, row strides ldg / ldx, M and N multiples of
 * 8); trs_wgrad_finish then folds the slices.  S comes from trs_wgrad_rows_splits (0: shape not handled -- more than
 * 32 blocks of 224 x 224, or fewer than 256 rows); that number halved any number of times down to 8 is accepted too
 * (longer row ranges on fewer workgroups, for two weight gradients that share the chip on two streams).              */
int32_t trs_wgrad_rows_splits(int32_t M, int32_t N, int64_t rows);
int trs_wgrad_rows(const void* g, int32_t ldg, const void* x, int32_t ldx, int64_t rows, int32_t M, int32_t N,
                   int32_t dtype, int32_t S, float* part, trs_stream_t stream);
/* n strided 2-D copies in one launch: desc (device, n x 6 int64) = {src address, dst address, rows, cols, src_ld,
 * dst_ld} (sizes in elements of elem_size bytes); max_elems = the largest rows*cols (sizes the grid).  Refreshes the
 * zero-padded copies of an MLP stack's nn.Linear parameters (multilayer_perceptron.py:55-61) before a forward.     */
int trs_copy_padded_many(const int64_t* desc, int32_t n, int32_t elem_size, int64_t max_elems, trs_stream_t stream);

/* ---- CIN layer glue on channels-last activations y (B,E,C) bf16 ---------------------------------
 * BatchNorm1d + ReLU + chunk(2) + sum over E of the direct half, compress_interaction_network.py:137-181:
 *   z = relu(y * scale[c] + shift[c])      scale = gamma * invstd, shift = beta - mean * scale (fp32, caller)
 *   hidden[b,e,c-Hs] = z  for c >= Hs ;   pooled[b,c] = sum_e z[b,e,c]  for c < D
 *   (split layers: D = Hs = C/2;  is_direct layers: D = C, Hs = 0)
 * stats: partial (trs_cin_glue_blocks(B), 2, C) fp32 = per-workgroup column sums / sums of squares of y.
 * bwd_reduce: partial (blocks, 2, C) = [sum gz*mask, sum gz*mask*xhat] with gz = g_pooled (c < D) + g_hidden
 * (c >= Hs), mask = [z > 0], xhat = (y - mean) * invstd;  g_hidden / g_pooled may be NULL.
 * bwd_apply: gy = scale * (gz*mask - c1 - xhat * c2), c1 = dbeta/R, c2 = dgamma/R (zeros with running stats). */
int32_t trs_cin_glue_blocks(int64_t B);
int trs_cin_glue_stats(const void* y, int64_t B, int32_t E, int32_t C, int32_t dtype, float* partial,
                       trs_stream_t stream);
int trs_cin_glue_fwd(const void* y, const float* scale, const float* shift, int64_t B, int32_t E, int32_t C, int32_t D,
                     int32_t Hs, int32_t dtype, void* hidden, void* pooled, trs_stream_t stream);
int trs_cin_glue_bwd_reduce(const void* y, const void* g_hidden, const void* g_pooled, const float* scale,
                            const float* shift, const float* mean, const float* invstd, int64_t B, int32_t E, int32_t C,
                            int32_t D, int32_t Hs, int32_t dtype, float* partial, trs_stream_t stream);
int trs_cin_glue_bwd_apply(const void* y, const void* g_hidden, const void* g_pooled, const float* scale,
                           const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2,
                           int64_t B, int32_t E, int32_t C, int32_t D, int32_t Hs, int32_t dtype, void* gy,
                           trs_stream_t stream);
/* The same two passes, also writing their output channels-first: hidden_cf (B, C-Hs, E), gy_cf (B, C, E) -- the
 * operand layout of trs_cin_dw, which otherwise costs one transposing copy of a 1-2 GB tensor per layer and step.
 * Shape limit: E == 8 * (256 / (C / 8)) (e.g. E = 64, C = 256); trs_cin_glue_cf_supported() tells.
 * colsum_partial (optional, may be NULL): [trs_cin_glue_blocks(B)][C] fp32 per-workgroup column sums of gy as stored
 * -- the Conv1d bias gradient of the layer (compress_interaction_network.py:62), which otherwise costs one more pass
 * over the 2 GB tensor.                                                                                             */
int trs_cin_glue_cf_supported(int32_t E, int32_t C);
int trs_cin_glue_fwd_cf(const void* y, const float* scale, const float* shift, int64_t B, int32_t E, int32_t C,
                        int32_t D, int32_t Hs, int32_t dtype, void* hidden, void* hidden_cf, void* pooled,
                        trs_stream_t stream);
int trs_cin_glue_bwd_apply_cf(const void* y, const void* g_hidden, const void* g_pooled, const float* scale,
                              const float* shift, const float* mean, const float* invstd, const float* c1,
                              const float* c2, int64_t B, int32_t E, int32_t C, int32_t D, int32_t Hs, int32_t dtype,
                              void* gy, void* gy_cf, float* colsum_partial, trs_stream_t stream);

/* ---- OuterProductNetwork / BilinearInteraction on the (i<j) pair pattern -------
 * Pair p = (i_p, j_p), i<j, lexicographic (the order of inner_product_network.py:51-52); NC2 = N(N-1)/2.
 *
 * OPN 'vec' / 'num'  (outer_product_network.py:123-129):
 *   out[b,p] = sum_e x[b,i_p,e] * x[b,j_p,e] * kern[p,e]      kern (NC2,E); kern_is_num: kern (NC2), no e index
 * bwd: gx (B,N,E) and gkern_vec (NC2,E) fp32, ACCUMULATED into (for 'num' the caller sums it over e);
 * either may be NULL.                                                                              */
int trs_opn_vec_fwd(const void* x, const void* kern, int32_t kern_is_num, int64_t B, int32_t N, int32_t E,
                    int32_t dtype, void* out, trs_stream_t stream);
size_t trs_opn_vec_bwd_workspace_bytes(int64_t B, int32_t N, int32_t E);
int trs_opn_vec_bwd(const void* g, const void* x, const void* kern, int32_t kern_is_num, int64_t B, int32_t N,
                    int32_t E, int32_t dtype, void* gx, float* gkern_vec, void* workspace, size_t ws_bytes,
                    trs_stream_t stream);

/* pair product:  out[b,p,:] = a[b,i_p,:] * c[b,j_p,:] + bias[(bias_per_pair ? p : 0), :]   (bias may be NULL)
 * Bilinear 'all' (bilinear_interaction.py:72-76) is  a = x W (one GEMM), c = x, shared bias.
 * bwd: ga[b,i,:] = sum_{p: i_p=i} g[b,p,:] c[b,j_p,:],  gc[b,j,:] = sum_{p: j_p=j} g[b,p,:] a[b,i_p,:].   */
int trs_pair_mul_fwd(const void* a, const void* c, const void* bias, int32_t bias_per_pair, int64_t B, int32_t N,
                     int32_t E, int32_t dtype, void* out, trs_stream_t stream);
int trs_pair_mul_bwd(const void* g, const void* a, const void* c, int64_t B, int32_t N, int32_t E, int32_t dtype,
                     void* ga, void* gc, trs_stream_t stream);

/* rows product with bias: the stand-alone forward of FieldAllTypeBilinear / FieldEachTypeBilinear
 * (bilinear_interaction.py:72-76, 144-149), whose operands arrive already gathered per pair as (B,P,E):
 *   out[r,:] = a[r,:] * c[r,:] + bias[(bias_per_pair ? r % P : 0), :]      r over rows = B*P   (bias may be NULL)
 * bwd: ga = g * c, gc = g * a (either may be NULL).  Element-wise passes, any E.                     */
int trs_rows_mul_bias_fwd(const void* a, const void* c, const void* bias, int32_t bias_per_pair, int64_t rows,
                          int32_t P, int32_t E, int32_t dtype, void* out, trs_stream_t stream);
int trs_rows_mul_bwd(const void* g, const void* a, const void* c, int64_t rows, int32_t E, int32_t dtype, void* ga,
                     void* gc, trs_stream_t stream);

/* per-pair bilinear form:  T[b,p,:] = x[b,i_p,:] @ W[(w_per_pair ? p : 0)]     W (NC2 | 1, E, E) row-major [e][h]
 *   mode 0 (OPN 'mat', outer_product_network.py:107-121 with W[p][e][h] = kernel[h,p,e]):
 *           out[b,p]   = sum_h T[b,p,h] * x[b,j_p,h]
 *   mode 1 (Bilinear 'each', bilinear_interaction.py:144-149):
 *           out[b,p,h] = T[b,p,h] * x[b,j_p,h] + bias[(bias_per_pair ? p : 0), h]
 * bwd_data: gx (B,N,E); gT (B,NC2,E) = dL/dT, written when not NULL (the weight gradient
 * dW[p] = sum_b x[b,i_p,:]^T gT[b,p,:] is a plain GEMM per field over it).                          */
int trs_pair_bilinear_fwd(const void* x, const void* W, int32_t w_per_pair, const void* bias, int32_t bias_per_pair,
                          int32_t mode, int64_t B, int32_t N, int32_t E, int32_t dtype, void* out,
                          trs_stream_t stream);
int trs_pair_bilinear_bwd_data(const void* g, const void* x, const void* W, int32_t w_per_pair, int32_t mode,
                               int64_t B, int32_t N, int32_t E, int32_t dtype, void* gx, void* gT,
                               trs_stream_t stream);

/* The same forward on the matrix cores (bf16, E = 32 | 64), one kernel, no (B,NC2,E) intermediate:
 * Wt (NC2, E, E) = the per-pair matrices TRANSPOSED, Wt[p][h][e] = W[p][e][h] (for OPN 'mat' that is kernel[h,p,e]
 * itself); tasks (ntasks, 3) int32 on the device = (i, j0, count <= 3): every pair exactly once, the pairs of a task
 * share i and are adjacent in p.  mode / bias / out as trs_pair_bilinear_fwd (bias per pair, (NC2,E), or NULL).   */
int trs_pair_bilinear_fwd_mfma(const void* x, const void* Wt, const void* bias, const int32_t* tasks, int32_t ntasks,
                               int32_t mode, int64_t B, int32_t N, int32_t E, int32_t dtype, void* out,
                               trs_stream_t stream);

/* Backward of the same on the matrix cores (bf16, E = 32 | 64).  gv = g[b,p] (mode 0) | g[b,p,h] (mode 1); dL/dT = gv * x_j.
 * data:   gx_i = sum_j W_p (gv x_j)   -- tasks_i (nti,3) = (i, j0, count) as in the forward, W resident;
 *         gx_j = sum_i gv * (x_i W_p) -- tasks_j (ntj,3) = (j, i0, count): pairs (i0..i0+count-1, j), Wt resident;
 *         every (sample, task) writes one row into contrib_i (B,nti,E) / contrib_j (B,ntj,E) (bf16 scratch), then
 *         gx[b,f,:] = sum of the rows of field f's tasks;  seg_i / seg_j (N+1) = first task of each field.
 * weight: gW (NC2,E,E) [e][h] = sum_b x_i^T (gv x_j), K = samples through per-wave LDS transposes.               */
int trs_pair_bilinear_bwd_data_mfma(const void* g, const void* x, const void* W, const void* Wt,
                                    const int32_t* tasks_i, int32_t nti, const int32_t* seg_i,
                                    const int32_t* tasks_j, int32_t ntj, const int32_t* seg_j, int32_t mode, int64_t B,
                                    int32_t N, int32_t E, int32_t dtype, void* contrib_i, void* contrib_j, void* gx,
                                    trs_stream_t stream);
size_t trs_pair_bilinear_bwd_w_mfma_workspace_bytes(int64_t B, int32_t N, int32_t E);
int trs_pair_bilinear_bwd_w_mfma(const void* g, const void* x, int32_t mode, int64_t B, int32_t N, int32_t E,
                                 int32_t dtype, void* gW, void* workspace, size_t ws_bytes, trs_stream_t stream);

/* GEMM route of the same form for training batch sizes: T[b,p,:] = x[b,i_p,:] @ W_p comes from one plain GEMM per
 * field i (pairs (i, j>i) are adjacent: (B x E) @ (E x n_i*E)) into a (B,NC2,E) buffer; these passes finish it.
 *   fwd  mode 0: out[b,p] = sum_h T[b,p,h] x[b,j_p,h]       mode 1: T <- T * x_j + bias   (in place; out unused)
 *   bwd  gv = g[b,p] (mode 0) | g[b,p,h] (mode 1):  gxj[b,j_p,:] = sum_i gv * T[b,p,:];  T <- gv * x_j  (= dL/dT)   */
int trs_pair_epilogue_fwd(void* T, const void* x, const void* bias, int32_t bias_per_pair, int32_t mode, int64_t B,
                          int32_t N, int32_t E, int32_t dtype, void* out, trs_stream_t stream);
int trs_pair_epilogue_bwd(const void* g, const void* x, void* T, int32_t mode, int64_t B, int32_t N, int32_t E,
                          int32_t dtype, void* gxj, trs_stream_t stream);

/* AttentionalFactorizationMachineLayer (attentional_factorization_machine.py:86-125, dropouts outside):
 *   prod[b,p,:] = x[b,i_p,:] * x[b,j_p,:];  attn[b,p] = softmax_p(w2 . relu(W1 prod + b1) + b2);
 *   out[b,:] = sum_p attn[b,p] prod[b,p,:]            W1 (A,E), b1 (A), w2 (A), b2 (1); out (B,E), attn (B,NC2)
 * bwd: g_out (B,E) / g_attn (B,NC2) may be NULL; gx (B,N,E); gW1 (A,E), gb1 (A), gw2 (A), gb2 (1) fp32,
 * ACCUMULATED into.  E, A <= 128.                                                                   */
int trs_afm_fwd(const void* x, const void* W1, const void* b1, const void* w2, const void* b2, int64_t B, int32_t N,
                int32_t E, int32_t A, int32_t dtype, void* out, void* attn, trs_stream_t stream);
/* Host-side helper (no device work): the backward kernel's schedule of 16-pair tiles for N fields -- every pair (i < j)
 * exactly once, the pairs of a tile sharing no field (what lets a wave add into per-field gradient rows without
 * atomics), packed greedily from the round-robin rounds.  tiles[16 t + k] = (i << 8) | j, 0xffff = empty slot;
 * *ntiles = 0 when the table does not fit the kernel-argument block (the kernel then walks the rounds themselves).
 * attentional_factorization_machine.py:86-120 (the pair enumeration the attention runs over).                   */
int trs_afm_pair_tiles(int32_t N, uint16_t* tiles, int32_t capacity, int32_t* ntiles);
size_t trs_afm_bwd_workspace_bytes(int64_t B, int32_t N, int32_t E, int32_t A);
int trs_afm_bwd(const void* g_out, const void* g_attn, const void* x, const void* attn, const void* W1,
                const void* b1, const void* w2, int64_t B, int32_t N, int32_t E, int32_t A, int32_t dtype, void* gx,
                float* gW1, float* gb1, float* gw2, float* gb2, void* workspace, size_t ws_bytes,
                trs_stream_t stream);

/* The same layer with the reference's dropout on the attention scores applied INSIDE the pass
 * (attentional_factorization_machine.py:82, 105-113: nn.Dropout is the last module of ``self.attention``, so in training
 * the weighted sum and the returned scores both see the dropped scores):
 *   keep (B,NC2) uint8, nonzero = kept; multiplier m[b,p] = keep ? keep_scale : 0   (keep_scale = 1/(1-p))
 *   attn      (B,NC2) = the softmax BEFORE dropout (what the backward needs)
 *   attn_drop (B,NC2) = attn * m   (what the layer returns);   out[b,:] = sum_p attn_drop[b,p] prod[b,p,:]
 * keep == NULL: no dropout, attn_drop is not written (== trs_afm_fwd / trs_afm_bwd).
 * bwd: g_attn is the gradient of attn_drop; ``attn`` is the un-dropped softmax written by the forward.        */
int trs_afm_fwd_dropout(const void* x, const void* W1, const void* b1, const void* w2, const void* b2,
                        const uint8_t* keep, float keep_scale, int64_t B, int32_t N, int32_t E, int32_t A,
                        int32_t dtype, void* out, void* attn, void* attn_drop, trs_stream_t stream);
int trs_afm_bwd_dropout(const void* g_out, const void* g_attn, const void* x, const void* attn, const uint8_t* keep,
                        float keep_scale, const void* W1, const void* b1, const void* w2, int64_t B, int32_t N, int32_t E,
                        int32_t A, int32_t dtype, void* gx, float* gW1, float* gb1, float* gw2, float* gb2,
                        void* workspace, size_t ws_bytes, trs_stream_t stream);

/* ---- index staging: pack per-field columns into the (B,N) index matrix --------
 * out[b, c] = src_j[b * width_j + t]  for the c-th output column = column t of source j.
 * replaces the per-field unsqueeze + torch.cat of inputs/inputs.py:75-80 by one pass.
 * srcs / widths are HOST arrays (nsrc device pointers, their column counts); sum(widths) <= 256.
 * src_dtype / out_dtype: TRS_I64 | TRS_I32 (narrowing to int32 is the caller's responsibility).   */
int trs_pack_columns(const void* const* srcs, const int32_t* widths, int32_t nsrc, int32_t src_dtype,
                     int64_t B, void* out, int32_t out_dtype, trs_stream_t stream);

/* ---- the scalar head of the CTR models and its loss (callers M1 / M2 / M4 of section 8a) ----------
 * logit[b] = sum_e fm[b,e] + sum_n feat[b,n] + sum_k extras[k][b * extra_strides[k]] + bias[0]          -> out (B,1)
 *   models/ctr/factorization_machine.py:55-66 (fm_second.sum('O') + feat_inputs.sum('N') + bias),
 *   models/ctr/deep_fm.py:75-104 (cat([fm_second, fm_first]).sum('O') + deep_out),
 *   models/ctr/xdeep_fm.py:117-121 (feat.sum('N') + cin_out + deep_out + bias).
 * fm (B,E) / feat (B,N) contiguous, either may be NULL (E / N = 0); extras: HOST array of n_extras <= 4 device pointers to
 * one value per sample with element stride extra_strides[k] (a (B,1) column, or column 0 of a wider padded output);
 * bias: one device value or NULL.  fp32 accumulation, one rounding on store.  The gradient of every operand is the
 * incoming (B,1) column itself (broadcast): there is no backward entry point.                                        */
int trs_ctr_logit_fwd(const void* fm, int32_t E, const void* feat, int32_t N, const void* const* extras,
                      const int64_t* extra_strides, int32_t n_extras, const void* bias, int64_t B, int32_t dtype,
                      void* out, trs_stream_t stream);
/* BCEWithLogitsLoss, reduction = mean (the loss the fwd+bwd metric is defined on):
 *   loss = mean_b( max(x,0) - x*y + log1p(exp(-|x|)) )   -> *loss (fp32, device);  logits (B) f32|bf16, labels (B) f32|bf16
 *   bwd: glogits[b] = (sigmoid(x) - y) * gout[0] / B     (gout: fp32 device scalar, NULL = 1)
 * Two launches forward (per-workgroup partial sums in a fixed order: reproducible), one backward.                    */
size_t trs_bce_logits_workspace_bytes(int64_t B);
int trs_bce_logits_fwd(const void* logits, int32_t dtype, const void* labels, int32_t label_dtype, int64_t B, float* loss,
                       void* workspace, size_t ws_bytes, trs_stream_t stream);
int trs_bce_logits_bwd(const void* logits, int32_t dtype, const void* labels, int32_t label_dtype, const float* gout,
                       int64_t B, void* glogits, trs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TRS_ABI_H_ */
